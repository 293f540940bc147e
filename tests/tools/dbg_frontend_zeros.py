"""Debug aid: where do the exact zeros of the HIP front end and of the oracle differ (DeepSpeech2 batch-16 test batch)?"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import features as OF  # noqa: E402
from tests import test_real_configs_gpu as RC  # noqa: E402

dc, plan = RC._frontend()
B, seed = 16, 777
audio, n = RC._audio(B, 15.0, short={1: 11.0, 6: 4.2, 15: 14.1}, seed=99)
feats = plan(torch.from_numpy(audio).cuda(), torch.from_numpy(n).cuda(), plan.num_frames(audio.shape[1]),
             seed=torch.tensor([seed], dtype=torch.int32, device="cuda")).cpu().numpy()
sa = {k: dc["spec_augment"][k] for k in ("F", "m_F", "T", "p", "m_T")}
ref = OF.batch_features(audio.astype(np.float64), n, dc, seed=seed, spec_aug=sa)
for ch in range(3):
    bad = np.argwhere((feats[..., ch] == 0.0) != (ref[..., ch] == 0.0))
    print(f"channel {ch}: {len(bad)} mismatching zero positions")
    for b, t, f in bad[:12]:
        print(f"  clip {b} (n={n[b]}, frames={1 + (n[b] - 320) // 160}) frame {t} bin {f}: hip {feats[b, t, f, ch]!r} oracle {ref[b, t, f, ch]!r}"
              f"   hip row zeros {int((feats[b, t, :, ch] == 0).sum())} oracle row zeros {int((ref[b, t, :, ch] == 0).sum())}")
    if len(bad):
        clips = sorted(set(int(x) for x in bad[:, 0]))
        for b in clips:
            fr = sorted(set(int(x) for x in bad[bad[:, 0] == b][:, 1]))
            print(f"  clip {b}: frames {fr[:20]}{' ...' if len(fr) > 20 else ''}")
