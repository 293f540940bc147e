"""Host decode throughput (not a test): FLAC / WAV -> mono float32 through the native decoders, single thread and
on the Dataset.map thread pool.  Reported as x real time at 16 kHz."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import flac_writer as FW
from speech_recognition_amd.data import Dataset, load_audio_file, AUTOTUNE

g = np.random.default_rng(0)
n = 16000 * 10
t = np.arange(n)
pcm = np.clip(8000 * np.sin(2 * np.pi * 0.01 * t) + g.normal(0, 400, n), -32768, 32767).astype(np.int64)
blob = FW.encode([pcm], 16000, 16, 4096, "independent", [dict(type="fixed", order=2, porder=3, method=0, params=[10] * 8)])
d = tempfile.mkdtemp()
path = os.path.join(d, "a.flac")
open(path, "wb").write(blob)
load = load_audio_file(16000, "flac")
assert np.array_equal(load(path), (pcm / 32768.0).astype(np.float32))
reps = 100
t0 = time.perf_counter()
for _ in range(reps): load(path)
dt = time.perf_counter() - t0
print(f"FLAC decode, 1 thread : {reps * 10 / dt:8.0f} x real time ({len(blob) / 1e3:.0f} KB per 10 s clip)")
ds = Dataset.from_iterable([path] * (reps * 4)).map(load, num_parallel_calls=AUTOTUNE)
t0 = time.perf_counter()
cnt = sum(1 for _ in ds)
dt = time.perf_counter() - t0
print(f"FLAC decode, thread pool: {cnt * 10 / dt:8.0f} x real time")
