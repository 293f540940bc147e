"""CPU: properties of the sweeps' DEVICE code that no functional test sees (hipcc cross-compiles gfx950 without a GPU).

The one-launch sweeps hand over between the waves of a workgroup through progress words in LDS.  Until round 4 those were
reached through `volatile int*` - a generic pointer, which the compiler keeps as flat_load / flat_store ... sc0 sc1 followed by
s_waitcnt vmcnt(0): the texture path's latency on every hand-over and an entry in the wave's in-order memory counter in front
of its polls (186 such instructions in the forward decoder sweep).  csrc/sweep_common.h's lds_flag_t / lds_peek / lds_poke give
ds_read / ds_write; this test keeps it that way."""
import os
import re
import shutil
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "speech-recognition_amd", "csrc")
SWEEPS = ["rnn_sweep", "rnn_sweep_bwd", "rnn_sweep_wide", "rnn_sweep_wide_bwd", "decoder_sweep", "decoder_sweep_bwd"]
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def device_asm(name, out_dir):
    out = os.path.join(out_dir, name + ".s")
    subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, name + ".hip")],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=CSRC)
    return open(out).read()


@pytest.fixture(scope="module")
def sweep_asm():
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not found")
    with tempfile.TemporaryDirectory() as d, ThreadPoolExecutor(max_workers=3) as pool:
        return dict(zip(SWEEPS, pool.map(lambda n: device_asm(n, d), SWEEPS)))


def test_sweeps_reach_their_lds_progress_words_with_ds_instructions(sweep_asm):
    for name, asm in sweep_asm.items():
        flat = re.findall(r"^\s*(flat_(?:load|store|atomic)\S*)", asm, flags=re.M)
        assert not flat, f"{name}.hip: {len(flat)} flat memory instructions in the device code ({sorted(set(flat))}): an LDS word is reached through a generic pointer"


def test_bptt_sweep_issues_its_operand_loads_without_waiting_between_them(sweep_asm):
    # the element-wise operands of the next step are buffer loads on (fixed per-lane offset, per-step scalar offset): six of them back
    # to back in the LSTM instance.  With flat addresses the compiler reused a destination register for the next address and put
    # s_waitcnt vmcnt(0) between the loads - the gather wave sat out a memory latency per step
    asm = sweep_asm["rnn_sweep_bwd"]
    start = asm.index("_Z20rnn_sweep_bwd_kernelILi0ELi2EEv6SbArgs:")
    body = asm[start:asm.index(".end_amdhsa_kernel", start)] if ".end_amdhsa_kernel" in asm[start:] else asm[start:]
    runs = re.findall(r"(?:^\s*buffer_load_dword\S*[^\n]*\n){6}", body, flags=re.M)
    assert len(runs) >= 2, "the two unrolled step bodies each fetch 2 x (k0, k1, dy) in one run of buffer loads"
    i = asm.index(".name:           _Z20rnn_sweep_bwd_kernelILi0ELi2EEv6SbArgs")
    assert re.search(r"\.vgpr_spill_count:\s+0\b", asm[i:i + 1500]), "the BPTT sweep must not spill"


def test_bptt_sweep_refuses_shapes_beyond_its_32_bit_buffer_offsets():
    # coefficient packs [B, T, H, 8] f32 are read through buffer loads with 32-bit offsets: 2 GB is the limit, larger layers take
    # the per-step kernels (asr_rnn_sweep_bwd_supported is what layers.py asks)
    from speech_recognition_amd import _lib
    lib = _lib.load()
    assert lib.asr_rnn_sweep_bwd_supported(0, 32, 8100, 256, 2) == 1          # 32 * 8100 * 256 * 32 B = 2.12e9 < 2^31
    assert lib.asr_rnn_sweep_bwd_supported(0, 32, 8300, 256, 2) == 0          # 2.18e9 >= 2^31
