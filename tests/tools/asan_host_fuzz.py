"""Drives the AddressSanitizer + UBSan build of the host parsers (make -C speech-recognition_amd/csrc asan ->
speech-recognition_amd/libasr_host_asan.so) with truncated and corrupted inputs.  Run by tests/test_host_asan.py in a child
process with libasan preloaded; any out-of-bounds access, use-after-free or undefined behaviour aborts the process (non-zero
exit), a malformed file must come back as an error CODE.

What is fed (the reference reads these bytes with tensorflow-io, data.py:94-117, and TFRecord framing, data.py:75):
  * test.wav / test.flac of the reference fixtures and synthetic FLAC streams of every subframe type, truncated at every length
    up to 4 KB and at 200 seeded random lengths beyond, then with 1-8 seeded byte flips each (1500 variants per file);
  * headers that lie: WAV chunk sizes beyond the file, 0 / 255 channels, 8- and 32-bit samples, FLAC block sizes and sample
    counts beyond the stream, capacity smaller than the clip;
  * asr_crc32c at every alignment and length 0..300;
  * asr_ctc_beam_search on random log-probabilities: T = 1, C = 2, beam 1, top_paths = beam, seq_len 0 / 1 / T, -inf rows, NaN rows.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
FIX = os.path.join(ROOT, "tests", "golden", "reference_fixtures", "audio_files")


class AudioInfo(C.Structure):
    _fields_ = [("sample_rate", C.c_int), ("channels", C.c_int), ("bits_per_sample", C.c_int), ("frames", C.c_long)]


def main():
    lib = C.CDLL(os.path.join(ROOT, "speech-recognition_amd", "libasr_host_asan.so"))
    lib.asr_audio_info.restype = C.c_int
    lib.asr_audio_info.argtypes = [C.c_char_p, C.c_long, C.c_int, C.POINTER(AudioInfo)]
    lib.asr_audio_decode.restype = C.c_int
    lib.asr_audio_decode.argtypes = [C.c_char_p, C.c_long, C.c_int, C.c_void_p, C.c_long, C.POINTER(C.c_long)]
    lib.asr_crc32c.restype = C.c_uint32
    lib.asr_crc32c.argtypes = [C.c_void_p, C.c_long, C.c_uint32]
    lib.asr_ctc_beam_search.restype = C.c_int
    lib.asr_ctc_beam_search.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    rng = np.random.default_rng(2024)
    stats = dict(decoded=0, rejected=0)

    def feed(blob, fmt, capacity=None):
        """info + decode of one blob: exact-size heap copies, so that one byte past the end is a redzone."""
        buf = C.create_string_buffer(bytes(blob), len(blob)) if len(blob) else C.create_string_buffer(1)
        info = AudioInfo()
        rc = lib.asr_audio_info(buf, len(blob), fmt, C.byref(info))
        if rc != 0:
            stats["rejected"] += 1
            return
        frames = int(info.frames)
        if frames < 0 or frames > (1 << 26):
            stats["rejected"] += 1
            return
        cap = frames if capacity is None else capacity
        out = np.empty(max(cap, 1), np.float32)
        n = C.c_long(0)
        rc = lib.asr_audio_decode(buf, len(blob), fmt, out.ctypes.data_as(C.c_void_p), cap, C.byref(n))
        if rc == 0:
            assert 0 <= n.value <= cap, (n.value, cap)
            assert np.isfinite(out[:n.value]).all() and (np.abs(out[:n.value]) <= 1.0).all()
            stats["decoded"] += 1
        else:
            stats["rejected"] += 1

    from tests import flac_writer as FW
    blobs = [(open(os.path.join(FIX, "test.wav"), "rb").read(), 0), (open(os.path.join(FIX, "test.flac"), "rb").read(), 1),
             (open(os.path.join(FIX, "test.pcm"), "rb").read()[:5001], 2)]
    t = np.arange(256 * 3 + 77)
    chans = [np.clip(9000 * np.sin(2 * np.pi * (0.01 + 0.013 * c) * t) + rng.normal(0, 300, t.size), -32768, 32767).astype(np.int64) for c in range(2)]
    specs = [dict(type="verbatim"), dict(type="constant"),
             dict(type="fixed", order=2, porder=2, method=1, params=[8, 9, 17, 8]),
             dict(type="fixed", order=4, porder=3, method=0, params=[10, 9, ("esc", 18), 9, 10, 11, 9, 10]),
             dict(type="fixed", order=1, porder=0, method=0, params=[7], wasted=3),
             dict(type="lpc", order=8, precision=14, shift=12, coefs=[5000, -2100, 900, -400, 150, -60, 20, -5], porder=2, method=1, params=[9, 9, 9, 9])]
    for spec in specs:
        try:
            src = [np.full_like(chans[0], 1234)] if spec["type"] == "constant" else ([(chans[0] >> 3) << 3] if spec.get("wasted") else [chans[0]])
            blobs.append((FW.encode(src, 16000, 16, 256, "independent", [spec]), 1))
        except Exception as e:                                      # the writer, not the library under test
            print("flac_writer:", spec["type"], e)
    st = dict(type="fixed", order=2, porder=1, method=0, params=[11, 11])
    for assignment in ("independent", "left_side", "right_side", "mid_side"):
        blobs.append((FW.encode(chans, 22050, 16, 512, assignment, [st, st]), 1))
    blobs.append((FW.encode([chans[0]], 16000, 16, 256, "independent", [specs[2]], total_known=False), 1))

    for blob, fmt in blobs:
        blob = bytes(blob)
        feed(blob, fmt)
        cuts = list(range(0, min(len(blob), 4096))) + sorted(rng.integers(0, len(blob), 200).tolist())
        for c in cuts:
            feed(blob[:c], fmt)
        for _ in range(1500):
            b = bytearray(blob)
            for _k in range(int(rng.integers(1, 9))):
                # bias the flips towards the headers, where lengths and counts live
                pos = int(rng.integers(0, min(len(b), 256))) if rng.random() < 0.6 else int(rng.integers(0, len(b)))
                b[pos] = int(rng.integers(0, 256))
            feed(bytes(b), fmt)
        feed(blob, fmt, capacity=7)              # a buffer smaller than the clip must be refused, not overrun
    # lying WAV headers
    wav = bytearray(blobs[0][0])
    for off, vals in ((4, (0, 0xFFFFFFFF)), (16, (0, 2, 0xFFFFFFF0)), (22, (0, 255)), (34, (8, 32, 0)), (40, (0xFFFFFFFF, 0x7FFFFFFF, 3))):
        for v in vals:
            b = bytearray(wav)
            width = 2 if off in (22, 34) else 4
            b[off:off + width] = int(v & (0xFFFF if width == 2 else 0xFFFFFFFF)).to_bytes(width, "little")
            feed(bytes(b), 0)
    # CRC-32C: every alignment and length
    raw = np.frombuffer(rng.bytes(400), np.uint8).copy()
    for a in range(8):
        for n in range(0, 300, 7):
            lib.asr_crc32c(raw[a:].ctypes.data_as(C.c_void_p), n, 0)
    # CTC prefix beam search
    cases = 0
    for B, T, Cc, beam, top in ((1, 1, 2, 1, 1), (2, 5, 3, 4, 4), (3, 17, 40, 8, 3), (1, 30, 200, 16, 16), (2, 9, 5, 32, 1), (4, 12, 7, 2, 2)):
        for variant in range(4):
            lp = rng.standard_normal((B, T, Cc)).astype(np.float32)
            lp = lp - np.log(np.exp(lp).sum(-1, keepdims=True))
            if variant == 1:
                lp[:, T // 2] = -np.inf
            if variant == 2:
                lp[0, 0, :] = np.nan
            seq = None
            if variant == 3:
                seq = np.array([0, 1, T, T // 2][:B] + [T] * max(0, B - 4), np.int32)
            toks = np.zeros((B, top, T), np.int32)
            lens = np.zeros((B, top), np.int32)
            score = np.zeros((B, top), np.float32)
            rc = lib.asr_ctc_beam_search(lp.ctypes.data_as(C.c_void_p), B, T, Cc, None if seq is None else seq.ctypes.data_as(C.c_void_p), beam, top,
                                         toks.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p), score.ctypes.data_as(C.c_void_p), 2)
            if rc == 0:
                assert (lens >= 0).all() and (lens <= T).all()
                assert ((toks >= 0) & (toks < Cc)).all()
            cases += 1
    print(f"asan host fuzz: {stats['decoded']} inputs decoded, {stats['rejected']} rejected, {cases} beam searches - no sanitizer report")


if __name__ == "__main__":
    main()
