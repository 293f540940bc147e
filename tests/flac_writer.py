"""A small FLAC *encoder* (test infrastructure) written from the format description, used to exercise
every branch of the native decoder (asr_audio_decode): subframe types CONSTANT / VERBATIM / FIXED /
LPC, both Rice parameter widths, escaped partitions, wasted bits, and the four channel assignments.
It makes no attempt to compress well - parameters are chosen by the test."""
import struct

import numpy as np


class BitWriter:
    def __init__(self):
        self.bits = []

    def write(self, value, n):
        value &= (1 << n) - 1 if n else 0
        for k in range(n - 1, -1, -1):
            self.bits.append((value >> k) & 1)

    def write_signed(self, value, n):
        self.write(int(value) & ((1 << n) - 1), n)

    def unary(self, q):
        self.bits.extend([0] * q)
        self.bits.append(1)

    def align(self):
        while len(self.bits) % 8:
            self.bits.append(0)

    def tobytes(self):
        assert len(self.bits) % 8 == 0
        arr = np.array(self.bits, np.uint8).reshape(-1, 8)
        return bytes(np.packbits(arr, axis=1).reshape(-1))


def crc8(data):
    c = 0
    for b in data:
        c ^= b
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xFF if c & 0x80 else (c << 1) & 0xFF
    return c


def crc16(data):
    c = 0
    for b in data:
        c ^= b << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xFFFF if c & 0x8000 else (c << 1) & 0xFFFF
    return c


def _residual(bw, res, order, blocksize, porder, method, params):
    """params: per-partition Rice parameter, or ('esc', nbits) for an escaped partition."""
    bw.write(method, 2)
    bw.write(porder, 4)
    pbits = 4 if method == 0 else 5
    pos = 0
    for part in range(1 << porder):
        count = (blocksize >> porder) - (order if part == 0 else 0)
        vals = res[pos:pos + count]
        pos += count
        p = params[part]
        if isinstance(p, tuple):
            bw.write((1 << pbits) - 1, pbits)
            bw.write(p[1], 5)
            for v in vals:
                bw.write_signed(v, p[1])
        else:
            bw.write(p, pbits)
            for v in vals:
                v = int(v)
                u = (v << 1) if v >= 0 else ((-v) << 1) - 1
                bw.unary(u >> p)
                bw.write(u & ((1 << p) - 1), p)


_FIXED = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}


def subframe(bw, x, bps, spec):
    """x: int samples of one channel.  spec: dict(type='constant'|'verbatim'|'fixed'|'lpc', order, porder,
    method, params, coefs, precision, shift, wasted)."""
    x = np.asarray(x, np.int64)
    wasted = spec.get("wasted", 0)
    kind = spec["type"]
    code = {"constant": 0, "verbatim": 1}.get(kind)
    if kind == "fixed":
        code = 8 + spec["order"]
    elif kind == "lpc":
        code = 32 + spec["order"] - 1
    bw.write(0, 1)
    bw.write(code, 6)
    if wasted:
        bw.write(1, 1)
        bw.unary(wasted - 1)
        assert (x % (1 << wasted) == 0).all()
        x = x >> wasted
        bps -= wasted
    else:
        bw.write(0, 1)
    n = len(x)
    if kind == "constant":
        bw.write_signed(x[0], bps)
    elif kind == "verbatim":
        for v in x:
            bw.write_signed(v, bps)
    else:
        order = spec["order"]
        for v in x[:order]:
            bw.write_signed(v, bps)
        if kind == "fixed":
            coefs, shift = _FIXED[order], 0
        else:
            coefs, shift = spec["coefs"], spec["shift"]
            bw.write(spec["precision"] - 1, 4)
            bw.write_signed(shift, 5)
            for c in coefs:
                bw.write_signed(c, spec["precision"])
        res = np.zeros(n - order, np.int64)
        for i in range(order, n):
            pred = sum(int(c) * int(x[i - 1 - j]) for j, c in enumerate(coefs)) >> shift
            res[i - order] = int(x[i]) - pred
        _residual(bw, res, order, n, spec.get("porder", 0), spec.get("method", 0), spec["params"])


def encode(channels, sample_rate=16000, bps=16, blocksize=256, assignment="independent", specs=None, total_known=True):
    """channels: list of int arrays (1 or 2).  specs: per-channel subframe spec (applied to every frame)."""
    chans = [np.asarray(c, np.int64) for c in channels]
    n = len(chans[0])
    nch = len(chans)
    out = bytearray(b"fLaC")
    info = BitWriter()
    info.write(blocksize, 16); info.write(blocksize, 16); info.write(0, 24); info.write(0, 24)
    info.write(sample_rate, 20); info.write(nch - 1, 3); info.write(bps - 1, 5); info.write(n if total_known else 0, 36)
    for _ in range(16):
        info.write(0, 8)
    body = info.tobytes()
    out += bytes([0x80]) + struct.pack(">I", len(body))[1:] + body
    assign_code = {"independent": nch - 1, "left_side": 8, "right_side": 9, "mid_side": 10}[assignment]
    frame_no = 0
    for start in range(0, n, blocksize):
        blk = [c[start:start + blocksize] for c in chans]
        bs = len(blk[0])
        bw = BitWriter()
        bw.write(0xFFF8, 16)
        bw.write(7, 4)                 # 16-bit (blocksize - 1) follows the frame number
        bw.write(0, 4)                 # sample rate from STREAMINFO
        bw.write(assign_code, 4)
        bw.write({8: 1, 12: 2, 16: 4, 20: 5, 24: 6}[bps], 3)
        bw.write(0, 1)
        assert frame_no < 128
        bw.write(frame_no, 8)
        bw.write(bs - 1, 16)
        header = bw.tobytes()
        bw.write(crc8(header), 8)
        if assignment == "left_side":
            coded, widths = [blk[0], blk[0] - blk[1]], [bps, bps + 1]
        elif assignment == "right_side":
            coded, widths = [blk[0] - blk[1], blk[1]], [bps + 1, bps]
        elif assignment == "mid_side":
            coded, widths = [(blk[0] + blk[1]) >> 1, blk[0] - blk[1]], [bps, bps + 1]
        else:
            coded, widths = blk, [bps] * nch
        for c, w, spec in zip(coded, widths, specs):
            spec = dict(spec)
            if spec["type"] in ("fixed", "lpc") and spec["order"] > bs:
                spec = dict(type="verbatim")
            if "params" in spec and (bs >> spec.get("porder", 0)) << spec.get("porder", 0) != bs:
                spec["porder"], spec["params"] = 0, spec["params"][:1]
            subframe(bw, c, w, spec)
        bw.align()
        frame = bw.tobytes()
        out += frame + struct.pack(">H", crc16(frame))
        frame_no += 1
    return bytes(out)
