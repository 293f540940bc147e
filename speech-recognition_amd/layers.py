"""Host-side building blocks shared by the LAS and DeepSpeech2 models: Dense / BiRNN layers with
explicit forward and backward passes that only launch kernels of libasr_mi355x.so.

There is no autograd here on purpose: every gradient is a hand-written kernel (or the MFMA GEMM)
writing straight into the flat gradient buffer, in an order the data-parallel step can overlap with
communication.  Buffers are allocated once per shape so that a whole training step can be captured
in a hipGraph.
"""
import math
from typing import Dict, List, Optional

import torch

from . import ops
from . import rng as R

import os

# ASR_PERSISTENT_RNN=0 forces the one-launch-per-step recurrent kernels (debugging / A-B timing)
PERSISTENT_RNN = os.environ.get("ASR_PERSISTENT_RNN", "1") != "0"
WIDE_SWEEP_BWD = os.environ.get("ASR_WIDE_SWEEP_BWD", "1") != "0"  # the wide layers' BPTT as one launch (rnn_sweep_wide_bwd.hip)
WIDE_SWEEP = os.environ.get("ASR_WIDE_SWEEP", "1") != "0"      # the bf16 weights-resident forward sweep of wide layers (mixed precision)

NG = {"lstm": 4, "gru": 3, "rnn": 1}   # gates in the Keras kernel layout
NS = {"lstm": 4, "gru": 4, "rnn": 1}   # columns saved per unit by the cell kernels ("slots")


class SideStream:
    """A second HIP stream for work that is off the backward pass's critical path (weight gradients): it is
    forked from the current stream by an event, so it also becomes a parallel branch of a captured hipGraph,
    and must be join()ed before the segment ends.  The recurrent sweeps and the decoder chain are latency
    bound and leave the matrix pipes idle; the weight-gradient GEMMs that nobody downstream waits for run
    beside them.  Whether that pays is measured per site: deepspeech 18.95 -> 18.38 ms/step (site "ds2", on by
    default); las_small 17.0 -> 17.6 (encoder weight gradients, "enc") and -> 17.5 (vocabulary weight gradient
    beside the decoder chain, "dec") - the co-running GEMM waves slow the hand-offs of the persistent sweeps and
    the short dependent decoder kernels by more than they save, so both are off by default.
    ASR_SIDE_STREAM = comma list of sites to enable ("enc,dec,ds2"), "1" = all, "0" = none."""

    sites = os.environ.get("ASR_SIDE_STREAM")

    def __init__(self, site="enc", default_on=False):
        if SideStream.sites is None:
            on = default_on
        else:
            on = SideStream.sites == "1" or site in SideStream.sites.split(",")
        self.on = on
        self.stream = None                                       # created on first use (a model may be built without a GPU)
        self._busy = False

    def run(self, fn):
        if not self.on:
            fn()
            return
        if self.stream is None:
            self.stream = torch.cuda.Stream()
        main = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(main)
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ev)
            fn()
        self._busy = True

    def fork(self):
        """An event on the current stream marking 'everything enqueued so far': work given to run_after() starts behind it, NOT
        behind what the caller enqueues between fork() and run_after() (a sweep the work is meant to run beside)."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        return ev

    def run_after(self, ev, fns, gate=None):
        """Enqueue the callables `fns` on the side stream behind the fork event `ev`; gate: the diagnosis words of a sweep that was
        launched on the main stream after fork() - the side stream then first waits (asr_sweep_gate, bounded) until that sweep is
        resident, so that its workgroups cannot be kept off the chip by the work released here."""
        if self.stream is None:
            self.stream = torch.cuda.Stream()
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ev)
            if gate is not None:
                ops.sweep_gate(gate)
            for fn in fns:
                fn()
        self._busy = True

    def join(self):
        if self.stream is not None and self._busy:
            ev = torch.cuda.Event()
            ev.record(self.stream)
            torch.cuda.current_stream().wait_event(ev)
            self._busy = False


class Overlap:
    """Off-critical-path work of the backward pass (weight and bias gradients: nothing downstream in the step reads them before
    Adam) held back and released BESIDE the next one-launch sweep, behind a gate that lets the sweep become resident first
    (SideStream.run_after / asr_sweep_gate).  The sweeps are bound by the latency of their dependent hand-offs and keep the matrix
    pipes ~5 % busy; the dW / dU / db products are throughput work (las_small: ~1.9 ms of them in an 11.8 ms step).
    MEASURED (round 3, MI355X) AND OFF BY DEFAULT: side by side both lose.  One 105-us GEMM beside the las_small BPTT sweep takes
    141 us and the sweep 968 instead of 882 us - the sweep advances at ~40 % of its speed while the GEMM's waves share its compute
    units (tests/tools/bench_beside.py); whole step 11.8 -> 13.2 ms (wave priority 0 / 3, probe polling, 64 x 64 tiles: the same).
    A latency-bound chain wants the chip to itself; ASR_OVERLAP=1 keeps the experiment runnable.  What the experiment did find: two
    races of the BPTT sweep that only other kernels' interference exposed (rnn_sweep_bwd.hip: ds out of place, re-arm three steps late)."""

    enabled = os.environ.get("ASR_OVERLAP", "0") == "1"
    # Round 4, the arrangement that does pay: off-critical-path products run on the side stream BESIDE THE OTHER DENSE WORK of their
    # stage (the input-gradient products, batch norm backward, the next projection's input gradient) and are joined BEFORE the next
    # sweep starts, so a sweep always has the chip to itself.  Two streams of short-K products fill each other's prologues, C writes
    # and ragged last rounds (each launch alone keeps the matrix pipes 40-53 % busy).  ASR_CONCURRENT=0 switches it off.
    concurrent = os.environ.get("ASR_CONCURRENT", "1") != "0"

    lanes = int(os.environ.get("ASR_CONCURRENT_LANES", "1"))       # side streams the concurrent mode deals its work over

    def __init__(self, site="overlap"):
        self.mode = "beside" if Overlap.enabled else ("concurrent" if Overlap.concurrent else "off")
        self.side = SideStream(site, default_on=self.mode != "off")
        # concurrent mode: independent deferred products are dealt round-robin over `lanes` side streams (a layer's two directions'
        # weight gradients run beside each other as well as beside the main stream's input-gradient products)
        self.more = [SideStream(site, default_on=self.side.on) for _ in range(max(0, Overlap.lanes - 1))] if self.mode == "concurrent" else []
        self._rr = 0
        self.pending = []

    @property
    def on(self):
        return self.side.on

    @property
    def late_buckets(self):
        """True when a stage's weight gradients complete one backward segment late (released beside the NEXT stage's sweep)."""
        return self.side.on and self.mode == "beside"

    def defer(self, fn):
        if not self.side.on:
            fn()
        elif self.mode == "concurrent":
            lanes = [self.side] + self.more    # now, on a side stream, beside whatever the caller enqueues next on the main stream
            lanes[self._rr % len(lanes)].run(fn)
            self._rr += 1
        else:
            self.pending.append(fn)

    def beside(self, launch, gate):
        """Run `launch()` (a sweep) on the current stream and release the pending work next to it ("beside" mode); in "concurrent"
        mode the sweep first waits for the side stream: it runs alone."""
        if self.mode == "concurrent":
            self.join_all()
            launch()
            return
        if not self.pending:
            launch()
            return
        ev = self.side.fork()
        launch()
        work, self.pending = self.pending, []
        self.side.run_after(ev, work, gate)

    def join_all(self):
        self.side.join()
        for sd in self.more:
            sd.join()
        self._rr = 0

    def flush(self, join=True):
        """Release what is still pending behind everything enqueued so far (no sweep to hide behind) and, with join, make the
        current stream wait for the side stream: the end of a backward pass / of a gradient bucket's segment."""
        if self.pending:
            work, self.pending = self.pending, []
            self.side.run_after(self.side.fork(), work, None)
        if join:
            self.join_all()


def auto_split_k(M, N, K):
    """K partitions for a weight-gradient GEMM (tiny M x N, huge K) so that ~1k workgroups run."""
    # ~2 workgroups per CU: more partitions only add atomic traffic (every partition adds M*N floats)
    tiles = math.ceil(M / 128) * math.ceil(N / 128) if M > 64 and N > 64 else math.ceil(M / 64) * math.ceil(N / 64)
    s = max(1, min(512 // max(tiles, 1), K // 256))
    return max(1, s)


def dense_fwd(x2d, W, b, out, relu=False, a_scale=None, a_rpg=0):
    return ops.gemm(x2d, W, out, bias=b, relu=relu, a_scale=a_scale, a_rpg=a_rpg)


def dense_bwd(x2d, W, dy2d, gW, gb, dx2d=None, dx_accumulate=False, a_scale=None, a_rpg=0, c_scale=None, c_rpg=0):
    """gW += x^T dy ; gb += colsum(dy) ; dx (+)= dy W^T.  Any of gW / gb / dx2d may be None (skipped)."""
    Kd = x2d.shape[0]
    if gW is not None:
        ops.gemm(x2d, dy2d, gW, trans_a=True, accumulate=1, split_k=auto_split_k(gW.shape[0], gW.shape[1], Kd), a_scale=a_scale,
                 a_rpg=a_rpg)
    if gb is not None:
        ops.colsum(dy2d, gb)
    if dx2d is not None:
        M, N, Kc = dx2d.shape[0], dx2d.shape[1], dy2d.shape[1]
        tiles = math.ceil(M / 64) * math.ceil(N / 64)
        if tiles < 256 and Kc >= 4096 and c_scale is None and dx2d.is_contiguous():
            # few output tiles, long reduction (vocabulary projection): partition K across workgroups
            if not dx_accumulate:
                ops.fill(dx2d, 0.0)
            ops.gemm(dy2d, W, dx2d, trans_b=True, accumulate=1, split_k=max(2, min(16, Kc // 1024)))
        else:
            ops.gemm(dy2d, W, dx2d, trans_b=True, accumulate=1 if dx_accumulate else 0, c_scale=c_scale, c_rpg=c_rpg)


def slot_cols(rnn_type, H, which):
    """Column ranges of the saved/dslots buffer [.., NS*H]: 'input' part (multiplies W) and 'rec' part
    (multiplies U), each as list of (src_col0, dst_col0, ncols) into the Keras [G*H] layout."""
    if rnn_type == "gru":
        if which == "input":
            return [(0, 0, 3 * H)]
        return [(0, 0, 2 * H), (3 * H, 2 * H, H)]
    n = NG[rnn_type] * H
    return [(0, 0, n)]


def cell_param_grads(rnn_type, H, x2d, hprev2d, ds2d, gW, gU, gb, a_scale=None, a_rpg=0):
    """Weight gradients of one cell from the gate-sum gradients ds2d [R, NS*H] (R = rows in time/batch).
    x2d [R, Din] (None = skip gW), hprev2d [R, H] (None = skip gU)."""
    Rr = ds2d.shape[0]
    for s0, d0, n in slot_cols(rnn_type, H, "input"):
        dsv = ds2d[:, s0:s0 + n]
        if x2d is not None:
            ops.gemm(x2d, dsv, gW[:, d0:d0 + n], trans_a=True, accumulate=1, split_k=auto_split_k(gW.shape[0], n, Rr),
                     a_scale=a_scale, a_rpg=a_rpg)
        if gb is not None:
            ops.colsum(dsv, (gb[0] if rnn_type == "gru" else gb)[d0:d0 + n])
    for s0, d0, n in slot_cols(rnn_type, H, "rec"):
        dsv = ds2d[:, s0:s0 + n]
        if hprev2d is not None:
            ops.gemm(hprev2d, dsv, gU[:, d0:d0 + n], trans_a=True, accumulate=1, split_k=auto_split_k(H, n, Rr))
        if rnn_type == "gru" and gb is not None:
            ops.colsum(dsv, gb[1][d0:d0 + n])


def cell_input_grad(rnn_type, H, ds2d, W, dx2d, accumulate=False, c_scale=None, c_rpg=0):
    """dx (+)= ds[:, input slots] W^T (optionally times a dropout table)."""
    (s0, d0, n), = slot_cols(rnn_type, H, "input")
    ops.gemm(ds2d[:, s0:s0 + n], W[:, d0:d0 + n], dx2d, trans_b=True, accumulate=1 if accumulate else 0, c_scale=c_scale, c_rpg=c_rpg)


class BiRNN:
    """BiRNN of las.py:62-126 on the step kernels: forward + backward LSTM/GRU/SimpleRNN over
    x [B,T,Din] with a frame mask, chained initial states and Keras input dropout."""

    fwd_side = None          # SideStream shared by all layers (set below the class): direction 1's input projection beside direction 0's

    def __init__(self, store, prefix, rnn_type, Din, H, dropout, stream_in, device="cuda", recurrent_dropout=0.0, stream_rec=None):
        ops.rnn_type_id(rnn_type)
        self.store, self.prefix, self.rt, self.Din, self.H = store, prefix, rnn_type, Din, H
        self.dropout, self.stream_in = float(dropout), stream_in
        # las.py:84-105 / deepspeech2.py:95-107: Keras recurrent dropout = one [B,H] multiplier per call on h_tm1, constant
        # over time (per direction).  Runs on the per-step kernels (the persistent launches do not take it).
        self.recurrent_dropout, self.stream_rec = float(recurrent_dropout), stream_rec
        if self.recurrent_dropout > 0 and stream_rec is None:
            raise ValueError("recurrent_dropout needs an RNG stream id")
        self.cells = [ops.PackedCell(rnn_type, H, [H], device) for _ in range(2)]
        self.names = [prefix + d + "/cell/" for d in ("forward_rnn", "backward_rnn")]

    @staticmethod
    def param_shapes(prefix, rnn_type, Din, H):
        g = NG[rnn_type]
        s = {}
        for d in ("forward_rnn", "backward_rnn"):
            s[f"{prefix}{d}/cell/kernel"] = (Din, g * H)
            s[f"{prefix}{d}/cell/recurrent_kernel"] = (H, g * H)
            s[f"{prefix}{d}/cell/bias"] = (2, g * H) if rnn_type == "gru" else (g * H,)
        return s

    def pack(self):
        for d in range(2):
            self.cells[d].pack([(self.store.p[self.names[d] + "recurrent_kernel"], True)])

    def pack_list(self):
        """(cell, weights) pairs of this layer for ops.pack_cells (one launch for a whole model)."""
        return [(self.cells[d], [(self.store.p[self.names[d] + "recurrent_kernel"], True)]) for d in range(2)]

    def dropout_table_list(self, buf, training):
        """(table, stream, rate) of this layer's input-dropout tables for ops.dropout_tables (one launch for a whole model); forward()
        called with tables_ready=True then does not draw them again."""
        if not (training and self.dropout > 0):
            return []
        return [(dd["mtab"], self.stream_in + d, self.dropout) for d, dd in enumerate(buf["dirs"])]

    def alloc(self, B, T, device="cuda"):
        H, rt = self.H, self.rt
        f = lambda *s: torch.empty(*s, device=device, dtype=torch.float32)
        buf = dict(B=B, T=T, y=f(B, T, 2 * H), dirs=[])
        for d in range(2):
            dd = dict(pre=f(B, T, NG[rt] * H), hseq=f(B, T, H), mtab=f(B, self.Din), direct=f(B, H),
                      dy_carry=f(B, H), dh0=f(B, H), reverse=(d == 1), cell=self.cells[d])
            if self.recurrent_dropout > 0:
                # Keras implementation 1 (forced by recurrent_dropout != 0): one mask per gate on h_tm1 AND on the input
                dd["rtab"] = f(NG[rt], B, H)
                dd["mtab_g"] = f(NG[rt], B, self.Din)
            dd["saved"] = f(B, T, NS[rt] * H) if rt == "gru" else dd["pre"]
            if rt == "lstm":
                dd["cseq"] = f(B, T, H)
            buf["dirs"].append(dd)
        # both one-launch sweeps take this layer: the forward sweep then leaves the element-wise backward of every (row, step, unit) as
        # coefficients (asr_rnn_seq.coef) instead of the saved activations, and the BPTT sweep writes the gate-sum gradients into the
        # pre-activation buffer (dead once the forward sweep has consumed it; GRU: a buffer of its own, 4 slots against 3 gates)
        buf["coef_mode"] = bool(PERSISTENT_RNN and self.recurrent_dropout == 0 and T >= 2 and ops.rnn_persist_supported(rt, B, T, H, 2)
                                and ops.rnn_persist_bwd_supported(rt, B, T, H, 2))
        if buf["coef_mode"]:
            for dd in buf["dirs"]:
                dd["coef_buf"] = f(B, T, H * ops.rnn_coef_width(rt))
                dd["ds"] = f(B, T, NS[rt] * H) if rt == "gru" else dd["pre"]
        # wide layers under mixed precision: the BPTT sweep with resident bf16 blocks of U (rnn_sweep_wide_bwd.hip).  It reads the saved
        # activations while it writes ds (the workgroups of a grid row repeat the gate gradients, each at its own pace): ds out of place
        buf["wide_bwd_mode"] = bool(PERSISTENT_RNN and WIDE_SWEEP_BWD and self.recurrent_dropout == 0 and T >= 2 and not buf["coef_mode"]
                                    and ops.device_exclusive() and ops.rnn_sweep_wide_bwd_supported(rt, B, T, H, 2)
                                    and torch.cuda.get_device_properties(device).multi_processor_count >= 256)
        if buf["wide_bwd_mode"]:
            # the sweep writes the gate-sum gradients as the two bf16 images the layer's products read (round 4: otherwise three image
            # passes over a 0.5 GB f32 ds per direction): straight [B T, 4H] for dX = ds W^T, transposed [4H][K8] with time-major columns
            # for dW = x^T ds and dU = h^T ds; the K padding is zeroed here and never written
            K8 = (B * T + 7) // 8 * 8
            for dd in buf["dirs"]:
                dd["ds16"] = torch.empty(B * T, NS[rt] * H, device=device, dtype=torch.bfloat16)
                dd["ds16T"] = torch.zeros(NS[rt] * H, K8, device=device, dtype=torch.bfloat16)
        return buf

    def final_states(self, buf):
        """[fh, (fc), bh, (bc)] as strided [B,H] views (las.py:126 order)."""
        T, out = buf["T"], []
        for dd in buf["dirs"]:
            tl = 0 if dd["reverse"] else T - 1
            out.append(dd["hseq"][:, tl])
            if self.rt == "lstm":
                out.append(dd["cseq"][:, tl])
        return out

    def forward(self, buf, x3d, mask, init_states, training, seed, tables_ready=False):
        B, T, H, rt = buf["B"], buf["T"], self.H, self.rt
        nst = 2 if rt == "lstm" else 1
        x2d = x3d.reshape(B * T, self.Din)
        drop = training and self.dropout > 0
        rdrop = training and self.recurrent_dropout > 0
        projections = []
        for d, dd in enumerate(buf["dirs"]):
            p = self.store.p
            dd["rec_mult"] = None
            W, b = p[self.names[d] + "kernel"], p[self.names[d] + "bias"]
            bin_ = b[0] if rt == "gru" else b
            if rdrop:
                # [TF-sem] implementation 1: a mask per gate on the recurrent operand and (if dropout > 0) on the input; gate g's stream ids
                # are the layer's + 128 g (rng.py); the input projection runs gate by gate, each with its own mask table
                ng = NG[rt]
                ops.dropout_tables([(dd["rtab"][g], self.stream_rec + d + 128 * g, self.recurrent_dropout) for g in range(ng)] +
                                   ([(dd["mtab_g"][g], self.stream_in + d + 128 * g, self.dropout) for g in range(ng)] if drop else []), seed)
                dd["rec_mult"] = dd["rtab"]
                pre2 = dd["pre"].view(B * T, -1)
                for g in range(ng):
                    ops.gemm(x2d, W[:, g * H:(g + 1) * H], pre2[:, g * H:(g + 1) * H], bias=bin_[g * H:(g + 1) * H],
                             a_scale=dd["mtab_g"][g] if drop else None, a_rpg=T)
            else:
                def project(dd=dd, W=W, bin_=bin_, d=d):
                    if drop and not tables_ready:
                        ops.dropout_table(dd["mtab"], seed, self.stream_in + d, self.dropout)
                    ops.gemm(x2d, W, dd["pre"].view(B * T, -1), bias=bin_, a_scale=dd["mtab"] if drop else None, a_rpg=T)
                projections.append(project)
            dd["bias_rec"] = b[1] if rt == "gru" else None
            dd["U"] = p[self.names[d] + "recurrent_kernel"]
            if init_states is not None:
                st = init_states[d * nst:(d + 1) * nst]
                dd["h0"] = st[0]
                dd["c0"] = st[1] if rt == "lstm" else None
            else:
                dd["h0"] = dd["c0"] = None
        # the two directions' input projections are independent products of the same x: side by side on two streams (joined in front
        # of the sweep below) they fill each other's prologue and last round.  Direction 1's goes to the side stream FIRST: the side
        # stream starts behind what the main stream holds at that moment, and until the end of round 4 that included direction 0's
        # product - the step's timeline showed the two one after the other (side by side they measure the same: 9.40 / 9.37-9.40 ms per step, each product fills the chip)
        if len(projections) == 2 and Overlap.concurrent and BiRNN.fwd_side.on:
            BiRNN.fwd_side.run(projections[1])
            projections[0]()
        else:
            for project in projections:
                project()
        BiRNN.fwd_side.join()
        buf["mask"] = mask
        buf["x3d"] = x3d
        buf["drop"] = drop
        buf["rdrop"] = rdrop
        coef = bool(buf.get("coef_mode")) and training
        for dd in buf["dirs"]:
            dd["coef"] = dd["coef_buf"] if coef else None
        buf["coef_fwd"] = coef
        seq_dirs = [dict(dd, saved=None) for dd in buf["dirs"]] if coef else buf["dirs"]     # (coefficients replace the saved activations)
        buf["seq"] = ops.make_rnn_seq(rt, B, T, H, seq_dirs, mask, buf["y"], [0, H])
        # one persistent launch for all T steps when the layer fits on the chip, else one launch per step
        if PERSISTENT_RNN and not rdrop and ops.rnn_persist_supported(rt, B, T, H, 2):
            if "persist_ws" not in buf:
                buf["persist_ws"] = ops.rnn_persist_ws(B, H, 2, x3d.device)
            ops.rnn_seq_fwd_persist(buf["seq"], buf["persist_ws"], getattr(self.store, "err_flag", None))
        elif PERSISTENT_RNN and WIDE_SWEEP and not rdrop and ops.device_exclusive() and ops.rnn_sweep_wide_supported(rt, B, T, H, 2) and \
                torch.cuda.get_device_properties(x3d.device).multi_processor_count >= H // 4:
            if "wide_ws" not in buf:
                buf["wide_ws"] = ops.rnn_sweep_wide_ws(B, H, 2, x3d.device)
            ops.rnn_sweep_wide_fwd(buf["seq"], buf["wide_ws"], getattr(self.store, "err_flag", None))
        else:
            ops.rnn_seq_fwd(buf["seq"])
        return buf["y"]

    def backward(self, buf, dy3d, dfinal_h, dc_bufs, dx3d, dx_accumulate=False, side=None, overlap=None):
        """dy3d [B,T,2H]; dfinal_h: per direction gradient wrt the final h state ([B,H] or None);
        dc_bufs: per direction [B,H] buffer holding the gradient wrt the final c state on entry and
        the gradient wrt the initial c state on exit (LSTM).  Returns per-direction dh0 buffers.
        Parameter gradients are accumulated into the store; dx3d (+)= input gradient if not None.
        side: a SideStream - the weight gradients (which read only this layer's own buffers) then run on it,
        beside whatever the caller enqueues next; the caller join()s it.
        overlap: an Overlap - the work it holds (the previous layer's weight gradients) is released beside THIS layer's
        backward sweep, and this layer's weight gradients are handed to it for the next sweep; the caller flush()es it."""
        B, T, H, rt = buf["B"], buf["T"], self.H, self.rt
        pws = None
        if buf.get("coef_fwd"):                          # the forward sweep left the coefficients the BPTT sweep reads
            if "persist_bwd_ws" not in buf:
                buf["persist_bwd_ws"] = ops.rnn_persist_bwd_ws(B, H, 2, dy3d.device)
            pws = buf["persist_bwd_ws"]
        wide = bool(buf.get("wide_bwd_mode")) and not buf["rdrop"] and pws is None and ops.mixed_precision()
        if wide and "wide_bwd_ws" not in buf:
            buf["wide_bwd_ws"] = ops.rnn_sweep_wide_bwd_ws(B, H, 2, dy3d.device)
        gds = []
        for d, dd in enumerate(buf["dirs"]):
            if buf["mask"] is not None:
                ops.fill(dd["dy_carry"], 0.0)
            gb = self.store.g[self.names[d] + "bias"]
            gds.append(dict(dh_last=dfinal_h[d], dc=dc_bufs[d] if rt == "lstm" else None,
                            dy_carry=dd["dy_carry"] if buf["mask"] is not None else None, direct=dd["direct"], dh0=dd["dh0"],
                            ds=dd["ds"] if pws is not None else None,
                            ds16=dd["ds16"] if wide else None, ds16T=dd["ds16T"] if wide else None,
                            # the BPTT sweeps sum the bias gradients themselves (no second pass over ds)
                            db=(gb[0] if rt == "gru" else gb) if (pws is not None or wide) else None,
                            db_rec=gb[1] if (rt == "gru" and pws is not None) else None))
        dskey = "ds" if pws is not None else "saved"               # where this backward pass leaves the f32 gate-sum gradients (wide: bf16 images only)
        if wide:
            sweep = lambda: ops.rnn_sweep_wide_bwd(buf["seq"], dy3d, gds, buf["wide_bwd_ws"], getattr(self.store, "err_flag", None))
        else:
            sweep = lambda: ops.rnn_seq_bwd(buf["seq"], dy3d, gds, pws, getattr(self.store, "err_flag", None) if pws is not None else None)
        if overlap is not None:
            overlap.beside(sweep, ops.sweep_diag_words(pws) if pws is not None else None)
        else:
            sweep()
        x2d = buf["x3d"].reshape(B * T, self.Din)
        g, p = self.store.g, self.store.p

        # recurrent dropout (Keras implementation 1): (ds slot, kernel column block, mask index) of every gate, input side and recurrent side
        gate_in = [(k, k, k) for k in range(NG[rt])]
        gate_rec = [(0, 0, 0), (1, 1, 1), (3, 2, 2)] if rt == "gru" else [(k, k, k) for k in range(NG[rt])]

        def param_grads_per_gate():     # the same sums with one dropout mask per gate on x and on h_prev
            for d, dd in enumerate(buf["dirs"]):
                nm = self.names[d]
                ds3 = dd[dskey]
                ds2 = ds3.view(B * T, -1)
                gW, gU, gb = g[nm + "kernel"], g[nm + "recurrent_kernel"], g[nm + "bias"]
                hs = dd["hseq"]
                for sl, cb, mi in gate_in:
                    dsv = ds2[:, sl * H:(sl + 1) * H]
                    ops.gemm(x2d, dsv, gW[:, cb * H:(cb + 1) * H], trans_a=True, accumulate=1, split_k=auto_split_k(self.Din, H, B * T),
                             a_scale=dd["mtab_g"][mi] if buf["drop"] else None, a_rpg=T)
                    ops.colsum(dsv, (gb[0] if rt == "gru" else gb)[cb * H:(cb + 1) * H])
                for sl, cb, mi in gate_rec:
                    dsg = ds3[:, :, sl * H:(sl + 1) * H]
                    gUg = gU[:, cb * H:(cb + 1) * H]
                    rt_ = dd["rtab"][mi]
                    if T > 1:
                        if dd["reverse"]:
                            ops.gemm(hs[:, 1:], dsg[:, :T - 1], gUg, trans_a=True, accumulate=1, a_scale=rt_, a_rpg=T, a_scale_stride=H)
                        else:
                            ops.gemm(hs[:, :T - 1], dsg[:, 1:], gUg, trans_a=True, accumulate=1, a_scale=rt_, a_rpg=T, a_scale_stride=H)
                    if dd["h0"] is not None:
                        t0 = T - 1 if dd["reverse"] else 0
                        ops.gemm(dd["h0"], dsg[:, t0], gUg, trans_a=True, accumulate=1, a_scale=rt_, a_rpg=1)
                    if rt == "gru":
                        ops.colsum(ds2[:, sl * H:(sl + 1) * H], gb[1][cb * H:(cb + 1) * H])

        def param_grads_wide(only=None):
            """dW = x^T ds, dU = h_prev^T ds from the transposed bf16 image the wide sweep wrote (time-major columns t B + b): x and the
            shifted h take the same column order (ops.f32_to_bf16_image_tb); the initial state fills the one column block of the h image
            that has no predecessor inside the sequence, so it needs no product of its own."""
            x3 = buf["x3d"]
            for d, dd in enumerate(buf["dirs"]):
                if only is not None and d != only:
                    continue
                nm = self.names[d]
                mt = dd["mtab"] if buf["drop"] else None
                dsT = dd["ds16T"]
                xT = ops.f32_to_bf16_image_tb(x3, ops.image_scratch("xT_tb", self.Din, B * T, layout=(B, T)), scale=mt)
                ops.gemm_bf16_nt(xT, dsT, g[nm + "kernel"], accumulate=1)
                hs = dd["hseq"]
                # (a scratch per layer and direction: the column block of the initial state is written by the layers that have one only)
                hT = ops.image_scratch("hT_tb", H, B * T, layout=(B, T, self.prefix, d))
                if T > 1:
                    if dd["reverse"]:
                        ops.f32_to_bf16_image_tb(hs[:, 1:], hT, dst_shift=0)
                    else:
                        ops.f32_to_bf16_image_tb(hs[:, :T - 1], hT, dst_shift=B)
                if dd["h0"] is not None:
                    ops.f32_to_bf16_image_tb(dd["h0"].unsqueeze(1), hT, dst_shift=(T - 1) * B if dd["reverse"] else 0)
                ops.gemm_bf16_nt(hT, dsT, g[nm + "recurrent_kernel"], accumulate=1)

        def param_grads(only=None):      # reads x, ds, hseq, h0, the dropout table of THIS layer only; writes this layer's gradients
            if buf["rdrop"]:
                return param_grads_per_gate()
            if wide:
                return param_grads_wide(only)
            shared = rt == "lstm" and T > 1 and ops.bf16_images_pay(H, 4 * H, B * (T - 1)) and ops.bf16_images_pay(self.Din, 4 * H, B * T)
            for d, dd in enumerate(buf["dirs"]):
                if only is not None and d != only:
                    continue
                nm = self.names[d]
                ds3 = dd[dskey]
                ds2 = ds3.view(B * T, -1)
                mt = dd["mtab"] if buf["drop"] else None
                if shared:
                    # wide layers under mixed precision: dW = x^T ds and dU = h_prev^T ds contract over the same B T rows of ds - ONE
                    # transposed bf16 image of ds serves both (ops.gemm alone would transpose ds twice: 0.23 ms each at las_large);
                    # h is written one column off inside every clip (h[b, t -+ 1] beside ds[b, t]), the column without a predecessor
                    # stays zero (a scratch of its own per direction: nothing else writes it)
                    dsT = ops.f32_to_bf16_image(ds2, ops.image_scratch("dsT", 4 * H, B * T), transpose=True)
                    xT = ops.f32_to_bf16_image(x2d, ops.image_scratch("a", self.Din, B * T), transpose=True, scale=mt, rows_per_group=T)
                    ops.gemm_bf16_nt(xT, dsT, g[nm + "kernel"], accumulate=1)
                    if pws is None:
                        ops.colsum(ds2, g[nm + "bias"])
                    hs = dd["hseq"]
                    hT = ops.image_scratch("hT_r" if dd["reverse"] else "hT_f", H, B * T, layout=(B, T))
                    if dd["reverse"]:
                        ops.f32_to_bf16_image(hs[:, 1:], hT, transpose=True, dst_rows_per_batch=T, dst_shift=0)
                    else:
                        ops.f32_to_bf16_image(hs[:, :T - 1], hT, transpose=True, dst_rows_per_batch=T, dst_shift=1)
                    ops.gemm_bf16_nt(hT, dsT, g[nm + "recurrent_kernel"], accumulate=1)
                    if dd["h0"] is not None:
                        t0 = T - 1 if dd["reverse"] else 0
                        ops.gemm(dd["h0"], ds3[:, t0], g[nm + "recurrent_kernel"], trans_a=True, accumulate=1)
                    continue
                cell_param_grads(rt, H, x2d, None, ds2, g[nm + "kernel"], None, None if pws is not None else g[nm + "bias"], a_scale=mt, a_rpg=T)
                # recurrent kernel: sum_t h_{prev(t)}^T ds_t with the sequence shifted by one processing step
                hs = dd["hseq"]
                for s0, d0, n in slot_cols(rt, H, "rec"):
                    gU = g[nm + "recurrent_kernel"][:, d0:d0 + n]
                    if T > 1:
                        if dd["reverse"]:
                            ops.gemm(hs[:, 1:], ds3[:, :T - 1, s0:s0 + n], gU, trans_a=True, accumulate=1)
                        else:
                            ops.gemm(hs[:, :T - 1], ds3[:, 1:, s0:s0 + n], gU, trans_a=True, accumulate=1)
                    if dd["h0"] is not None:
                        t0 = T - 1 if dd["reverse"] else 0
                        ops.gemm(dd["h0"], ds3[:, t0, s0:s0 + n], gU, trans_a=True, accumulate=1)

        if overlap is not None and overlap.mode == "concurrent" and overlap.on and not buf["rdrop"]:
            for d in range(len(buf["dirs"])):                      # the two directions' products are independent: one lane each
                overlap.defer(lambda d=d: param_grads(d))
        elif overlap is not None:
            overlap.defer(param_grads)
        elif side is not None:
            side.run(param_grads)
        else:
            param_grads()
        if dx3d is not None and buf["rdrop"]:                     # one input mask per gate: dx = sum_g (ds_g W_g^T) (.) mask_g
            for d, dd in enumerate(buf["dirs"]):
                ds2 = dd[dskey].view(B * T, -1)
                W = p[self.names[d] + "kernel"]
                for k, (sl, cb, mi) in enumerate(gate_in):
                    ops.gemm(ds2[:, sl * H:(sl + 1) * H], W[:, cb * H:(cb + 1) * H], dx3d.view(B * T, self.Din), trans_b=True,
                             accumulate=1 if (dx_accumulate or d == 1 or k > 0) else 0, c_scale=dd["mtab_g"][mi] if buf["drop"] else None, c_rpg=T)
        elif dx3d is not None and wide:
            for d, dd in enumerate(buf["dirs"]):                    # dX (+)= ds W^T straight from the sweep's bf16 image of ds
                mt = dd["mtab"] if buf["drop"] else None
                W = p[self.names[d] + "kernel"]
                W16 = ops.f32_to_bf16_image(W, ops.image_scratch("b", self.Din, NG[rt] * H))
                ops.gemm_bf16_nt(dd["ds16"], W16, dx3d.view(B * T, self.Din), accumulate=1 if (dx_accumulate or d == 1) else 0, c_scale=mt, c_rpg=T)
        elif dx3d is not None:
            for d, dd in enumerate(buf["dirs"]):
                mt = dd["mtab"] if buf["drop"] else None
                cell_input_grad(rt, H, dd[dskey].view(B * T, -1), p[self.names[d] + "kernel"], dx3d.view(B * T, self.Din),
                                accumulate=(dx_accumulate or d == 1), c_scale=mt, c_rpg=T)
        return [dd["dh0"] for dd in buf["dirs"]]


BiRNN.fwd_side = SideStream("fwd", default_on=Overlap.concurrent)
