"""ctypes binding of libasr_mi355x.so (C ABI declared in include/asr_mi355x.h).

The library is the product: there is no CPU or PyTorch fallback.  If the shared object is
missing or a symbol cannot be resolved, importing the ops raises immediately.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libasr_mi355x.so")

RNN_TYPES = {"lstm": 0, "gru": 1, "rnn": 2}
RNN_MAXSEG = 3

c_f32p = C.c_void_p  # device pointers travel as raw addresses
c_long = C.c_long


class LogmelCfg(C.Structure):
    _fields_ = [("sample_rate", C.c_int), ("frame_length", C.c_int), ("frame_step", C.c_int), ("fft_length", C.c_int),
                ("num_mel_bins", C.c_int), ("lower_edge_hertz", C.c_float), ("upper_edge_hertz", C.c_float),
                ("epsilon", C.c_float), ("use_delta", C.c_int), ("sa_enable", C.c_int), ("sa_F", C.c_int),
                ("sa_mF", C.c_int), ("sa_T", C.c_int), ("sa_mT", C.c_int), ("sa_p", C.c_float),
                ("feature_type", C.c_int), ("num_mfcc", C.c_int)]


class GemmDesc(C.Structure):
    _fields_ = [("trans_a", C.c_int), ("trans_b", C.c_int), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("batch", C.c_int), ("split_k", C.c_int), ("lda", c_long), ("ldb", c_long), ("ldc", c_long), ("stride_a", c_long),
                ("stride_b", c_long), ("stride_c", c_long), ("stride_a_scale", c_long), ("alpha", C.c_float),
                ("accumulate", C.c_int), ("relu", C.c_int), ("bias", c_f32p), ("a_scale", c_f32p), ("a_rpg", C.c_int),
                ("c_scale", c_f32p), ("c_rpg", C.c_int), ("compute", C.c_int)]


class RnnGeom(C.Structure):
    _fields_ = [("Q", C.c_int), ("KSt", C.c_int), ("ks0", C.c_int * RNN_MAXSEG), ("wp_floats", c_long)]


class RnnPackDesc(C.Structure):
    _fields_ = [("rnn_type", C.c_int), ("H", C.c_int), ("nseg", C.c_int), ("W", c_f32p * RNN_MAXSEG), ("ldw", c_long * RNN_MAXSEG),
                ("K", C.c_int * RNN_MAXSEG), ("is_rec", C.c_int * RNN_MAXSEG), ("Wp", c_f32p)]


class RnnStepFwd(C.Structure):
    _fields_ = [("nseg", C.c_int), ("KSt", C.c_int), ("Wp", c_f32p), ("Wp16", c_f32p),
                ("seg_x", c_f32p * RNN_MAXSEG), ("seg_ld", c_long * RNN_MAXSEG),
                ("seg_K", C.c_int * RNN_MAXSEG), ("seg_ks0", C.c_int * RNN_MAXSEG),
                ("seg_drop_rate", C.c_float * RNN_MAXSEG), ("seg_drop_stream", C.c_uint32 * RNN_MAXSEG),
                ("seg_drop_ld", c_long * RNN_MAXSEG), ("seg_drop_off", C.c_int * RNN_MAXSEG),
                ("pre", c_f32p), ("pre_ld", c_long), ("bias", c_f32p), ("bias_rec", c_f32p),
                ("h_prev", c_f32p), ("h_prev_ld", c_long), ("c_prev", c_f32p), ("c_prev_ld", c_long),
                ("y_prev", c_f32p), ("y_prev_ld", c_long), ("mask", c_f32p), ("mask_ld", c_long),
                ("h_out", c_f32p), ("h_out_ld", c_long), ("c_out", c_f32p), ("c_out_ld", c_long),
                ("y_out", c_f32p), ("y_out_ld", c_long), ("saved", c_f32p), ("saved_ld", c_long)]


class RnnBackSrc(C.Structure):
    _fields_ = [("D", c_f32p), ("ldd", c_long), ("W", c_f32p), ("ldw", c_long), ("W16", c_f32p), ("nseg", C.c_int), ("d_col0", C.c_int * 2),
                ("w_col0", C.c_int * 2), ("len", C.c_int * 2), ("drop_rate", C.c_float), ("drop_stream", C.c_uint32),
                ("drop_ld", c_long), ("drop_off", C.c_int)]


class RnnStepBwd(C.Structure):
    _fields_ = [("n_units", C.c_int), ("srcA", RnnBackSrc), ("srcB", RnnBackSrc),
                ("addA", c_f32p), ("addA_ld", c_long), ("addB", c_f32p), ("addB_ld", c_long),
                ("direct", c_f32p), ("direct_ld", c_long), ("out", c_f32p), ("out_ld", c_long),
                ("dc", c_f32p), ("dc_ld", c_long), ("dy_carry", c_f32p), ("dy_carry_ld", c_long),
                ("mask", c_f32p), ("mask_ld", c_long), ("saved", c_f32p), ("saved_ld", c_long),
                ("h_prev", c_f32p), ("h_prev_ld", c_long), ("c_prev", c_f32p), ("c_prev_ld", c_long),
                ("c_out", c_f32p), ("c_out_ld", c_long), ("dslots", c_f32p), ("dslots_ld", c_long)]


class RnnSeq(C.Structure):
    _fields_ = [("rnn_type", C.c_int), ("B", C.c_int), ("T", C.c_int), ("H", C.c_int), ("ndir", C.c_int),
                ("reverse", C.c_int * 2), ("pre", c_f32p * 2), ("Wp", c_f32p * 2), ("U", c_f32p * 2), ("ldu", c_long * 2),
                ("Wp16", c_f32p * 2), ("U16", c_f32p * 2),
                ("bias_rec", c_f32p * 2), ("h0", c_f32p * 2), ("h0_ld", c_long * 2), ("c0", c_f32p * 2),
                ("c0_ld", c_long * 2), ("rec_mult", c_f32p * 2), ("mask", c_f32p),
                ("hseq", c_f32p * 2), ("cseq", c_f32p * 2), ("y", c_f32p), ("y_ld", c_long), ("y_col", C.c_int * 2),
                ("saved", c_f32p * 2), ("coef", c_f32p * 2)]


class DecoderSweep(C.Structure):
    _fields_ = [("B", C.c_int), ("U", C.c_int), ("T2", C.c_int), ("Hd", C.c_int), ("D", C.c_int),
                ("Kq", c_f32p), ("enc", c_f32p), ("s0", c_f32p), ("mask", c_f32p), ("h_init", c_f32p), ("c_init", c_f32p),
                ("Wp0", c_f32p), ("KSt0", C.c_int), ("ks0_ctx", C.c_int), ("ks0_h", C.c_int),
                ("Wp1", c_f32p), ("KSt1", C.c_int), ("ks1_x", C.c_int), ("ks1_h", C.c_int),
                ("pre0", c_f32p), ("bias1", c_f32p), ("tokmask", c_f32p),
                ("seed", c_f32p), ("drop_rate", C.c_float), ("drop_stream0", C.c_uint32), ("drop_stream_step", C.c_uint32),
                ("p", c_f32p), ("ctx", c_f32p), ("hin", c_f32p), ("cin", c_f32p),
                ("y0", c_f32p), ("saved0", c_f32p), ("h0", c_f32p), ("c0", c_f32p), ("y1", c_f32p), ("saved1", c_f32p)]


class DecoderSweepGrad(C.Structure):
    _fields_ = [("B", C.c_int), ("U", C.c_int), ("T2", C.c_int), ("Hd", C.c_int), ("D", C.c_int),
                ("Kq", c_f32p), ("enc", c_f32p), ("p", c_f32p), ("ctx", c_f32p), ("saved0", c_f32p), ("saved1", c_f32p),
                ("cin", c_f32p), ("c0", c_f32p), ("tokmask", c_f32p), ("dy1", c_f32p), ("dy1_ld", c_long),
                ("U1", c_f32p), ("W1", c_f32p), ("U0", c_f32p), ("W0", c_f32p),
                ("seed", c_f32p), ("drop_rate", C.c_float), ("drop_stream0", C.c_uint32), ("drop_stream_step", C.c_uint32),
                ("ds0", c_f32p), ("ds1", c_f32p), ("de", c_f32p), ("dctx", c_f32p), ("dh_init", c_f32p), ("dc_init", c_f32p),
                ("de_sum", c_f32p)]


class RnnSeqGrad(C.Structure):
    _fields_ = [("dy", c_f32p), ("dy_ld", c_long), ("dh_last", c_f32p * 2), ("dh_last_ld", c_long * 2),
                ("dc", c_f32p * 2), ("dy_carry", c_f32p * 2), ("direct", c_f32p * 2), ("dh0", c_f32p * 2),
                ("dh0_ld", c_long * 2), ("ds", c_f32p * 2), ("db", c_f32p * 2), ("db_rec", c_f32p * 2),
                ("ds16", C.c_void_p * 2), ("ds16T", C.c_void_p * 2), ("ds16T_ld", c_long)]


class ConvDesc(C.Structure):
    _fields_ = [("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C", C.c_int), ("kh", C.c_int), ("kw", C.c_int),
                ("sh", C.c_int), ("sw", C.c_int), ("O", C.c_int)]


class RowDrop(C.Structure):
    _fields_ = [("stream0", C.c_uint32), ("stream_step", C.c_uint32), ("period", C.c_int), ("idx_ld", c_long),
                ("idx_off", C.c_int), ("rate", C.c_float)]


class LrSchedule(C.Structure):
    _fields_ = [("increasing_delta", C.c_float), ("decreasing_delta", C.c_float), ("max_learning_rate", C.c_float),
                ("min_learning_rate", C.c_float), ("warmup_steps", C.c_int), ("offset_steps", C.c_int)]


class AudioInfo(C.Structure):
    _fields_ = [("sample_rate", C.c_int), ("channels", C.c_int), ("bits_per_sample", C.c_int), ("frames", c_long)]


AUDIO_FORMATS = {"wav": 0, "flac": 1, "pcm": 2}

STRUCTS = {"asr_rnn_pack_desc": RnnPackDesc, "asr_logmel_cfg": LogmelCfg, "asr_gemm_desc": GemmDesc, "asr_rnn_geom": RnnGeom, "asr_audio_info_t": AudioInfo,
           "asr_rnn_step_fwd": RnnStepFwd, "asr_rnn_back_src": RnnBackSrc, "asr_rnn_step_bwd": RnnStepBwd, "asr_rnn_seq": RnnSeq,
           "asr_rnn_seq_grad": RnnSeqGrad, "asr_decoder_sweep": DecoderSweep, "asr_decoder_sweep_grad": DecoderSweepGrad, "asr_conv_desc": ConvDesc, "asr_rowdrop": RowDrop,
           "asr_lr_schedule": LrSchedule}

# symbol -> (restype, argtypes); every function declared in include/asr_mi355x.h
_P = C.c_void_p
SIGNATURES = {
    "asr_last_error": (C.c_char_p, []),
    "asr_version": (C.c_int, []),
    "asr_struct_size": (c_long, [C.c_char_p]),
    "asr_runtime_init": (C.c_int, []),
    "asr_logmel_table_sizes": (C.c_int, [C.POINTER(LogmelCfg), C.POINTER(c_long), C.POINTER(c_long), C.POINTER(c_long)]),
    "asr_logmel_build_tables": (C.c_int, [C.POINTER(LogmelCfg), _P, _P, _P]),
    "asr_logmel_features": (C.c_int, [C.POINTER(LogmelCfg), _P, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P, C.c_int, _P]),
    "asr_spec_augment": (C.c_int, [C.POINTER(LogmelCfg), _P, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "asr_delta_accelerate": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "asr_time_warp": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    "asr_gemm_f32": (C.c_int, [C.POINTER(GemmDesc), _P, _P, _P, _P]),
    "asr_gemm_bf16_nt": (C.c_int, [C.POINTER(GemmDesc), _P, _P, _P, _P]),
    "asr_gemm_bf16_config": (C.c_int, [C.c_int]),
    "asr_debug_sweep_trace": (C.c_int, [C.POINTER(C.c_ulonglong), C.c_int]),
    "asr_debug_decoder_trace": (C.c_int, [C.POINTER(C.c_ulonglong), C.c_int]),
    "asr_set_f32_product_mode": (C.c_int, [C.c_int]),
    "asr_comm_available": (C.c_int, []),
    "asr_comm_unique_id": (C.c_int, [_P]),
    "asr_comm_init": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(_P)]),
    "asr_comm_destroy": (C.c_int, [_P]),
    "asr_allreduce_bucket": (C.c_int, [_P, _P, c_long, _P, _P]),
    "asr_f32_to_bf16_image": (C.c_int, [_P, c_long, C.c_int, C.c_int, C.c_int, c_long, _P, C.c_int, C.c_int, _P, c_long, C.c_int, C.c_int, _P]),
    "asr_f32_to_bf16_image_tb": (C.c_int, [_P, c_long, C.c_int, C.c_int, C.c_int, c_long, _P, _P, c_long, c_long, _P]),
    "asr_rnn_geometry": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(RnnGeom)]),
    "asr_rnn_pack": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(_P), C.POINTER(c_long), C.POINTER(C.c_int),
                               C.POINTER(C.c_int), _P, _P]),
    "asr_rnn_pack_many": (C.c_int, [C.c_int, C.POINTER(RnnPackDesc), _P]),
    "asr_dropout_tables": (C.c_int, [C.c_int, C.POINTER(_P), C.POINTER(c_long), C.POINTER(C.c_uint32), C.POINTER(C.c_float), _P, _P]),
    "asr_rnn_cell_fwd": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(RnnStepFwd), _P, _P]),
    "asr_rnn_cell_bwd": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(RnnStepBwd), _P, _P]),
    "asr_f32_to_bf16": (C.c_int, [_P, _P, c_long, _P]),
    "asr_bf16_to_f32": (C.c_int, [_P, _P, c_long, _P]),
    "asr_debug_occupy": (C.c_int, [C.c_int, C.c_int, C.c_int, _P]),
    "asr_debug_stream_memory": (C.c_int, [_P, c_long, C.c_int, C.c_int, _P]),
    "asr_token_mask": (C.c_int, [_P, c_long, C.c_int, _P, c_long, _P]),
    "asr_rnn_seq_fwd": (C.c_int, [C.POINTER(RnnSeq), _P]),
    "asr_rnn_seq_bwd": (C.c_int, [C.POINTER(RnnSeq), C.POINTER(RnnSeqGrad), _P]),
    "asr_rnn_sweep_ws_floats": (c_long, [C.c_int, C.c_int, C.c_int]),
    "asr_rnn_sweep_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "asr_rnn_sweep_fwd": (C.c_int, [C.POINTER(RnnSeq), _P, _P, _P]),
    "asr_rnn_sweep_bwd_ws_floats": (c_long, [C.c_int, C.c_int, C.c_int]),
    "asr_rnn_sweep_bwd_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "asr_rnn_sweep_bwd": (C.c_int, [C.POINTER(RnnSeq), C.POINTER(RnnSeqGrad), _P, _P, _P]),
    "asr_rnn_sweep_set_spin_limit": (None, [C.c_int]),
    "asr_rnn_sweep_spin_limit": (C.c_int, []),
    "asr_sweep_gate": (C.c_int, [_P, C.c_int, _P]),
    "asr_rnn_sweep_wide_supported": (C.c_int, [C.c_int] * 5),
    "asr_rnn_sweep_wide_ws_floats": (c_long, [C.c_int, C.c_int, C.c_int]),
    "asr_rnn_sweep_wide_fwd": (C.c_int, [C.POINTER(RnnSeq), _P, _P, _P]),
    "asr_rnn_sweep_wide_bwd_supported": (C.c_int, [C.c_int] * 5),
    "asr_rnn_sweep_wide_bwd_ws_floats": (c_long, [C.c_int, C.c_int, C.c_int]),
    "asr_rnn_sweep_wide_bwd": (C.c_int, [C.POINTER(RnnSeq), C.POINTER(RnnSeqGrad), _P, _P, _P]),
    "asr_decoder_sweep_supported": (C.c_int, [C.c_int] * 7),
    "asr_decoder_sweep_ws_floats": (c_long, [C.c_int, C.c_int]),
    "asr_decoder_sweep_fwd": (C.c_int, [C.POINTER(DecoderSweep), _P, _P, _P]),
    "asr_decoder_sweep_bwd_supported": (C.c_int, [C.c_int] * 7),
    "asr_decoder_sweep_bwd_ws_floats": (c_long, [C.c_int, C.c_int]),
    "asr_decoder_sweep_bwd": (C.c_int, [C.POINTER(DecoderSweepGrad), _P, _P, _P]),
    "asr_conv2d_out_dims": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "asr_conv2d_fwd": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, C.c_uint32, C.c_float, _P]),
    "asr_conv2d_bwd_filter": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P]),
    "asr_conv2d_bwd_data": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P]),
    "asr_conv2d_halo_workspace": (C.c_long, [C.POINTER(ConvDesc), C.c_int]),
    "asr_conv2d_halo_force": (C.c_int, [C.c_int]),
    "asr_conv2d_fwd_halo": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, C.c_long, _P]),
    "asr_conv2d_bwd_data_halo": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, C.c_long, _P]),
    "asr_fill_f32": (C.c_int, [_P, c_long, C.c_float, _P]),
    "asr_frame_mask": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "asr_colsum": (C.c_int, [_P, C.c_int, C.c_int, c_long, _P, _P]),
    "asr_colsum_weighted": (C.c_int, [_P, C.c_int, C.c_int, c_long, _P, _P, _P]),
    "asr_rowdot": (C.c_int, [_P, C.c_int, C.c_int, c_long, _P, _P, _P]),
    "asr_rank1_add": (C.c_int, [_P, C.c_int, C.c_int, c_long, _P, _P, _P]),
    "asr_bn_fwd": (C.c_int, [_P, C.c_int, C.c_int, c_long, _P, _P, C.c_float, C.c_float, C.c_int, C.c_int, _P, c_long,
                             _P, _P, _P, _P, _P, _P]),
    "asr_bn_bwd": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, c_long, c_long, c_long, _P, _P, _P, C.c_int, _P, c_long, _P,
                             _P, _P, _P]),
    "asr_dropout_rows": (C.c_int, [_P, c_long, _P, c_long, C.c_int, C.c_int, _P, C.c_uint32, C.c_uint32, C.c_int,
                                   c_long, C.c_int, C.c_float, _P]),
    "asr_dropout_flat": (C.c_int, [_P, c_long, _P, C.c_uint32, C.c_float, _P]),
    "asr_dropout_table": (C.c_int, [_P, c_long, _P, C.c_uint32, C.c_float, _P]),
    "asr_embedding": (C.c_int, [C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, _P, c_long, _P, C.POINTER(RowDrop),
                                C.POINTER(RowDrop), _P]),
    "asr_argmax_rows": (C.c_int, [_P, c_long, C.c_int, C.c_int, _P, _P]),
    "asr_attn_step_fwd": (C.c_int, [_P, c_long, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, c_long, _P]),
    "asr_attn_step_fwd_bf16": (C.c_int, [_P, c_long, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, c_long, _P]),
    "asr_attn_step_bwd_bf16": (C.c_int, [_P, c_long, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, c_long, C.c_int, _P]),
    "asr_attn_step_bwd": (C.c_int, [_P, c_long, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, c_long,
                                    C.c_int, _P]),
    "asr_softmax_xent": (C.c_int, [_P, c_long, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_float, _P]),
    "asr_lr_schedule_init": (C.c_int, [C.POINTER(LrSchedule), c_long, C.c_double, C.c_double, C.c_double, c_long, c_long]),
    "asr_adam_step": (C.c_int, [_P, _P, _P, _P, c_long, _P, C.POINTER(LrSchedule), C.c_float, C.c_float, C.c_float,
                                C.c_float, _P, _P]),
    "asr_advance_state": (C.c_int, [_P, _P, _P]),
    "asr_ctc_workspace_floats": (c_long, [C.c_int, C.c_int, C.c_int]),
    "asr_ctc_loss": (C.c_int, [_P, c_long, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, C.c_int,
                               C.c_float, _P]),
    "asr_mask_rows": (C.c_int, [_P, c_long, _P, C.c_int, C.c_int, _P, c_long, _P]),
    "asr_attn_fused_ws_floats": (c_long, [C.c_int, C.c_int, C.c_int]),
    "asr_attn_fused_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "asr_attn_fused_fwd": (C.c_int, [_P, c_long, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, c_long, _P]),
    "asr_attn_fused_bwd": (C.c_int, [_P, c_long, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, c_long, C.c_int, _P]),
    "asr_greedy_update": (C.c_int, [_P, c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P]),
    "asr_ctc_greedy": (C.c_int, [_P, c_long, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P]),
    "asr_beam_topk": (C.c_int, [_P, c_long, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "asr_beam_select": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, _P, _P, _P, _P, _P, _P, _P, _P,
                                  _P, _P, _P, _P]),
    "asr_ctc_log_softmax": (C.c_int, [_P, c_long, c_long, C.c_int, C.c_int, _P, _P]),
    "asr_ctc_beam_search": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P, _P, _P, C.c_int]),
    "asr_audio_info": (C.c_int, [C.c_char_p, c_long, C.c_int, C.POINTER(AudioInfo)]),
    "asr_audio_decode": (C.c_int, [C.c_char_p, c_long, C.c_int, _P, c_long, C.POINTER(c_long)]),
    "asr_crc32c": (C.c_uint32, [C.c_char_p, c_long, C.c_uint32]),
}

_lib = None


class AsrError(RuntimeError):
    pass


def load():
    """Load the HIP library (once). Fails loudly: no fallback path exists."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C speech-recognition_amd/csrc`). speech_recognition_amd has no CPU fallback.")
    # PyTorch first: it ships its own HIP runtime (libamdhip64) and hands us its streams and allocations, so that copy must be
    # the one this process binds - loading the library before torch would pull in the system runtime instead, and every launch
    # then fails with "no ROCm-capable device is detected"
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    for cname, cls in STRUCTS.items():
        n = lib.asr_struct_size(cname.encode())
        if n != C.sizeof(cls):
            raise ImportError(f"ABI mismatch: sizeof({cname}) is {n} in the library, {C.sizeof(cls)} in the binding")
    lib.asr_runtime_init()               # drops a stale hipErrorNoDevice once; ASR_CHECK itself never touches the runtime's error state
    _lib = lib
    return lib


_EXC = {-1: ValueError, -2: ValueError, -3: ValueError, -4: AsrError}


def check(rc):
    if rc != 0:
        msg = load().asr_last_error().decode()
        raise _EXC.get(rc, AsrError)(msg)
