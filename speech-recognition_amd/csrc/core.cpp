// Runtime plumbing of libasr_mi355x: runtime-error hand-over and the ABI struct sizes.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <hip/hip_runtime.h>

#include "../../include/asr_mi355x.h"

// (asr_set_error / asr_last_error / asr_version live in status.cpp: plain C++, shared with the sanitizer build of the host parsers)

// Called once by the binding after the library (and the process's HIP runtime) is loaded: reports - and thereby clears - whatever
// error an earlier runtime call of this thread left behind, so that the first ASR_LAUNCH_CHECK does not blame a launch for it.
// Returns the hipError_t it found (0 = none); the entry points themselves never clear the runtime's error state.
extern "C" int asr_runtime_init(void) { return (int)hipGetLastError(); }

// sizeof of every ABI struct, so that a binding (ctypes, cgo, JNI...) can verify its mirror
extern "C" long asr_struct_size(const char* name) {
#define SZ(T) if (!strcmp(name, #T)) return (long)sizeof(T)
  SZ(asr_logmel_cfg);
  SZ(asr_gemm_desc);
  SZ(asr_rnn_geom);
  SZ(asr_rnn_pack_desc);
  SZ(asr_rnn_step_fwd);
  SZ(asr_rnn_back_src);
  SZ(asr_rnn_step_bwd);
  SZ(asr_rnn_seq);
  SZ(asr_rnn_seq_grad);
  SZ(asr_decoder_sweep);
  SZ(asr_decoder_sweep_grad);
  SZ(asr_conv_desc);
  SZ(asr_rowdrop);
  SZ(asr_lr_schedule);
  SZ(asr_audio_info_t);
#undef SZ
  return -1;
}
