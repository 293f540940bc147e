// f32 GEMM whose products run on the bf16 matrix pipe as nine exact bf16 pair products per element pair (gemm_core.h run_split<9>):
// gemm.hip compiled with GEMM_BF = 2, as a translation unit of its own (parallel build).
#define GEMM_BF 2
#include "gemm.hip"
