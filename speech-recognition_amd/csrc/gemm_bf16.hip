// The bf16-operand instantiations of the GEMM kernels (--mixed-precision): gemm.hip compiled with GEMM_BF = 1, as its own
// translation unit so that the two halves of the library's longest compile run in parallel.
#define GEMM_BF 1
#include "gemm.hip"
