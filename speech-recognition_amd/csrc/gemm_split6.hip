// As gemm_split9.hip with the three pairs of weight <= 2^-24 left out (gemm_core.h run_split<6>): gemm.hip compiled with GEMM_BF = 3.
#define GEMM_BF 3
#include "gemm.hip"
