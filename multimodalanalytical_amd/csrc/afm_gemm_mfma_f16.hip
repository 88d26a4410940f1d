// fp16 instantiation of the single-pass MFMA GEMMs (AFM_F16 operands)
#define AFM_E16_F16 1
#include "afm_gemm_mfma_impl.h"
