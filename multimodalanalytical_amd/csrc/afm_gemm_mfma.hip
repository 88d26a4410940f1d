// placeholder until the MFMA GEMM lands (replaced in the next commit)
#include "afm_common.h"
int afm_gemm_mfma_try(const afm_gemm_desc* d, hipStream_t st) { (void)d; (void)st; return AFM_ERR_UNSUPPORTED; }
