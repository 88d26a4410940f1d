// placeholder until the MFMA attention lands (replaced in a later commit)
#include "afm_common.h"
int afm_attn_fwd_mfma_try(const afm_attn_shape* s, const void* Q, const void* K, const void* V,
                          void* O, float* lse, hipStream_t st) { return AFM_ERR_UNSUPPORTED; }
int afm_attn_bwd_mfma_try(const afm_attn_shape* s, const void* Q, const void* K, const void* V,
                          const void* O, const void* dO, const float* lse, float* delta, void* dQ,
                          void* dK, void* dV, int lddq, int lddk, int lddv, hipStream_t st) { return AFM_ERR_UNSUPPORTED; }
