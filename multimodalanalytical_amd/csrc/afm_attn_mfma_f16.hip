// fp16 instantiation of the single-pass MFMA attention kernels (AFM_F16 operands)
#define AFM_E16_F16 1
#include "afm_attn_mfma_impl.h"
