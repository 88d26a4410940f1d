// bf16 instantiation of the single-pass MFMA attention kernels
#include "afm_attn_mfma_impl.h"
