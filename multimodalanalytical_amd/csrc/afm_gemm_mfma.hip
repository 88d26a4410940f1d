// bf16 instantiation of the single-pass MFMA GEMMs
#include "afm_gemm_mfma_impl.h"
