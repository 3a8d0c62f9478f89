// dec_gemv_wide.h -- decode-step skinny GEMM for K = 6144 (1.7B preset down-projection), see dec_gemv_wide.hip.
#pragma once
#include "dec_kernels.h"

namespace qasr {

// true where decode_gemv_wide_launch has an instantiation: fragment-major weights present, K = 6144, plain / residual epilogue
bool decode_gemv_wide_supported(DecEpi epi, const DecGemvArgs& a);
// out = epi(X . W^T), same contract as decode_gemv_fused_launch without a norm; returns the number of column tiles
int decode_gemv_wide_launch(DecEpi epi, const DecGemvArgs& a, hipStream_t s);

}  // namespace qasr
