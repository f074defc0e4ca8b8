// gemm_st.hip -- dpp_gemm's kernels instantiated for bf16-STORED operands / outputs (ABI v9, DPP_ST_*): a translation unit of its own,
// so that the float32 instantiations of gemm.hip carry none of the run-time type tests and the two sets compile in parallel.
#include "gemm_kernels.h"

int dpp_gemm_dispatch_st(GemmArgs& ga, int bm, int bn, int wm, hipStream_t st) { return gemm_dispatch<true>(ga, bm, bn, wm, st); }
