// gemm_pb.hip -- dpp_gemm's LDS-tiled, K-split and 16-column-stream kernels instantiated for bf16 MFMA operands (dpp_gemm_desc.precision = 1,
// BASELINE config 5; round 6): a translation unit of its own.  The operands may be float32 or bf16-stored (the ST = true forms take both: the
// storage flags are tested at run time there), so ONE set of instantiations serves the bf16 mode.
#include "gemm_kernels.h"

int dpp_gemm_dispatch_pb(GemmArgs& ga, int bm, int bn, int wm, hipStream_t st) { return gemm_dispatch<true, true>(ga, bm, bn, wm, st); }
