// gemm_args.h -- what dpp_gemm's kernels get: the caller's descriptor plus the alignment / geometry facts gemm_prepare derives from it.
// Shared by gemm.hip (generic tile, K-split, row-stream and 16-column stream kernels; the entry points) and gemm_expand.hip (variant 4).
#pragma once
#include "dpp_common.h"

struct GemmArgs {
    dpp_gemm_desc d;
    int vecA, vecB;   // float4 loads legal for the operand
    int Kper;         // K-slice length per blockIdx.z (multiple of the chunk depth)
    int bk;           // chunk depth for K-contiguous A: 16 or 32 (both-MN-contiguous layout always uses 64)
    int wide;         // epilogue goes through an LDS image of the tile and touches C / residual / bn_x with 16-B accesses
    int shA, shB;     // 1: the operand's elements are bf16 (DPP_ST_A / DPP_ST_B): a `const float*` cursor advances by (element offset >> 1)
    unsigned long long* prof;   // phase stamps (profiling build only, see dpp_stamp)
};

// dpp_gemm variant 4 (gemm_expand.hip): rows per wave for this problem (0: the kernel does not take it), and its launch
int dpp_gemm_expand_rows(const dpp_gemm_desc& d, const GemmArgs& ga);
int dpp_gemm_expand_launch(const GemmArgs& ga, int rpw, hipStream_t st);
// variants 0-3 on bf16-stored tensors (gemm_st.hip)
int dpp_gemm_dispatch_st(GemmArgs& ga, int bm, int bn, int wm, hipStream_t st);
// variants 0, 2, 3 with bf16 MFMA operands, float32 or bf16-stored tensors (gemm_pb.hip)
int dpp_gemm_dispatch_pb(GemmArgs& ga, int bm, int bn, int wm, hipStream_t st);
