// gemm.hip -- generic f32 MFMA GEMM for gfx950 (see dpp_gemm in include/dpp_hip.h).
//
// One kernel family covers every GEMM-shaped op of the hot path: 1x1 ConvLayer forward / data
// gradient / filter gradient (pixel rows, optional stride-2 row maps) and HiddenLayer forward / data
// gradient / weight gradient, with the pre-activation BatchNorm+ReLU fused into the operand staging and
// bias + residual fused into the epilogue.
//
// Structure (CDNA4): 256 threads = 4 wave64; block tile BM x BN, K walked in chunks of 16 through LDS;
// each wave owns RM x CN tiles of 16x16 and issues v_mfma_f32_16x16x4_f32 (exact f32, k-ordered fma chain).
//   * K-contiguous operand  -> LDS image [row][16+4]; a lane fetches its 4 k-values with ONE ds_read_b128
//     (lane (i, kq) owns k = 4*kq + t, t = 0..3; the +4 pad makes the 16 rows of a lane group hit 16
//     distinct 16-byte slots of the 256-byte bank row);
//   * MN-contiguous operand -> LDS image [k][rows+4]; a lane fetches with ds_read_b32 (rows+4 == 16 mod 32
//     dwords apart for kq = 0/1, so the two halves of a 32-lane group use disjoint banks).
// Global loads are float4 (16 B/lane) along the contiguous dimension whenever alignment allows.
// f32 MFMA runs at the f32 vector rate (157 TF), so these layers are HBM/L2-bound for the small-channel
// stages; the kernel keeps LDS small (<= 15 KB) to run 8 blocks per CU and hide load latency with TLP.
#include "gemm_kernels.h"

namespace {

// out[i] = sum_z partial[z][i] (+ bias).  Threads are laid out as CB columns x ZL z-lanes: lane zl sums z = zl, zl+ZL, ...
// (a fixed order), then the ZL partial sums are combined through LDS in a fixed order: deterministic, and parallel in z
// when there are many slices of a small output (filter gradients: hundreds of slices of a few thousand elements).
__global__ __launch_bounds__(DPP_THREADS) void reduce_partials_kernel(const float* __restrict__ partial, int nz, int n,
                                                                      const float* __restrict__ bias, int nbias,
                                                                      float* __restrict__ out, int ZL) {
    __shared__ float red[DPP_THREADS];
    const int CB = DPP_THREADS / ZL;
    const int col = threadIdx.x % CB, zl = threadIdx.x / CB;
    for (int i0 = blockIdx.x * CB; i0 < n; i0 += gridDim.x * CB) {
        int i = i0 + col;
        float s = 0.0f;
        if (i < n)
            for (int z = zl; z < nz; z += ZL) s += partial[(size_t)z * n + i];
        if (ZL > 1) {
            red[threadIdx.x] = s;
            __syncthreads();
            if (zl == 0) {
                for (int j = 1; j < ZL; ++j) s += red[j * CB + col];
            }
        }
        if (zl == 0 && i < n) {
            if (bias) s += bias[i % nbias];
            out[i] = s;
        }
        if (ZL > 1) __syncthreads();
    }
}

// Many independent partial reductions in ONE launch: job j sums nz slices of n floats into out (the filter / bias gradient
// partials of the whole backward pass).  A workgroup finds its job from the block-offset table, then works exactly like
// reduce_partials_kernel with 16 z-lanes x 16 columns.
struct ReduceJob {
    const float* partial;
    float* out;
    int nz, n, block0, pad;
};

constexpr int RM_COLS = 64;    // columns per workgroup of reduce_multi_kernel (dpp_reduce_multi_block_cols)

// 16 z-lanes x 16 column quads per workgroup: a thread sums every 16th slice of its four columns with 16-byte loads (a row of
// 16 quads is one 256-byte segment), four slices in flight, and the z-lanes meet in LDS.  The order of the sum is fixed.
__global__ __launch_bounds__(DPP_THREADS) void reduce_multi_kernel(const ReduceJob* __restrict__ jobs, int njobs) {
    __shared__ __attribute__((aligned(16))) float red[DPP_THREADS * 4];
    __shared__ int s_job;
    if (threadIdx.x == 0) {
        int lo = 0, hi = njobs - 1;                   // last job whose block0 <= blockIdx.x
        while (lo < hi) {
            int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
        }
        s_job = lo;
    }
    __syncthreads();
    const ReduceJob jb = jobs[s_job];
    constexpr int ZL = 16, CQ = DPP_THREADS / ZL;
    const int cq = threadIdx.x % CQ, zl = threadIdx.x / CQ;
    const int i = ((int)blockIdx.x - jb.block0) * RM_COLS + cq * 4;
    const int n = jb.n, nz = jb.nz;
    const bool vec = (n & 3) == 0 && (reinterpret_cast<uintptr_t>(jb.partial) & 15) == 0;
    float s[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) s[u][j] = 0.0f;
    if (i < n) {
        if (vec) {
            const float* p = jb.partial + i;
            int z = zl;
            for (; z + 3 * ZL < nz; z += 4 * ZL) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(p + (size_t)(z + u * ZL) * n);
#pragma unroll
                for (int u = 0; u < 4; ++u) { s[u][0] += v[u].x; s[u][1] += v[u].y; s[u][2] += v[u].z; s[u][3] += v[u].w; }
            }
            for (; z < nz; z += ZL) {
                const float4 v = *reinterpret_cast<const float4*>(p + (size_t)z * n);
                s[0][0] += v.x; s[0][1] += v.y; s[0][2] += v.z; s[0][3] += v.w;
            }
        } else {
            for (int z = zl; z < nz; z += ZL)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (i + j < n) s[0][j] += jb.partial[(size_t)z * n + i + j];
        }
    }
    float4 t;
    t.x = (s[0][0] + s[1][0]) + (s[2][0] + s[3][0]);
    t.y = (s[0][1] + s[1][1]) + (s[2][1] + s[3][1]);
    t.z = (s[0][2] + s[1][2]) + (s[2][2] + s[3][2]);
    t.w = (s[0][3] + s[1][3]) + (s[2][3] + s[3][3]);
    *reinterpret_cast<float4*>(&red[threadIdx.x * 4]) = t;
    __syncthreads();
    if (zl == 0 && i < n) {
        for (int j = 1; j < ZL; ++j) {
            const float4 o = *reinterpret_cast<const float4*>(&red[(j * CQ + cq) * 4]);
            t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
        }
        if (vec && (reinterpret_cast<uintptr_t>(jb.out) & 15) == 0) *reinterpret_cast<float4*>(jb.out + i) = t;
        else {
            const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (i + j < n) jb.out[i + j] = tv[j];
        }
    }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// argument checks and the alignment facts every kernel choice rests on (shared by dpp_gemm and dpp_gemm_variant_rows)
static int gemm_prepare(const dpp_gemm_desc* dp, GemmArgs& ga) {
    if (!dp || !dp->A || !dp->B || dp->M <= 0 || dp->N <= 0 || dp->K <= 0) return DPP_E_BADARG;
    ga.d = *dp;
    ga.prof = dpp_prof_buffer;
    dpp_gemm_desc& d = ga.d;
    if (d.splitk < 1) d.splitk = 1;
    if (d.splitk > 1 && !d.partial) return DPP_E_BADARG;
    if (d.splitk > 1 && (d.epi.stats || d.epi.bn_x)) return DPP_E_BADARG;
    if (d.epi.bn_x && !(d.epi.bn_mean && d.epi.bn_inv_std && d.epi.bn_scale && d.epi.bn_beta && d.epi.bn_partial)) return DPP_E_BADARG;
    if (d.splitk == 1 && !d.C) return DPP_E_BADARG;
    if (d.actA.mode && d.actA.cmod <= 0) return DPP_E_BADARG;
    if (d.actB.mode && d.actB.cmod <= 0) return DPP_E_BADARG;
    ga.vecA = aligned16(d.A) && (d.lda % 4 == 0);
    if (d.actB.mode & 4) return DPP_E_UNSUPPORTED;
    if (d.actA.mode & 4) {
        if (d.actA.mode != 4 || d.variant == 1 || !d.actA.x2 || !d.actA.aux || !d.actA.mean || !d.actA.scale || !d.actA.beta) return DPP_E_BADARG;
        ga.vecA = ga.vecA && aligned16(d.actA.x2) && aligned16(d.actA.out);
        if (d.actA.out && (!d.a_kc || d.splitk != 1 || d.mapA.s != 1)) return DPP_E_UNSUPPORTED;
        if (!(aligned16(d.actA.mean) && aligned16(d.actA.scale) && aligned16(d.actA.beta) && aligned16(d.actA.aux))) return DPP_E_BADARG;
    }
    ga.vecB = aligned16(d.B) && (d.ldb % 4 == 0);
    // bf16-stored operands / outputs (DPP_ST_*): only with whole-quad geometry (element offsets are halved on `const float*` cursors)
    // and through the 16-byte epilogue; the mode-4 operand (two f32 tensors) does not combine with them
    if (d.store & ~(DPP_ST_A | DPP_ST_B | DPP_ST_C | DPP_ST_BNX)) return DPP_E_BADARG;
    if (d.precision != 0 && d.precision != 1) return DPP_E_BADARG;
    // bf16 MFMA operands (round 6: every variant but the row-stream kernel; not with the two-tensor operand, not split over K slices whose
    // partials would each be rounded differently -- they are not: partials are f32 sums, so split-K is fine)
    if (d.precision == 1 && (d.variant == 1 || (d.actA.mode & 4))) return DPP_E_UNSUPPORTED;
    if ((d.store & DPP_ST_A) && (!ga.vecA || (d.actA.mode & 4))) return DPP_E_UNSUPPORTED;
    if ((d.store & DPP_ST_B) && !ga.vecB) return DPP_E_UNSUPPORTED;
    if ((d.store & (DPP_ST_C | DPP_ST_BNX)) && d.splitk != 1) return DPP_E_UNSUPPORTED;
    ga.shA = (d.store & DPP_ST_A) ? 1 : 0;
    ga.shB = (d.store & DPP_ST_B) ? 1 : 0;
    static const bool wide_ok = []() { const char* e = getenv("DPP_GEMM_WIDE_EPILOGUE"); return !(e && e[0] == '0'); }();
    ga.wide = wide_ok && d.splitk == 1 && d.N % 4 == 0 && d.ldc % 4 == 0 && aligned16(d.C) && aligned16(d.residual) &&
              aligned16(d.epi.bn_x) && aligned16(d.bias) && aligned16(d.epi.bn_mean) && aligned16(d.epi.bn_scale) &&
              aligned16(d.epi.bn_beta) && aligned16(d.epi.bn_inv_std);
    if ((d.actA.mode & 2) && !(aligned16(d.actA.mean) && aligned16(d.actA.scale) && aligned16(d.actA.beta))) return DPP_E_BADARG;
    if ((d.actB.mode & 2) && !(aligned16(d.actB.mean) && aligned16(d.actB.scale) && aligned16(d.actB.beta))) return DPP_E_BADARG;
    if ((d.store & (DPP_ST_C | DPP_ST_BNX)) && !ga.wide) return DPP_E_UNSUPPORTED;
    return DPP_OK;
}

}  // namespace

extern "C" int dpp_abi_version(void) { return DPP_ABI_VERSION; }

extern "C" int dpp_gemm_variant_rows(const dpp_gemm_desc* dp) {
    GemmArgs ga;
    if (gemm_prepare(dp, ga) != DPP_OK) return 0;
    if (ga.d.variant == 2) return ksplit_bn(ga.d, ga) ? 32 : 0;
    if (ga.d.variant == 3) return stream16_rows(ga.d, ga);
    if (ga.d.variant == 4) return dpp_gemm_expand_rows(ga.d, ga);
    return 0;
}

extern "C" int dpp_gemm(const dpp_gemm_desc* dp, dpp_stream_t stream) {
    GemmArgs ga;
    const int prep = gemm_prepare(dp, ga);
    if (prep != DPP_OK) return prep;
    dpp_gemm_desc& d = ga.d;
    const bool red_layout = !d.a_kc && !d.b_kc;
    static const int bk64_min_k = []() { const char* e = getenv("DPP_GEMM_BK64_MINK"); return e ? atoi(e) : 128; }();
    ga.bk = (d.K > 16) ? 32 : 16;
    if (d.a_kc && d.K / d.splitk >= bk64_min_k) ga.bk = 64;     // per K slice when split
    const int chunk = red_layout ? 64 : ga.bk;
    int kper = dpp_cdiv(d.K, d.splitk);
    ga.Kper = dpp_cdiv(kper, chunk) * chunk;
    // every requested slice is written (slices beyond K hold zeros), so the caller's reduce over `splitk` slices is exact
    int bm = d.bm, bn = d.bn, wm = d.wm;
    if (bm == 0) {
        if (d.M <= 16) { bm = 16; bn = 64; wm = 1; }
        else if (d.M <= 32) { bm = 32; bn = 64; wm = 1; }
        else {
            wm = 4;
            bn = d.N > 32 ? 64 : (d.N > 16 ? 32 : 16);
            long blocks128 = (long)dpp_cdiv(d.M, 128) * dpp_cdiv(d.N, bn) * d.splitk;
            bm = blocks128 >= 512 ? 128 : 64;
        }
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (d.variant == 4) {
        const int rpw = dpp_gemm_expand_rows(d, ga);
        if (!rpw) return DPP_E_UNSUPPORTED;
        return dpp_gemm_expand_launch(ga, rpw, st);
    }
    if (d.precision == 1) {
        // 8 k-values per lane and chunk: chunks of 32 / 64 (K > 16), or the K = 16 stream kernel (zero upper half)
        if (d.variant == 0 && !red_layout && ga.bk < 32) return DPP_E_UNSUPPORTED;
        return dpp_gemm_dispatch_pb(ga, bm, bn, wm, st);
    }
    return d.store ? dpp_gemm_dispatch_st(ga, bm, bn, wm, st) : gemm_dispatch<false>(ga, bm, bn, wm, st);
}

extern "C" size_t dpp_reduce_job_bytes(void) { return sizeof(ReduceJob); }

extern "C" int dpp_reduce_multi_block_cols(void) { return RM_COLS; }

extern "C" int dpp_reduce_multi(const void* jobs_dev, int njobs, int total_blocks, dpp_stream_t stream) {
    if (!jobs_dev || njobs < 1 || total_blocks < 1) return DPP_E_BADARG;
    DPP_LAUNCH(reduce_multi_kernel, dim3(total_blocks), dim3(DPP_THREADS), 0, static_cast<hipStream_t>(stream),
                       static_cast<const ReduceJob*>(jobs_dev), njobs);
    return dpp_launch_status();
}

extern "C" int dpp_reduce_partials(const float* partial, int nz, int n, const float* bias, int nbias, float* out,
                                   dpp_stream_t stream) {
    if (!partial || !out || nz < 1 || n < 1) return DPP_E_BADARG;
    int ZL = 1;                                   // z-lanes: trade column parallelism for slice parallelism on small outputs
    while (ZL < 16 && ZL * 2 <= nz && dpp_cdiv(n, DPP_THREADS / ZL) < 512) ZL *= 2;
    int blocks = dpp_cdiv(n, DPP_THREADS / ZL);
    if (blocks > 2048) blocks = 2048;
    DPP_LAUNCH(reduce_partials_kernel, dim3(blocks), dim3(DPP_THREADS), 0, static_cast<hipStream_t>(stream),
                       partial, nz, n, bias, nbias > 0 ? nbias : 1, out, ZL);
    return dpp_launch_status();
}
