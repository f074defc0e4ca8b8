// prior.hip -- the one-off set-up and the final reporting of the embedding scripts, on the device:
//
//   pose_sample_kernel   HandDetector.sampleRandomPoses   /root/reference/src/util/handdetector.py:805-909
//                        1e6 label-space augmentations (com / rot / sc / none / rot+com / rot+com+sc) of the training poses
//   colsum / cov / jacobi  sklearn PCA(n_components=30).fit of those samples, /root/reference/src/main_nyu_posereg_embedding.py:86-92:
//                        column means, the D x D scatter matrix about them (f64), and its eigen-decomposition (cyclic Jacobi,
//                        one workgroup, f64) -- D = 3 x joints = 42 (NYU) / 48 (ICVL) / 63 (MSRA)
//   joint_error / eval_reduce  HandposeEvaluation's numeric methods, /root/reference/src/util/handpose_evaluation.py:92-228:
//                        per-joint Euclidean errors, per-frame mean / max, per-joint mean / std / max, frames within a distance
//
// All of it is bandwidth-trivial (168 MB of samples); it is here so that the training script's set-up and evaluation need no host
// pass over the data.  Reductions are two-level with fixed order (deterministic).  Compiled with -ffp-contract=off like
// augment.hip: the sampling restates NumPy arithmetic that rounds after every operation.
#include "dpp_common.h"

namespace {

struct PoseCam {
    double fx, fy, ux, uy;
    int flip_y;
};

// joint3DToImg / jointImgTo3D on float32 arrays evaluated in float64 and stored as float32, as util.handdetector's host
// restatement does (importers.py:80-119)
__device__ __forceinline__ void ps_to_img(const PoseCam& c, double x, double y, double z, float out[3]) {
    if (z == 0.0) { out[0] = (float)c.ux; out[1] = (float)c.uy; out[2] = 0.0f; return; }
    out[0] = (float)(x / z * c.fx + c.ux);
    out[1] = (float)(c.flip_y ? (c.uy - y / z * c.fy) : (y / z * c.fy + c.uy));
    out[2] = (float)z;
}
__device__ __forceinline__ void ps_to_3d(const PoseCam& c, double u, double v, double d, float out[3]) {
    out[0] = (float)((u - c.ux) * d / c.fx);
    out[1] = (float)((c.flip_y ? (c.uy - v) : (v - c.uy)) * d / c.fy);
    out[2] = (float)d;
}

constexpr int PM_NONE = 0, PM_COM = 1, PM_ROT = 2, PM_SC = 3, PM_ROTCOM = 4, PM_ROTCOMSC = 5;

// one thread per (sample, joint)
__global__ __launch_bounds__(DPP_THREADS) void pose_sample_kernel(const float* __restrict__ base_poses, const float* __restrict__ base_com,
                                                                  const float* __restrict__ base_cube, int J, const int* __restrict__ mode,
                                                                  const int* __restrict__ ridx, const double* __restrict__ off,
                                                                  const double* __restrict__ sc, const double* __restrict__ rot,
                                                                  const double* __restrict__ rot3, long n,
                                                                  PoseCam cam, float* __restrict__ out_poses, float* __restrict__ out_com,
                                                                  float* __restrict__ out_cube) {
    const long t = (long)blockIdx.x * DPP_THREADS + threadIdx.x;
    if (t >= n * J) return;
    const long i = t / J;
    const int j = (int)(t - i * J);
    const int r = ridx[i], m = mode[i];
    const float cube[3] = {base_cube[r * 3], base_cube[r * 3 + 1], base_cube[r * 3 + 2]};
    const float com[3] = {base_com[r * 3], base_com[r * 3 + 1], base_com[r * 3 + 2]};
    float p[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) p[d] = base_poses[((size_t)r * J + j) * 3 + d];
    float ncom[3] = {com[0], com[1], com[2]}, ncube[3] = {cube[0], cube[1], cube[2]};
    float o[3];
    if (m == PM_COM || m == PM_ROTCOM || m == PM_ROTCOMSC) {
#pragma unroll
        for (int d = 0; d < 3; ++d) ncom[d] = (float)((double)com[d] + off[i * 3 + d]);        // float32 array + float64 draws, stored float32
    }
    if (m == PM_SC) {
        const float s = (float)sc[i];
#pragma unroll
        for (int d = 0; d < 3; ++d) ncube[d] = cube[d] * s;                                     // float32 array x float32 scalar
    }
    const float half = ncube[2] / 2.0f;
    if (m == PM_NONE || m == PM_SC) {
#pragma unroll
        for (int d = 0; d < 3; ++d) o[d] = p[d] / half;
    } else if (m == PM_COM) {
#pragma unroll
        for (int d = 0; d < 3; ++d) o[d] = ((p[d] + com[d]) - ncom[d]) / half;
    } else if (rot3 != nullptr) {
        // rot3D=True (handdetector.py:870, 891, 903): rotatePoints3D about the (new) centre with the sample's 3x3 matrix R = rot3[i]
        // (row-major; getRotationMatrix of the three drawn angles, formed on the host), re-centred on the same centre.  As the
        // reference: the offset from the centre in float32, product and re-centring in float64, stored float32.
        float q[3];
        if (m == PM_ROT) {
#pragma unroll
            for (int d = 0; d < 3; ++d) q[d] = p[d] + ncom[d];
        } else {
#pragma unroll
            for (int d = 0; d < 3; ++d) q[d] = (p[d] + com[d]) - ncom[d];
            if (m == PM_ROTCOMSC) {
                const float s = (float)sc[i];
#pragma unroll
                for (int d = 0; d < 3; ++d) q[d] = q[d] * s;
            }
#pragma unroll
            for (int d = 0; d < 3; ++d) q[d] = q[d] + ncom[d];
        }
        const double* R = rot3 + i * 9;
        const double r0 = (double)(q[0] - ncom[0]), r1 = (double)(q[1] - ncom[1]), r2 = (double)(q[2] - ncom[2]);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float b = (float)(((R[d * 3] * r0 + R[d * 3 + 1] * r1) + R[d * 3 + 2] * r2) + (double)ncom[d]);
            o[d] = (b - ncom[d]) / half;
        }
    } else {
        // rotation in the image plane about the projected centre: 'rot' about com3D with the pose re-centred on new_com = com3D;
        // the combined modes shift (and scale) the pose first and rotate about the NEW centre, re-centring on the OLD one
        float q[3], ctr2[3], ref[3];
        if (m == PM_ROT) {
#pragma unroll
            for (int d = 0; d < 3; ++d) { q[d] = p[d] + ncom[d]; ref[d] = ncom[d]; }
            ps_to_img(cam, com[0], com[1], com[2], ctr2);
        } else {
#pragma unroll
            for (int d = 0; d < 3; ++d) q[d] = (p[d] + com[d]) - ncom[d];
            if (m == PM_ROTCOMSC) {
                const float s = (float)sc[i];
#pragma unroll
                for (int d = 0; d < 3; ++d) q[d] = q[d] * s;
            }
#pragma unroll
            for (int d = 0; d < 3; ++d) { q[d] = q[d] + com[d]; ref[d] = com[d]; }
            ps_to_img(cam, ncom[0], ncom[1], ncom[2], ctr2);
        }
        float j2[3], r2[3], b3[3];
        ps_to_img(cam, q[0], q[1], q[2], j2);
        const double alpha = rot[i] * 3.141592653589793 / 180.;
        const double ca = cos(alpha), sa = sin(alpha);
        const float pp0 = j2[0] - ctr2[0], pp1 = j2[1] - ctr2[1];
        r2[0] = (float)((double)pp0 * ca - (double)pp1 * sa) + ctr2[0];
        r2[1] = (float)((double)pp0 * sa + (double)pp1 * ca) + ctr2[1];
        r2[2] = j2[2];
        ps_to_3d(cam, r2[0], r2[1], r2[2], b3);
#pragma unroll
        for (int d = 0; d < 3; ++d) o[d] = (b3[d] - ref[d]) / half;
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) out_poses[((size_t)i * J + j) * 3 + d] = o[d];
    if (j == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if (out_com) out_com[i * 3 + d] = ncom[d];
            if (out_cube) out_cube[i * 3 + d] = ncube[d];
        }
    }
}

// ---- PCA: column sums (f64 partials per block of rows), scatter matrix about the mean, Jacobi ---------------------------------
constexpr int PCA_ROWS = 256;          // rows per workgroup
constexpr int PCA_MAXD = 80;           // 26 joints: the 256-row scatter tile (256 * D * 8 bytes) and the two D x D Jacobi matrices live in LDS

__global__ __launch_bounds__(DPP_THREADS) void pca_colsum_kernel(const float* __restrict__ X, long N, int D, double* __restrict__ partial) {
    // partial[block][d] = sum of the block's rows of column d; thread d walks its column (rows are D floats apart: the block's
    // 256 x D tile is read once, lines are shared between neighbouring threads through L1 / L2)
    const long r0 = (long)blockIdx.x * PCA_ROWS;
    const long r1 = (r0 + PCA_ROWS < N) ? r0 + PCA_ROWS : N;
    for (int d = threadIdx.x; d < D; d += DPP_THREADS) {
        double s = 0.0;
        for (long r = r0; r < r1; ++r) s += (double)X[r * D + d];
        partial[(size_t)blockIdx.x * D + d] = s;
    }
}

__global__ __launch_bounds__(DPP_THREADS) void pca_reduce_kernel(const double* __restrict__ partial, int nblk, int n, double scale,
                                                                 double* __restrict__ out) {
    // out[i] = scale * sum_b partial[b][i], fixed order
    const int i = blockIdx.x * DPP_THREADS + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += partial[(size_t)b * n + i];
    out[i] = s * scale;
}

__global__ __launch_bounds__(DPP_THREADS) void pca_scatter_kernel(const float* __restrict__ X, long N, int D, const double* __restrict__ mean,
                                                                  double* __restrict__ partial) {
    // partial[block][i][j] = sum over the block's rows of (x_i - mean_i)(x_j - mean_j); the centred rows sit in LDS as f64
    HIP_DYNAMIC_SHARED(double, xs)                          // [rows][D]
    const long r0 = (long)blockIdx.x * PCA_ROWS;
    const int rows = (int)((r0 + PCA_ROWS < N ? r0 + PCA_ROWS : N) - r0);
    for (int s = threadIdx.x; s < rows * D; s += DPP_THREADS) {
        const int r = s / D, d = s - r * D;
        xs[s] = (double)X[(r0 + r) * D + d] - mean[d];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < D * D; e += DPP_THREADS) {
        const int i = e / D, j = e - i * D;
        double a = 0.0;
        if (j >= i)                                          // upper triangle; mirrored by the reduction's reader
            for (int r = 0; r < rows; ++r) a += xs[r * D + i] * xs[r * D + j];
        partial[(size_t)blockIdx.x * D * D + e] = a;
    }
}

// Cyclic Jacobi eigenvalue iteration on a symmetric D x D matrix (one workgroup, f64).  A: in = the matrix (upper triangle
// valid), out = destroyed; evals[D] descending, evecs[k][D] = k-th eigenvector as a ROW (sklearn's components_ layout), each
// with its largest-magnitude entry positive (sklearn.utils.extmath.svd_flip with u_based_decision=False).
__global__ __launch_bounds__(DPP_THREADS) void pca_jacobi_kernel(double* __restrict__ A, int D, double* __restrict__ evals,
                                                                 double* __restrict__ evecs, int sweeps) {
    HIP_DYNAMIC_SHARED(double, sm)
    double* a = sm;                 // [D][D]
    double* v = sm + D * D;         // [D][D], columns = eigenvectors
    __shared__ double cs[2];
    __shared__ double red[2][DPP_THREADS];
    __shared__ int order[PCA_MAXD];
    const int tid = threadIdx.x;
    for (int e = tid; e < D * D; e += DPP_THREADS) {
        const int i = e / D, j = e - i * D;
        a[e] = (j >= i) ? A[e] : A[j * D + i];
        v[e] = (i == j) ? 1.0 : 0.0;
    }
    __syncthreads();
    for (int sw = 0; sw < sweeps; ++sw) {
        for (int p = 0; p < D - 1; ++p) {
            for (int q = p + 1; q < D; ++q) {
                if (tid == 0) {
                    const double apq = a[p * D + q];
                    double c = 1.0, s = 0.0;
                    if (fabs(apq) > 1e-300) {
                        const double theta = (a[q * D + q] - a[p * D + p]) / (2.0 * apq);
                        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                        c = 1.0 / sqrt(t * t + 1.0);
                        s = t * c;
                    }
                    cs[0] = c; cs[1] = s;
                }
                __syncthreads();
                const double c = cs[0], s = cs[1];
                if (s != 0.0) {
                    // rows / columns p and q of A, columns p and q of V
                    for (int k = tid; k < D; k += DPP_THREADS) {
                        const double akp = a[k * D + p], akq = a[k * D + q];
                        a[k * D + p] = c * akp - s * akq;
                        a[k * D + q] = s * akp + c * akq;
                        const double vkp = v[k * D + p], vkq = v[k * D + q];
                        v[k * D + p] = c * vkp - s * vkq;
                        v[k * D + q] = s * vkp + c * vkq;
                    }
                    __syncthreads();
                    for (int k = tid; k < D; k += DPP_THREADS) {
                        const double apk = a[p * D + k], aqk = a[q * D + k];
                        a[p * D + k] = c * apk - s * aqk;
                        a[q * D + k] = s * apk + c * aqk;
                    }
                }
                __syncthreads();
            }
        }
        // converged when the off-diagonal mass is at rounding level of the diagonal's: ||off||_F <= eps * ||diag||_F (cyclic Jacobi
        // converges quadratically, the sweep after 1e-8 is already there); `sweeps` caps the loop for pathological input
        double off = 0.0, dg = 0.0;
        for (int e = tid; e < D * D; e += DPP_THREADS) {
            const int i = e / D, j = e - i * D;
            const double x = a[e] * a[e];
            if (i == j) dg += x; else off += x;
        }
        red[0][tid] = off; red[1][tid] = dg;
        __syncthreads();
        for (int st = DPP_THREADS / 2; st > 0; st >>= 1) {
            if (tid < st) { red[0][tid] += red[0][tid + st]; red[1][tid] += red[1][tid + st]; }
            __syncthreads();
        }
        const bool done = red[0][0] <= 4.93e-32 * red[1][0];
        __syncthreads();
        if (done) break;
    }
    // sort by eigenvalue, descending (selection by rank; ties broken by index)
    for (int k = tid; k < D; k += DPP_THREADS) {
        const double ek = a[k * D + k];
        int rank = 0;
        for (int m = 0; m < D; ++m) {
            const double em = a[m * D + m];
            if (em > ek || (em == ek && m < k)) ++rank;
        }
        order[rank] = k;
    }
    __syncthreads();
    for (int k = tid; k < D; k += DPP_THREADS) {
        const int src = order[k];
        evals[k] = a[src * D + src];
        int arg = 0;
        double best = -1.0;
        for (int i = 0; i < D; ++i) {
            const double m = fabs(v[i * D + src]);
            if (m > best) { best = m; arg = i; }
        }
        const double sign = v[arg * D + src] < 0.0 ? -1.0 : 1.0;
        for (int i = 0; i < D; ++i) evecs[(size_t)k * D + i] = sign * v[i * D + src];
    }
}

// ---- evaluation -----------------------------------------------------------------------------------------------------------
// err[n][j] = |gt - pred| (NaN if any coordinate is NaN); frame[n] = (nanmean_j, nanmax_j, count_j)
__global__ __launch_bounds__(DPP_THREADS) void joint_error_kernel(const float* __restrict__ gt, const float* __restrict__ pred, int N, int J,
                                                                  double* __restrict__ err, double* __restrict__ frame) {
    const int n = blockIdx.x * DPP_THREADS + threadIdx.x;
    if (n >= N) return;
    double s = 0.0, mx = -1.0, s2 = 0.0;
    int cnt = 0;
    for (int j = 0; j < J; ++j) {
        double e2 = 0.0;
        for (int d = 0; d < 3; ++d) {
            const double dv = (double)gt[((size_t)n * J + j) * 3 + d] - (double)pred[((size_t)n * J + j) * 3 + d];
            e2 += dv * dv;
        }
        const double e = sqrt(e2);                          // NaN propagates
        err[(size_t)n * J + j] = e;
        if (e == e) { s += e; s2 += e * e; mx = e > mx ? e : mx; ++cnt; }
    }
    const double nan = __builtin_nan("");
    const double mean = cnt ? s / cnt : nan;
    frame[(size_t)n * 4 + 0] = mean;
    frame[(size_t)n * 4 + 1] = cnt ? mx : nan;
    frame[(size_t)n * 4 + 2] = (double)cnt;
    // nanstd over the joints of the frame (population), two-pass for accuracy
    double v = 0.0;
    if (cnt) {
        for (int j = 0; j < J; ++j) {
            const double e = err[(size_t)n * J + j];
            if (e == e) v += (e - mean) * (e - mean);
        }
        v = sqrt(v / cnt);
    }
    frame[(size_t)n * 4 + 3] = cnt ? v : nan;
}

// One workgroup; out layout (doubles):
//   [0] mean over frames of the frame means   [1] max error   [2] mean over frames of the frame stds   [3] frames counted
//   [4 .. 4+J)        per-joint nanmean       [4+J .. 4+2J)  per-joint nanstd       [4+2J .. 4+3J)  per-joint nanmax
//   [4+3J .. +T)      frames whose max error <= thr[t]        [4+3J+T .. +T)  frames whose mean error <= thr[t]
__global__ __launch_bounds__(DPP_THREADS) void eval_reduce_kernel(const double* __restrict__ err, const double* __restrict__ frame, int N,
                                                                  int J, const double* __restrict__ thr, int T, double* __restrict__ out) {
    __shared__ double red[DPP_THREADS];
    const int tid = threadIdx.x;
    auto block_sum = [&](double v) {
        red[tid] = v;
        __syncthreads();
        for (int s = DPP_THREADS / 2; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        const double r = red[0];
        __syncthreads();
        return r;
    };
    auto block_max = [&](double v) {
        red[tid] = v;
        __syncthreads();
        for (int s = DPP_THREADS / 2; s > 0; s >>= 1) {
            if (tid < s) red[tid] = red[tid] > red[tid + s] ? red[tid] : red[tid + s];
            __syncthreads();
        }
        const double r = red[0];
        __syncthreads();
        return r;
    };
    double sm = 0.0, ss = 0.0, mx = -1.0, cnt = 0.0;
    for (int n = tid; n < N; n += DPP_THREADS) {
        const double m = frame[(size_t)n * 4], x = frame[(size_t)n * 4 + 1], sd = frame[(size_t)n * 4 + 3];
        if (m == m) { sm += m; ss += sd; cnt += 1.0; mx = x > mx ? x : mx; }
    }
    const double tot = block_sum(cnt);
    const double a0 = block_sum(sm), a2 = block_sum(ss), a1 = block_max(mx);
    const double nan = __builtin_nan("");
    if (tid == 0) { out[0] = tot > 0 ? a0 / tot : nan; out[1] = tot > 0 ? a1 : nan; out[2] = tot > 0 ? a2 / tot : nan; out[3] = tot; }
    for (int j = 0; j < J; ++j) {
        double s = 0.0, c = 0.0, m = -1.0;
        for (int n = tid; n < N; n += DPP_THREADS) {
            const double e = err[(size_t)n * J + j];
            if (e == e) { s += e; c += 1.0; m = e > m ? e : m; }
        }
        const double S = block_sum(s), Cn = block_sum(c), M = block_max(m);
        const double mean = Cn > 0 ? S / Cn : nan;
        double v = 0.0;
        for (int n = tid; n < N; n += DPP_THREADS) {
            const double e = err[(size_t)n * J + j];
            if (e == e) v += (e - mean) * (e - mean);
        }
        const double V = block_sum(v);
        if (tid == 0) { out[4 + j] = mean; out[4 + J + j] = Cn > 0 ? sqrt(V / Cn) : nan; out[4 + 2 * J + j] = Cn > 0 ? M : nan; }
    }
    for (int t = 0; t < T; ++t) {
        double c1 = 0.0, c2 = 0.0;
        const double th = thr[t];
        for (int n = tid; n < N; n += DPP_THREADS) {
            const double m = frame[(size_t)n * 4], x = frame[(size_t)n * 4 + 1];
            if (x <= th) c1 += 1.0;                            // NaN compares false, like numpy's (nan <= dist)
            if (m <= th) c2 += 1.0;
        }
        const double C1 = block_sum(c1), C2 = block_sum(c2);
        if (tid == 0) { out[4 + 3 * J + t] = C1; out[4 + 3 * J + T + t] = C2; }
    }
}

}  // namespace

static int pose_sample_launch(const float* base_poses, const float* base_com, const float* base_cube, int n_base, int J, const int* mode,
                              const int* ridx, const double* off, const double* sc, const double* rot, const double* rot3, long n, double fx,
                              double fy, double ux, double uy, int flip_y, float* out_poses, float* out_com, float* out_cube,
                              dpp_stream_t stream) {
    if (!base_poses || !base_com || !base_cube || !mode || !ridx || !off || !sc || (!rot && !rot3) || !out_poses || n_base < 1 || J < 1 || n < 1)
        return DPP_E_BADARG;
    PoseCam cam = {fx, fy, ux, uy, flip_y};
    const long total = n * J;
    DPP_LAUNCH(pose_sample_kernel, dim3((unsigned)((total + DPP_THREADS - 1) / DPP_THREADS)), dim3(DPP_THREADS), 0,
               static_cast<hipStream_t>(stream), base_poses, base_com, base_cube, J, mode, ridx, off, sc, rot, rot3, n, cam, out_poses, out_com,
               out_cube);
    return dpp_launch_status();
}

extern "C" int dpp_pose_sample(const float* base_poses, const float* base_com, const float* base_cube, int n_base, int J, const int* mode,
                               const int* ridx, const double* off, const double* sc, const double* rot, long n, double fx, double fy,
                               double ux, double uy, int flip_y, float* out_poses, float* out_com, float* out_cube, dpp_stream_t stream) {
    if (!rot) return DPP_E_BADARG;
    return pose_sample_launch(base_poses, base_com, base_cube, n_base, J, mode, ridx, off, sc, rot, nullptr, n, fx, fy, ux, uy, flip_y, out_poses,
                              out_com, out_cube, stream);
}

extern "C" int dpp_pose_sample_rot3d(const float* base_poses, const float* base_com, const float* base_cube, int n_base, int J, const int* mode,
                                     const int* ridx, const double* off, const double* sc, const double* rot3, long n, double fx, double fy,
                                     double ux, double uy, int flip_y, float* out_poses, float* out_com, float* out_cube, dpp_stream_t stream) {
    if (!rot3) return DPP_E_BADARG;
    return pose_sample_launch(base_poses, base_com, base_cube, n_base, J, mode, ridx, off, sc, nullptr, rot3, n, fx, fy, ux, uy, flip_y, out_poses,
                              out_com, out_cube, stream);
}

extern "C" size_t dpp_pca_workspace_bytes(long N, int D) {
    const long nblk = (N + PCA_ROWS - 1) / PCA_ROWS;
    return (size_t)nblk * D * D * sizeof(double) + (size_t)D * D * sizeof(double);
}

// mean[D] (f64), evals[D] (f64, eigenvalues of the covariance with the 1/(N-1) normalisation, descending), components[D][D] (f64, rows)
extern "C" int dpp_pca_fit(const float* X, long N, int D, void* workspace, double* mean, double* evals, double* components,
                           dpp_stream_t stream) {
    if (!X || !workspace || !mean || !evals || !components || N < 2 || D < 1 || D > PCA_MAXD) return DPP_E_BADARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nblk = (int)((N + PCA_ROWS - 1) / PCA_ROWS);
    double* partial = static_cast<double*>(workspace);
    double* scatter = partial + (size_t)nblk * D * D;
    DPP_LAUNCH(pca_colsum_kernel, dim3(nblk), dim3(DPP_THREADS), 0, st, X, N, D, partial);
    DPP_LAUNCH(pca_reduce_kernel, dim3(dpp_cdiv(D, DPP_THREADS)), dim3(DPP_THREADS), 0, st, (const double*)partial, nblk, D, 1.0 / (double)N, mean);
    const size_t lds1 = (size_t)PCA_ROWS * D * sizeof(double);
    if (lds1 > 160 * 1024) return DPP_E_UNSUPPORTED;
    if (lds1 > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(&pca_scatter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1) != hipSuccess)
        return DPP_E_UNSUPPORTED;                        // a part with less LDS than the MI355X's 160 KB
    DPP_LAUNCH(pca_scatter_kernel, dim3(nblk), dim3(DPP_THREADS), lds1, st, X, N, D, (const double*)mean, partial);
    DPP_LAUNCH(pca_reduce_kernel, dim3(dpp_cdiv(D * D, DPP_THREADS)), dim3(DPP_THREADS), 0, st, (const double*)partial, nblk, D * D,
               1.0 / (double)(N - 1), scatter);
    const size_t lds2 = (size_t)2 * D * D * sizeof(double);
    if (lds2 > 160 * 1024) return DPP_E_UNSUPPORTED;
    if (lds2 > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(&pca_jacobi_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess)
        return DPP_E_UNSUPPORTED;
    DPP_LAUNCH(pca_jacobi_kernel, dim3(1), dim3(DPP_THREADS), lds2, st, scatter, D, evals, components, 30);      // sweeps: a cap, see the kernel
    return dpp_launch_status();
}

// err [N][J], frame [N][4] = (mean, max, count, std) per frame, out: see eval_reduce_kernel (4 + 3J + 2T doubles)
extern "C" int dpp_pose_eval(const float* gt, const float* pred, int N, int J, const double* thresholds, int T, double* err, double* frame,
                             double* out, dpp_stream_t stream) {
    if (!gt || !pred || !err || !frame || !out || N < 1 || J < 1 || T < 0 || (T > 0 && !thresholds)) return DPP_E_BADARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    DPP_LAUNCH(joint_error_kernel, dim3(dpp_cdiv(N, DPP_THREADS)), dim3(DPP_THREADS), 0, st, gt, pred, N, J, err, frame);
    DPP_LAUNCH(eval_reduce_kernel, dim3(1), dim3(DPP_THREADS), 0, st, (const double*)err, (const double*)frame, N, J, thresholds, T, out);
    return dpp_launch_status();
}
