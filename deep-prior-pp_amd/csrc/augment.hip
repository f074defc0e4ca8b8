// augment.hip -- the online depth-crop augmentation (rotate / scale / translate CoM) fused for gfx950.
//
// Replaces the per-sample Python + cv2 loop that the reference runs in 8 worker processes:
//   NetTrainer.augmentCrop                 /root/reference/src/trainer/nettrainer.py:919-997
//   HandDetector.moveCoM / rotateHand / scaleHand / recropHand / comToBounds / comToTransform
//                                          /root/reference/src/util/handdetector.py:204-258, 678-803
//   importer (un)projection                /root/reference/src/data/importers.py:80-119, 1187-1224
//   rotatePoint2D                          /root/reference/src/data/transformations.py:71-88
//   proj.transform(label) (PCA prior)      /root/reference/src/trainer/poseregnettrainer.py:262
// cv2.warpAffine / warpPerspective (INTER_NEAREST, BORDER_CONSTANT 0) are restated from OpenCV 2.4's
// imgwarp.cpp exactly as oracle/augment.py documents (10-bit fixed point for the affine warp, cvRound of the
// double coordinate on 64-wide blocks for the perspective warp).
//
// One launch per batch (augment_fused_kernel, dpp_augment); the two stages also exist as separate launches
// (dpp_augment_prepare + dpp_augment_warp, used by the parity tests to look at the per-sample records):
//   augment_prepare_kernel  one workgroup per crop: wave-shuffle max of the crop (the reference's `premax`),
//                           then lane 0 does the per-sample geometry in f64 (new CoM, crop transform, inverse
//                           warp matrix, z-thresholds, joint labels) and the workgroup projects the label onto
//                           the PCA prior;
//   augment_warp_kernel     one thread per output pixel: de-normalise, gather through the inverse map, z-clamp,
//                           far-plane fill, re-normalise.  HBM traffic = read 64 KB + write 64 KB per crop; the
//                           gather hits L2 (a 64 KB crop is resident).
// Compiled with -ffp-contract=off: the reference's NumPy/OpenCV arithmetic rounds after every operation, so no
// fused multiply-add may be formed here (pixel coordinates at rounding boundaries would move).
#include <stdlib.h>
#include "dpp_common.h"

namespace {

constexpr int AUG_NONE = 0, AUG_COM = 1, AUG_ROT = 2, AUG_SC = 3;
constexpr int WARP_NONE = 0, WARP_AFFINE = 1, WARP_PERSP = 2;
constexpr int MAXJ3 = 192;     // up to 64 joints x 3

struct AugRec {                // per-sample record written by prepare, read by warp
    double m[9];               // inverse map (affine uses m[0..5])
    int warp;
    int thresh;                // apply the 32000 / z-threshold rules of recropHand
    float zlo, zhi;            // zstart / zend
    float den_scale, den_off;  // img * (cz/2) + com_z  (old cube / old com)
    float premax;
    float far_v, near_v;       // com_z' +- cz'/2
    float norm_off, norm_div;  // (v - com_z') / (cz'/2)
    int binarize;              // then < 0.5 -> 0, >= 0.5 -> 1 (augment_poses' binarizeImage, poseregnettrainer.py:255-257)
};

struct AugCam {
    double fx, fy, ux, uy;
    int flip_y;
};

__device__ __forceinline__ void to3d(const AugCam& c, double u, double v, double d, float out[3]) {
    out[0] = (float)((u - c.ux) * d / c.fx);
    out[1] = (float)((c.flip_y ? (c.uy - v) : (v - c.uy)) * d / c.fy);
    out[2] = (float)d;
}

// joint3DToImg; f32in: the sample is a float32 array, so sample[0]/sample[2] is a float32 division
__device__ __forceinline__ void toimg(const AugCam& c, double x, double y, double z, bool f32in, float out[3]) {
    if (z == 0.0) { out[0] = (float)c.ux; out[1] = (float)c.uy; out[2] = 0.0f; return; }
    double q0 = x / z, q1 = y / z;
    if (f32in) { q0 = (double)((float)x / (float)z); q1 = (double)((float)y / (float)z); }
    out[0] = (float)(q0 * c.fx + c.ux);
    out[1] = (float)(c.flip_y ? (c.uy - q1 * c.fy) : (q1 * c.fy + c.uy));
    out[2] = (float)z;
}

__device__ __forceinline__ void com_to_bounds(const float com[3], const double size[3], double fx, double fy, int b[4]) {
    double c0 = com[0], c1 = com[1], c2 = com[2];
    b[0] = (int)floor((c0 * c2 / fx - size[0] / 2.) / c2 * fx + 0.5);
    b[1] = (int)floor((c0 * c2 / fx + size[0] / 2.) / c2 * fx + 0.5);
    b[2] = (int)floor((c1 * c2 / fy - size[1] / 2.) / c2 * fy + 0.5);
    b[3] = (int)floor((c1 * c2 / fy + size[1] / 2.) / c2 * fy + 0.5);
}

__device__ __forceinline__ long long floordiv(long long a, long long b) {   // python-2 integer division
    long long q = a / b;
    if ((a % b != 0) && ((a < 0) != (b < 0))) --q;
    return q;
}

// comToTransform: off . scale . trans  (3x3, row-major)
__device__ void com_to_transform(const float com[3], const double size[3], double fx, double fy, int dsz, double T[9]) {
    int b[4];
    com_to_bounds(com, size, fx, fy, b);
    int wb = b[1] - b[0], hb = b[3] - b[2];
    double s;
    long long sz0, sz1;
    if (wb > hb) { s = (double)dsz / (double)wb; sz0 = dsz; sz1 = floordiv((long long)hb * dsz, wb); }
    else { s = (double)dsz / (double)hb; sz0 = floordiv((long long)wb * dsz, hb); sz1 = dsz; }
    double xs = floor(dsz / 2. - sz1 / 2.);
    double ys = floor(dsz / 2. - sz0 / 2.);
    // off * (scale * trans): scale*trans = [[s,0,-s*xstart],[0,s,-s*ystart],[0,0,1]]
    T[0] = s; T[1] = 0.; T[2] = s * (double)(-b[0]) + xs;
    T[3] = 0.; T[4] = s; T[5] = s * (double)(-b[2]) + ys;
    T[6] = 0.; T[7] = 0.; T[8] = 1.;
}

__device__ void mat3_mul(const double A[9], const double B[9], double C[9]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += A[i * 3 + k] * B[k * 3 + j];
            C[i * 3 + j] = s;
        }
}

__device__ void mat3_inv(const double S[9], double t[9]) {     // cv::invert, 3x3 cofactor branch
    double d = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
    if (d == 0.0) { for (int i = 0; i < 9; ++i) t[i] = 0.0; return; }
    d = 1. / d;
    t[0] = (S[4] * S[8] - S[5] * S[7]) * d;
    t[1] = (S[2] * S[7] - S[1] * S[8]) * d;
    t[2] = (S[1] * S[5] - S[2] * S[4]) * d;
    t[3] = (S[5] * S[6] - S[3] * S[8]) * d;
    t[4] = (S[0] * S[8] - S[2] * S[6]) * d;
    t[5] = (S[2] * S[3] - S[0] * S[5]) * d;
    t[6] = (S[3] * S[7] - S[4] * S[6]) * d;
    t[7] = (S[1] * S[6] - S[0] * S[7]) * d;
    t[8] = (S[0] * S[4] - S[1] * S[3]) * d;
}

// ---- correctly rounded sin / cos --------------------------------------------------------------------------------------------
// cv2.getRotationMatrix2D and rotatePoint2D call libm, which on the reference's platform (glibc 2.19, IBM Accurate Mathematical
// Library) returns the CORRECTLY ROUNDED cosine / sine of the float64 angle; ocml's cos / sin (and today's NumPy) are only
// "< 1 ulp".  Rotation coefficients that differ in the last bit are a different affine map, and nearest-neighbour gathering is
// index work, so the coefficients are computed here in plain IEEE arithmetic that oracle/augment.py:sincos_cr repeats operation
// for operation: Cody-Waite reduction by pi/2 in three 33-bit parts (k * part is exact), Taylor series of sin and cos in
// double-double (two_prod through one fma: the exact error term, the pair Dekker's splitting gives the oracle), result = the
// high word.  ~106 bits, so a rounding can only be missed when the true value lies within 2^-100 of a midpoint of two doubles.
struct dd2 { double hi, lo; };
__device__ __forceinline__ dd2 dd_two_sum(double a, double b) {
    const double s = a + b, bb = s - a;
    return dd2{s, (a - (s - bb)) + (b - bb)};
}
__device__ __forceinline__ dd2 dd_fast_two_sum(double a, double b) {
    const double s = a + b;
    return dd2{s, b - (s - a)};
}
__device__ __forceinline__ dd2 dd_mul(dd2 x, dd2 y) {
    const double p = x.hi * y.hi;
    double e = __builtin_fma(x.hi, y.hi, -p);
    e = e + (x.hi * y.lo + x.lo * y.hi);
    return dd_fast_two_sum(p, e);
}
__device__ __forceinline__ dd2 dd_add(dd2 x, dd2 y) {
    dd2 s = dd_two_sum(x.hi, y.hi);
    return dd_fast_two_sum(s.hi, s.lo + (x.lo + y.lo));
}
__device__ void dpp_sincos_cr(double a, double* sn, double* cs) {      // |a| < 8
    static const double SC[12][2] = {
        {-0x1.5555555555555p-3, -0x1.5555555555555p-57}, {0x1.1111111111111p-7, 0x1.1111111111111p-63},
        {-0x1.a01a01a01a01ap-13, -0x1.a01a01a01a01ap-73}, {0x1.71de3a556c734p-19, -0x1.c154f8ddc6c00p-73},
        {-0x1.ae64567f544e4p-26, 0x1.c062e06d1f209p-80}, {0x1.6124613a86d09p-33, 0x1.f28e0cc748ebep-87},
        {-0x1.ae7f3e733b81fp-41, -0x1.1d8656b0ee8cbp-97}, {0x1.952c77030ad4ap-49, 0x1.ac981465ddc6cp-103},
        {-0x1.2f49b46814157p-57, -0x1.2650f61dbdcb4p-112}, {0x1.71b8ef6dcf572p-66, -0x1.d043ae40c4647p-120},
        {-0x1.761b41316381ap-75, 0x1.3423c7d91404fp-130}, {0x1.3f3ccdd165fa9p-84, -0x1.58ddadf344487p-139}};
    static const double CC[13][2] = {
        {-0x1.0000000000000p-1, 0x0.0p+0}, {0x1.5555555555555p-5, 0x1.5555555555555p-59},
        {-0x1.6c16c16c16c17p-10, 0x1.f49f49f49f49fp-65}, {0x1.a01a01a01a01ap-16, 0x1.a01a01a01a01ap-76},
        {-0x1.27e4fb7789f5cp-22, -0x1.cbbc05b4fa99ap-76}, {0x1.1eed8eff8d898p-29, -0x1.2aec959e14c06p-83},
        {-0x1.93974a8c07c9dp-37, -0x1.05d6f8a2efd1fp-92}, {0x1.ae7f3e733b81fp-45, 0x1.1d8656b0ee8cbp-101},
        {-0x1.6827863b97d97p-53, -0x1.eec01221a8b0bp-107}, {0x1.e542ba4020225p-62, 0x1.ea72b4afe3c2fp-120},
        {-0x1.0ce396db7f853p-70, 0x1.aebcdbd20331cp-124}, {0x1.f2cf01972f578p-80, -0x1.9ada5fcc1ab14p-135},
        {-0x1.88e85fc6a4e5ap-89, 0x1.71c37ebd16540p-143}};
    const double P1 = 0x1.921fb54400000p+0, P2 = 0x1.0b4611a600000p-34, P3 = 0x1.3198a2e000000p-69, P3T = 0x1.b839a252049c1p-104;
    const double k = rint(a * 0x1.45f306dc9c883p-1);
    dd2 r = dd_two_sum(a - k * P1, -(k * P2));
    r = dd_add(r, dd2{-(k * P3), -(k * P3T)});
    const dd2 z = dd_mul(r, r);
    dd2 ps = dd2{SC[11][0], SC[11][1]};
    for (int i = 10; i >= 0; --i) ps = dd_add(dd_mul(ps, z), dd2{SC[i][0], SC[i][1]});
    const dd2 s = dd_add(dd_mul(dd_mul(ps, z), r), r);
    dd2 pc = dd2{CC[12][0], CC[12][1]};
    for (int i = 11; i >= 0; --i) pc = dd_add(dd_mul(pc, z), dd2{CC[i][0], CC[i][1]});
    const dd2 c = dd_add(dd_mul(pc, z), dd2{1.0, 0.0});
    const int q = (int)k & 3;
    *sn = q == 0 ? s.hi : (q == 1 ? c.hi : (q == 2 ? -s.hi : -c.hi));
    *cs = q == 0 ? c.hi : (q == 1 ? -s.hi : (q == 2 ? -c.hi : s.hi));
}

// ---- counter-based RNG for on-device parameter draws (Philox-4x32-10) -----------------------------------
__device__ __forceinline__ void philox_round(unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3, unsigned k0, unsigned k1) {
    const unsigned M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    unsigned hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
    unsigned hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    unsigned n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
__device__ void philox4(unsigned long long seed, unsigned long long ctr, unsigned sub, unsigned out[4]) {
    unsigned c0 = (unsigned)ctr, c1 = (unsigned)(ctr >> 32), c2 = sub, c3 = 0;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
    for (int r = 0; r < 10; ++r) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float u01f(unsigned a) {                   // (0,1), 24 bits
    return ((float)(a >> 8) + 0.5f) * (1.0f / 16777216.0f);
}
__device__ __forceinline__ double u01(unsigned a, unsigned b) {       // (0,1), 53 bits
    unsigned long long v = (((unsigned long long)a << 32) | b) >> 11;
    return ((double)v + 0.5) * (1.0 / 9007199254740992.0);
}

struct PrepArgs {
    const float* img;        // [B][H*W] normalised crops
    const float* com3d;      // [B][3]
    const float* cube;       // [B][3]
    const float* Mcrop;      // [B][9]
    const float* gt3d;       // [B][J][3] joints relative to the CoM (mm)
    const int* mode;         // [B] or null -> drawn on device
    const double* off;       // [B][3]
    const double* rot;       // [B]
    const double* sc;        // [B]
    const int* mode_table;   // [n_modes] aug_modes as AUG_* codes (device draws index into it)
    int n_modes;
    unsigned long long seed, counter;
    const unsigned long long* counter_dev;
    unsigned long long sample0, gbatch;   // draws are keyed by (seed, step, sample0 + b) with step * gbatch + ... as the counter
    double sigma_com, sigma_sc, rot_range;
    AugCam cam;
    int B, J, dsz;
    int norm01;              // crops normalised to [0, 1] (normZeroOne) instead of [-1, 1]
    int binarize;            // binarizeImage: the augmented crop is thresholded at 0.5 (bit 1 of the ABI's norm_zero_one argument)
    const float* pca_mean;   // [J*3] or null
    const float* pca_comp;   // [E][J*3]
    int E;
    AugRec* rec;             // [B]
    float* out_y;            // [B][E] or [B][J*3]
    int* out_mode;           // [B] (optional: the mode actually used)
    unsigned long long* prof; // phase stamps (profiling build only, see dpp_stamp)
};

// What the per-joint label transforms need from the per-sample geometry (written by thread 0, read by the joint threads).
struct AugLabelCtx {
    int mode, zero;            // augmentation mode actually applied; the reference's early-out (zero offset / angle)
    float c3[3], n3[3];        // old / new CoM in 3-D
    float com[3];              // old CoM in image coordinates
    double ca, sa;             // cos / sin of the label rotation
    float half_old, half_new;  // cube_z / 2 before / after scaling
};

// The per-sample geometry of augmentCrop, run by ONE thread: the four draws, the new CoM / cube, the inverse warp matrix and
// z-thresholds (-> r) and what the label transforms need (-> lc).  `mx` is the maximum of the stored crop.
__device__ void aug_prepare_geometry(const PrepArgs& a, int b, float mx, AugRec& r, AugLabelCtx& lc) {
    const AugCam cam = a.cam;
    const double fx = fabs(cam.fx), fy = fabs(cam.fy);      // HandDetector(..., abs(di.fx), abs(di.fy))
    // every global value this lane needs, requested BEFORE anything is computed: the draws, the mode table lookup and the geometry
    // used to reach them one after the other -- five dependent memory round trips on a lane that does nothing else (tools/
    // augment_phase.py: 9 us of geometry even in 'none' mode)
    const float g_cube[3] = {a.cube[b * 3], a.cube[b * 3 + 1], a.cube[b * 3 + 2]};
    const float g_com[3] = {a.com3d[b * 3], a.com3d[b * 3 + 1], a.com3d[b * 3 + 2]};
    float g_M[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) g_M[i] = a.Mcrop[b * 9 + i];
    int g_tab[4] = {0, 0, 0, 0};
    if (!a.mode) {
#pragma unroll
        for (int i = 0; i < 4; ++i) g_tab[i] = a.mode_table[i < a.n_modes ? i : 0];
    }
    const unsigned long long g_ctr = (!a.mode && a.counter_dev) ? *a.counter_dev : 0ull;
    DPP_SCHED_FENCE();
    // ---- the four draws of augmentCrop (nettrainer.py:954-957) ----
    int mode; double off[3], rot, sc;
    if (a.mode) {
        mode = a.mode[b]; off[0] = a.off[b * 3]; off[1] = a.off[b * 3 + 1]; off[2] = a.off[b * 3 + 2]; rot = a.rot[b]; sc = a.sc[b];
    } else {
        unsigned r0[4], r1[4], r2[4];
        unsigned long long ctr = (a.counter + g_ctr) * a.gbatch + a.sample0 + b;
        philox4(a.seed, ctr, 0, r0); philox4(a.seed, ctr, 1, r1); philox4(a.seed, ctr, 2, r2);
        const unsigned mi = r0[0] % (unsigned)a.n_modes;
        mode = mi < 4u ? (mi == 0 ? g_tab[0] : (mi == 1 ? g_tab[1] : (mi == 2 ? g_tab[2] : g_tab[3]))) : a.mode_table[mi];
        // Box-Muller in float32 (hardware log / sin / cos rates): the draws are this kernel's own random numbers -- nothing pins
        // their low bits (the reference draws from NumPy's Mersenne twister) -- and in float64 the six ocml transcendentals were
        // ~8 us of the ~25 us this lane spends before the first pixel moves.  Everything DOWNSTREAM of the draws stays float64.
        const float v1 = u01f(r0[1]), v2 = u01f(r0[3]), v3 = u01f(r1[1]), v4 = u01f(r1[3]);
        const float ra = sqrtf(-2.0f * logf(v1)), rb = sqrtf(-2.0f * logf(v3));
        float s2, c2, s4, c4;
        sincosf(6.2831853f * v2, &s2, &c2);
        sincosf(6.2831853f * v4, &s4, &c4);
        off[0] = (double)(ra * c2) * a.sigma_com;
        off[1] = (double)(ra * s2) * a.sigma_com;
        off[2] = (double)(rb * c4) * a.sigma_com;
        rot = (2.0 * u01(r2[1], r2[2]) - 1.0) * a.rot_range;
        sc = fabs(1.0 + (double)(rb * s4) * a.sigma_sc);
    }
    if (a.out_mode) a.out_mode[b] = mode;

    double cube[3] = {(double)g_cube[0], (double)g_cube[1], (double)g_cube[2]};
    float com[3];           // CoM in image coordinates (float32 array in the reference)
    toimg(cam, g_com[0], g_com[1], g_com[2], true, com);
    double Mold[9];
    for (int i = 0; i < 9; ++i) Mold[i] = (double)g_M[i];

    r.warp = WARP_NONE; r.thresh = 0; r.zlo = 0.f; r.zhi = 0.f;
    for (int i = 0; i < 9; ++i) r.m[i] = (i % 4 == 0) ? 1.0 : 0.0;
    // de-normalisation of the stored crop (nettrainer.py:948-951): [-1, 1] about the CoM, or [0, 1] from the cube's front face
    r.den_scale = a.norm01 ? (float)cube[2] : (float)(cube[2] / 2.);
    r.den_off = a.norm01 ? (float)((double)com[2] - cube[2] / 2.) : com[2];
    r.premax = mx * r.den_scale + r.den_off;      // two roundings (contraction is off)
    float ncom[3] = {com[0], com[1], com[2]};
    double ncube[3] = {cube[0], cube[1], cube[2]};
    lc.mode = mode; lc.zero = 0; lc.ca = 1.0; lc.sa = 0.0;
    lc.com[0] = com[0]; lc.com[1] = com[1]; lc.com[2] = com[2];
    lc.half_old = (float)(cube[2] / 2.);
    to3d(cam, com[0], com[1], com[2], lc.c3);
    lc.n3[0] = lc.c3[0]; lc.n3[1] = lc.c3[1]; lc.n3[2] = lc.c3[2];

    if (mode == AUG_COM) {
        const bool zero = fabs(off[0]) <= 1e-8 && fabs(off[1]) <= 1e-8 && fabs(off[2]) <= 1e-8;
        lc.zero = zero;
        if (!zero) {
            toimg(cam, (double)lc.c3[0] + off[0], (double)lc.c3[1] + off[1], (double)lc.c3[2] + off[2], false, ncom);
            if (!(fabs((double)com[2]) <= 1e-8 || fabs((double)ncom[2]) <= 1e-8)) {
                double Mn[9], Mi[9], Mt[9];
                com_to_transform(ncom, cube, fx, fy, a.dsz, Mn);
                mat3_inv(Mold, Mi);                 // numpy.linalg.inv(M) restated as cofactor inverse (f64)
                mat3_mul(Mn, Mi, Mt);
                mat3_inv(Mt, r.m);                  // warpPerspective inverts the forward matrix
                r.warp = WARP_PERSP; r.thresh = 1;
                r.zlo = (float)((double)ncom[2] - cube[2] / 2.);
                r.zhi = (float)((double)ncom[2] + cube[2] / 2.);
            }
        }
        to3d(cam, ncom[0], ncom[1], ncom[2], lc.n3);
    } else if (mode == AUG_ROT) {
        const bool zero = fabs(rot) <= 1e-8;
        lc.zero = zero;
        if (!zero) {
            rot = rot - floor(rot / 360.0) * 360.0;                 // numpy.mod(rot, 360)
            // cv2.getRotationMatrix2D((W//2, H//2), -rot, 1) then the inversion at the top of cv::warpAffine
            // the warp angle -rot*pi/180 is exactly the negative of the label angle rot*pi/180 (negation is exact and commutes with
            // the two roundings), and the correctly rounded sine / cosine are odd / even: ONE evaluation serves both
            double sl, cl;
            dpp_sincos_cr(rot * 3.141592653589793 / 180., &sl, &cl);
            lc.sa = sl; lc.ca = cl;
            const double al = cl, be = -sl;
            double cx = (double)(a.dsz / 2), cy = (double)(a.dsz / 2);
            double F[6] = {al, be, (1 - al) * cx - be * cy, -be, al, be * cx + (1 - al) * cy};
            double D = F[0] * F[4] - F[1] * F[3];
            D = D != 0 ? 1. / D : 0.;
            double A11 = F[4] * D, A22 = F[0] * D;
            r.m[0] = A11; r.m[1] = F[1] * (-D); r.m[3] = F[3] * (-D); r.m[4] = A22;
            r.m[2] = -r.m[0] * F[2] - r.m[1] * F[5];
            r.m[5] = -r.m[3] * F[2] - r.m[4] * F[5];
            r.warp = WARP_AFFINE;
        }
        if (zero) {
            const double alpha = rot * 3.141592653589793 / 180.;
            dpp_sincos_cr(alpha, &lc.sa, &lc.ca);
        }
    } else if (mode == AUG_SC) {
        const bool one = fabs(sc - 1.0) <= (1e-8 + 1e-5);                  // numpy.allclose(sc, 1.)
        if (!one) {
            for (int d = 0; d < 3; ++d) ncube[d] = cube[d] * sc;
            if (!(fabs((double)com[2]) <= 1e-8)) {
                double Mn[9], Mi[9], Mt[9];
                com_to_transform(com, ncube, fx, fy, a.dsz, Mn);
                mat3_inv(Mold, Mi);
                mat3_mul(Mn, Mi, Mt);
                mat3_inv(Mt, r.m);
                r.warp = WARP_PERSP; r.thresh = 1;
                r.zlo = (float)((double)com[2] - cube[2] / 2.);      // thresholds use the OLD cube
                r.zhi = (float)((double)com[2] + cube[2] / 2.);
            }
        }
    }
    lc.half_new = (float)(ncube[2] / 2.);
    double far_d = (double)ncom[2] + ncube[2] / 2., near_d = (double)ncom[2] - ncube[2] / 2.;
    r.far_v = (float)far_d; r.near_v = (float)near_d;
    r.norm_off = a.norm01 ? r.near_v : ncom[2];              // nettrainer.py:982-995
    r.norm_div = a.norm01 ? (float)ncube[2] : (float)(ncube[2] / 2.);
    r.binarize = a.binarize;
}

// The label of ONE joint j under the augmentation described by lc (the per-joint bodies of moveCoM / rotateHand / scaleHand's
// label handling, nettrainer.py:958-981): one thread per joint.
__device__ __forceinline__ void aug_label_joint(const PrepArgs& a, int b, const AugLabelCtx& lc, int j, float* s_label) {
    const AugCam cam = a.cam;
    float g[3];
    for (int d = 0; d < 3; ++d) g[d] = a.gt3d[((size_t)b * a.J + j) * 3 + d];
    if (lc.mode == AUG_COM) {
        for (int d = 0; d < 3; ++d) {
            const float nj = lc.zero ? g[d] : ((g[d] + lc.c3[d]) - lc.n3[d]);
            s_label[j * 3 + d] = nj / lc.half_old;
        }
    } else if (lc.mode == AUG_ROT) {
        if (lc.zero) { for (int d = 0; d < 3; ++d) s_label[j * 3 + d] = g[d] / lc.half_old; return; }
        float p3[3], p2[3], pr[3], q3[3];
        for (int d = 0; d < 3; ++d) p3[d] = g[d] + lc.c3[d];
        toimg(cam, p3[0], p3[1], p3[2], true, p2);
        const float px = p2[0] - lc.com[0], py = p2[1] - lc.com[1];          // rotatePoint2D on float32 arrays
        pr[0] = (float)((double)px * lc.ca - (double)py * lc.sa);
        pr[1] = (float)((double)px * lc.sa + (double)py * lc.ca);
        pr[0] = pr[0] + lc.com[0]; pr[1] = pr[1] + lc.com[1]; pr[2] = p2[2];
        to3d(cam, pr[0], pr[1], pr[2], q3);
        for (int d = 0; d < 3; ++d) s_label[j * 3 + d] = (q3[d] - lc.c3[d]) / lc.half_old;
    } else if (lc.mode == AUG_SC) {
        for (int d = 0; d < 3; ++d) s_label[j * 3 + d] = g[d] / lc.half_new;
    } else {
        for (int d = 0; d < 3; ++d) s_label[j * 3 + d] = g[d] / lc.half_old;
    }
}

// label -> PCA prior (poseregnettrainer.py:262) or the raw normalised joints, by threads t = 0 .. nt-1 (nt a multiple of 8: the whole
// workgroup, or one wave)
// (part / nparts: this caller's share of the outputs -- the rounds r = e0 / (nt / 8) with r % nparts == part; the fused launch deals
//  them to the `splits` workgroups of a crop)
__device__ __forceinline__ void aug_project_label(const PrepArgs& a, int b, const float* s_label, int t, int nt, int part = 0, int nparts = 1) {
    const int D = a.J * 3;
    if (a.pca_comp) {
        // 8 lanes per output: lane p sums d = p, p + 8, ..., the eight partial sums meet in an xor butterfly (a fixed order).  One
        // thread per output walked D dependent multiply-adds with two global loads each: 11 us for 30 x 48.
        const int p = t & 7;
        for (int e0 = part * (nt / 8); e0 < a.E; e0 += nparts * (nt / 8)) {
            const int e = e0 + (t >> 3);
            double sum = 0.0;
            if (e < a.E) {
                const float* comp = a.pca_comp + (size_t)e * D;
                for (int d0 = p; d0 < D; d0 += 48) {               // six terms per round, their loads issued together
                    float c[6], m[6];
#pragma unroll
                    for (int u = 0; u < 6; ++u) {
                        const int d = d0 + 8 * u;
                        c[u] = d < D ? comp[d] : 0.0f;
                        m[u] = d < D ? a.pca_mean[d] : 0.0f;
                    }
#pragma unroll
                    for (int u = 0; u < 6; ++u) {
                        const int d = d0 + 8 * u;
                        if (d < D) sum += ((double)s_label[d] - (double)m[u]) * (double)c[u];
                    }
                }
            }
            sum += __shfl_xor(sum, 1);
            sum += __shfl_xor(sum, 2);
            sum += __shfl_xor(sum, 4);
            if (e < a.E && p == 0) a.out_y[(size_t)b * a.E + e] = (float)sum;
        }
    } else if (part == 0) {
        for (int d = t; d < D; d += nt) a.out_y[(size_t)b * D + d] = s_label[d];
    }
}

// maximum of a stored crop over the workgroup (the reference's `premax`): monotone under the f32 de-normalisation, so max
// first, de-normalise after.  Every thread returns the maximum.
__device__ __forceinline__ float aug_crop_max(const float* __restrict__ im, int npix, float* s_red) {
    const int tid = threadIdx.x;
    float mx = -3.4e38f;
    int i0 = 0;
    if ((npix & 3) == 0 && (reinterpret_cast<uintptr_t>(im) & 15) == 0) {
        const float4* im4 = reinterpret_cast<const float4*>(im);
        for (int i = tid; i < (npix >> 2); i += DPP_THREADS) {
            const float4 v = im4[i];
            mx = fmaxf(mx, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
        }
        i0 = npix;
    }
    for (int i = i0 + tid; i < npix; i += DPP_THREADS) mx = fmaxf(mx, im[i]);
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((tid & 63) == 0) s_red[tid >> 6] = mx;
    __syncthreads();
    mx = s_red[0];
    for (int w = 1; w < DPP_THREADS / DPP_WAVE; ++w) mx = fmaxf(mx, s_red[w]);
    return mx;
}

__global__ __launch_bounds__(DPP_THREADS) void augment_prepare_kernel(PrepArgs a) {
    dpp_kernarg_warm<sizeof(PrepArgs)>();
    __shared__ float s_red[DPP_THREADS / DPP_WAVE];
    __shared__ float s_label[MAXJ3];
    const int b = blockIdx.x;
    const float mx = aug_crop_max(a.img + (size_t)b * a.dsz * a.dsz, a.dsz * a.dsz, s_red);
    __shared__ AugLabelCtx s_lc;
    if (threadIdx.x == 0) {
        AugRec r;
        aug_prepare_geometry(a, b, mx, r, s_lc);
        a.rec[b] = r;
    }
    __syncthreads();
    if ((int)threadIdx.x < a.J) aug_label_joint(a, b, s_lc, threadIdx.x, s_label);
    __syncthreads();
    aug_project_label(a, b, s_label, threadIdx.x, DPP_THREADS);
}

__device__ __forceinline__ long long cv_round(double v) { return (long long)rint(v); }

// One output pixel (x, y) of the augmented crop: gather through the inverse map, z-rules, far-plane fill, re-normalise.
__device__ __forceinline__ float aug_warp_pixel(const AugRec& r, const float* __restrict__ im, int dsz, int x, int y) {
    float v;
    if (r.warp == WARP_NONE) {
        v = im[y * dsz + x] * r.den_scale + r.den_off;
    } else {
        long long X, Y;
        if (r.warp == WARP_AFFINE) {
            long long ad = cv_round(r.m[0] * (double)x * 1024.), bd = cv_round(r.m[3] * (double)x * 1024.);
            long long X0 = cv_round((r.m[1] * (double)y + r.m[2]) * 1024.) + 512;
            long long Y0 = cv_round((r.m[4] * (double)y + r.m[5]) * 1024.) + 512;
            X = (X0 + ad) >> 10; Y = (Y0 + bd) >> 10;
        } else {
            const int bx = (x >> 6) << 6;                       // 64-wide destination blocks of cv::warpPerspective
            const double x1 = (double)(x - bx), fbx = (double)bx, fy_ = (double)y;
            double X0 = r.m[0] * fbx + r.m[1] * fy_ + r.m[2];
            double Y0 = r.m[3] * fbx + r.m[4] * fy_ + r.m[5];
            double W0 = r.m[6] * fbx + r.m[7] * fy_ + r.m[8];
            double Wv = W0 + r.m[6] * x1;
            Wv = (Wv != 0.0) ? 1. / Wv : 0.;
            double fX = fmax(-2147483648.0, fmin(2147483647.0, (X0 + r.m[0] * x1) * Wv));
            double fY = fmax(-2147483648.0, fmin(2147483647.0, (Y0 + r.m[3] * x1) * Wv));
            X = cv_round(fX); Y = cv_round(fY);
            X = X < -32768 ? -32768 : (X > 32767 ? 32767 : X);
            Y = Y < -32768 ? -32768 : (Y > 32767 ? 32767 : Y);
        }
        v = 0.0f;                                                // BORDER_CONSTANT 0
        if (X >= 0 && X < dsz && Y >= 0 && Y < dsz) v = im[(int)Y * dsz + (int)X] * r.den_scale + r.den_off;
        if (r.thresh) {
            if (fabs((double)v - 32000.0) <= 1e-8 + 1e-5 * 32000.0) v = 0.0f;     // numpy.isclose(warped, nv_val)
            if (v < r.zlo && v != 0.0f) v = r.zlo;
            else if (v > r.zhi && v != 0.0f) v = 0.0f;
        }
    }
    if (v == r.premax) v = r.far_v;
    if (v == 0.0f) v = r.far_v;
    if (v >= r.far_v) v = r.far_v;
    if (v <= r.near_v) v = r.near_v;
    v = (v - r.norm_off) / r.norm_div;
    if (r.binarize) v = v < 0.5f ? 0.0f : 1.0f;
    return v;
}

// Four consecutive output pixels (x .. x + 3, x % 4 == 0: one row, one 64-wide block of cv::warpPerspective) with the terms that
// do not depend on x formed once: the row part of the perspective map (nine float64 products) and of the affine map (two cvRound).
// The same expressions on the same inputs as aug_warp_pixel -- bit-identical results, a third of the float64 work per pixel.
__device__ __forceinline__ float4 aug_warp_quad(const AugRec& r, const float* __restrict__ im, int dsz, int x, int y) {
    float out[4];
    if (r.warp == WARP_NONE) {
        const float4 t = *reinterpret_cast<const float4*>(im + y * dsz + x);
        out[0] = t.x * r.den_scale + r.den_off; out[1] = t.y * r.den_scale + r.den_off;
        out[2] = t.z * r.den_scale + r.den_off; out[3] = t.w * r.den_scale + r.den_off;
    } else {
        long long X[4], Y[4];
        if (r.warp == WARP_AFFINE) {
            const long long X0 = cv_round((r.m[1] * (double)y + r.m[2]) * 1024.) + 512;
            const long long Y0 = cv_round((r.m[4] * (double)y + r.m[5]) * 1024.) + 512;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long long ad = cv_round(r.m[0] * (double)(x + j) * 1024.), bd = cv_round(r.m[3] * (double)(x + j) * 1024.);
                X[j] = (X0 + ad) >> 10; Y[j] = (Y0 + bd) >> 10;
            }
        } else {
            const int bx = (x >> 6) << 6;
            const double fbx = (double)bx, fy_ = (double)y;
            const double X0 = r.m[0] * fbx + r.m[1] * fy_ + r.m[2];
            const double Y0 = r.m[3] * fbx + r.m[4] * fy_ + r.m[5];
            const double W0 = r.m[6] * fbx + r.m[7] * fy_ + r.m[8];
            // The maps the augmentation builds are products of comToTransform matrices and their cofactor inverses: the bottom row is
            // EXACTLY (0, 0, 1) (0 / det and det / det of the same rounded products), so W = 0 * x + 0 * y + 1 = 1, 1. / W = 1 and
            // X * 1. = X bit for bit -- the float64 division per pixel (a third of the pass) only runs for a truly projective map.
            const bool unit_w = r.m[6] == 0.0 && r.m[7] == 0.0 && r.m[8] == 1.0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double x1 = (double)(x + j - bx);
                double Wv = 1.0;
                if (!unit_w) {
                    Wv = W0 + r.m[6] * x1;
                    Wv = (Wv != 0.0) ? 1. / Wv : 0.;
                }
                const double fX = fmax(-2147483648.0, fmin(2147483647.0, (X0 + r.m[0] * x1) * Wv));
                const double fY = fmax(-2147483648.0, fmin(2147483647.0, (Y0 + r.m[3] * x1) * Wv));
                long long Xj = cv_round(fX), Yj = cv_round(fY);
                X[j] = Xj < -32768 ? -32768 : (Xj > 32767 ? 32767 : Xj);
                Y[j] = Yj < -32768 ? -32768 : (Yj > 32767 ? 32767 : Yj);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = 0.0f;                                          // BORDER_CONSTANT 0
            if (X[j] >= 0 && X[j] < dsz && Y[j] >= 0 && Y[j] < dsz) v = im[(int)Y[j] * dsz + (int)X[j]] * r.den_scale + r.den_off;
            if (r.thresh) {
                if (fabs((double)v - 32000.0) <= 1e-8 + 1e-5 * 32000.0) v = 0.0f;     // numpy.isclose(warped, nv_val)
                if (v < r.zlo && v != 0.0f) v = r.zlo;
                else if (v > r.zhi && v != 0.0f) v = 0.0f;
            }
            out[j] = v;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float v = out[j];
        if (v == r.premax) v = r.far_v;
        if (v == 0.0f) v = r.far_v;
        if (v >= r.far_v) v = r.far_v;
        if (v <= r.near_v) v = r.near_v;
        v = (v - r.norm_off) / r.norm_div;
        if (r.binarize) v = v < 0.5f ? 0.0f : 1.0f;
        out[j] = v;
    }
    return make_float4(out[0], out[1], out[2], out[3]);
}

// aug_warp_quad in two halves, for a pixel pass that keeps MANY gathers in flight (round 6): the source element index of each of the
// four pixels (-1: outside the crop, BORDER_CONSTANT 0) -- the same integer / float64 expressions as aug_warp_quad --, and the value
// rules applied to the gathered elements.  Bit-identical to aug_warp_quad by construction (tests: nbad == 0 in every mode).
__device__ __forceinline__ void aug_quad_index(const AugRec& r, int dsz, int x, int y, int (&idx)[4]) {
    long long X[4], Y[4];
    if (r.warp == WARP_AFFINE) {
        const long long X0 = cv_round((r.m[1] * (double)y + r.m[2]) * 1024.) + 512;
        const long long Y0 = cv_round((r.m[4] * (double)y + r.m[5]) * 1024.) + 512;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long ad = cv_round(r.m[0] * (double)(x + j) * 1024.), bd = cv_round(r.m[3] * (double)(x + j) * 1024.);
            X[j] = (X0 + ad) >> 10; Y[j] = (Y0 + bd) >> 10;
        }
    } else {
        const int bx = (x >> 6) << 6;
        const double fbx = (double)bx, fy_ = (double)y;
        const double X0 = r.m[0] * fbx + r.m[1] * fy_ + r.m[2];
        const double Y0 = r.m[3] * fbx + r.m[4] * fy_ + r.m[5];
        const double W0 = r.m[6] * fbx + r.m[7] * fy_ + r.m[8];
        const bool unit_w = r.m[6] == 0.0 && r.m[7] == 0.0 && r.m[8] == 1.0;          // see aug_warp_quad
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double x1 = (double)(x + j - bx);
            double Wv = 1.0;
            if (!unit_w) {
                Wv = W0 + r.m[6] * x1;
                Wv = (Wv != 0.0) ? 1. / Wv : 0.;
            }
            const double fX = fmax(-2147483648.0, fmin(2147483647.0, (X0 + r.m[0] * x1) * Wv));
            const double fY = fmax(-2147483648.0, fmin(2147483647.0, (Y0 + r.m[3] * x1) * Wv));
            long long Xj = cv_round(fX), Yj = cv_round(fY);
            X[j] = Xj < -32768 ? -32768 : (Xj > 32767 ? 32767 : Xj);
            Y[j] = Yj < -32768 ? -32768 : (Yj > 32767 ? 32767 : Yj);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) idx[j] = (X[j] >= 0 && X[j] < dsz && Y[j] >= 0 && Y[j] < dsz) ? (int)Y[j] * dsz + (int)X[j] : -1;
}

__device__ __forceinline__ float4 aug_quad_value(const AugRec& r, const float (&raw)[4], const int (&idx)[4]) {
    float out[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float v = idx[j] >= 0 ? raw[j] * r.den_scale + r.den_off : 0.0f;
        if (r.thresh) {
            if (fabs((double)v - 32000.0) <= 1e-8 + 1e-5 * 32000.0) v = 0.0f;     // numpy.isclose(warped, nv_val)
            if (v < r.zlo && v != 0.0f) v = r.zlo;
            else if (v > r.zhi && v != 0.0f) v = 0.0f;
        }
        if (v == r.premax) v = r.far_v;
        if (v == 0.0f) v = r.far_v;
        if (v >= r.far_v) v = r.far_v;
        if (v <= r.near_v) v = r.near_v;
        v = (v - r.norm_off) / r.norm_div;
        if (r.binarize) v = v < 0.5f ? 0.0f : 1.0f;
        out[j] = v;
    }
    return make_float4(out[0], out[1], out[2], out[3]);
}

__global__ __launch_bounds__(DPP_THREADS) void augment_warp_kernel(const float* __restrict__ img, const AugRec* __restrict__ rec, int dsz,
                                                                   float* __restrict__ out) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * DPP_THREADS + threadIdx.x;
    const int npix = dsz * dsz;
    if (p >= npix) return;
    const AugRec r = rec[b];
    const int y = p / dsz, x = p - y * dsz;
    out[(size_t)b * npix + p] = aug_warp_pixel(r, img + (size_t)b * npix, dsz, x, y);
}

// Advance the device draw counter: every workgroup takes a ticket once its own read of the counter is done (it is: the geometry
// consumed it before the barrier), the last one bumps the counter for the next launch and resets the ticket.  At the END of the
// kernel: an atomic with a returned value is a memory round trip, and in front of the barrier it delayed every pixel of the crop.
__device__ __forceinline__ void aug_take_ticket(unsigned long long* counter_rw, unsigned* ticket) {
    if (ticket == nullptr || threadIdx.x != 0) return;
    const unsigned long long c = *counter_rw;
    const unsigned one = 1u + (unsigned)(c >> 63);
    const unsigned t = atomicAdd(ticket, one);
    if (t == gridDim.x - 1) {
        *counter_rw = c + 1ull;
        *ticket = 0u;
    }
}

// The whole augmentation of a (macro-)batch as ONE launch.  Crop b is handled by S workgroups (S = a power of two chosen by the
// host so that the grid fills the chip at training batch sizes); each of them finds the crop maximum (the crop is 64 KB: after the
// first touch it is served by the XCD's L2, and the block -> (crop, split) map keeps the S workgroups of a crop on ONE XCD:
// workgroup w runs on XCD w % 8), lets its thread 0 do the per-sample geometry (identical, deterministic results in every
// split) and warps its share of the pixels, four consecutive x per thread and one 16-byte store.  Split 0 also writes the
// labels (and the optional records).  The draw counter lives on the device: after reading it every workgroup takes a ticket,
// and the one that draws the last ticket advances the counter for the next launch and resets the ticket -- so a recorded
// launch plan draws fresh parameters every step without a separate launch.
__global__ __launch_bounds__(DPP_THREADS) void augment_fused_kernel(PrepArgs a, float* __restrict__ out_x, int S,
                                                                    unsigned long long* counter_rw, unsigned* ticket) {
    dpp_kernarg_warm<sizeof(PrepArgs) + 32>();
    __shared__ float s_red[DPP_THREADS / DPP_WAVE];
    __shared__ float s_label[MAXJ3];
    __shared__ AugRec s_rec;
    const int w = blockIdx.x, tid = threadIdx.x;
    const int xcd = w & 7, q = w >> 3;                 // q-th workgroup of this XCD
    const int split = q % S, b = (q / S) * 8 + xcd;    // crops b with b % 8 == xcd live on this XCD
    const int npix = a.dsz * a.dsz;
    const bool live = b < a.B;
    const float* im = a.img + (size_t)(live ? b : 0) * npix;
    __shared__ AugLabelCtx s_lc;
    dpp_stamp(a.prof, 0);
    // Two jobs side by side: waves 1-3 find the maximum of the stored crop (`premax`, one pass over 64 KB), wave 0's first lane runs
    // the per-sample geometry, which needs the maximum only for ONE value (premax, patched in after the barrier).
    // (rotating the serial lane's wave with the workgroup index, so that co-resident workgroups do not queue their serial lanes on one
    //  SIMD, changed nothing: 26.0-26.5 us for every rotation, profiles/r03_augment_phases.txt)
    const int vt = tid;
    if (vt >= DPP_WAVE) {
        float mx = -3.4e38f;
        const int t = vt - DPP_WAVE, nt = DPP_THREADS - DPP_WAVE;
        int i0 = 0;
        if ((npix & 3) == 0 && (reinterpret_cast<uintptr_t>(im) & 15) == 0) {
            // batches of 8 loads in flight per thread: one load per iteration made this pass ~20 dependent L2 / HBM round trips,
            // the longest single item of the kernel at training batch sizes
            const float4* im4 = reinterpret_cast<const float4*>(im);
            const int n4 = npix >> 2;
            for (int i = t; i < n4; i += 8 * nt) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int j = i + u * nt;
                    v[u] = im4[j < n4 ? j : t];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) mx = fmaxf(mx, fmaxf(fmaxf(v[u].x, v[u].y), fmaxf(v[u].z, v[u].w)));
            }
            i0 = npix;
        }
        for (int i = i0 + t; i < npix; i += nt) mx = fmaxf(mx, im[i]);
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        if ((vt & 63) == 0) s_red[vt >> 6] = mx;
    }
    if (vt < DPP_WAVE) {
        // the serial lane is what the workgroup waits for: its wave issues ahead of the streaming waves that share its SIMD
        DPP_SETPRIO(3);
        if (vt == 0 && live) aug_prepare_geometry(a, b, 0.0f, s_rec, s_lc);
        DPP_SETPRIO(0);
    }
    if (vt == 0) dpp_stamp(a.prof, 1);                 // geometry (+ ticket) done on lane 0
    __syncthreads();
    dpp_stamp(a.prof, 2);                              // ... and the crop maximum on waves 1-3
    if (!live) { aug_take_ticket(counter_rw, ticket); return; }
    if (vt == 0) {
        const float mx = fmaxf(fmaxf(s_red[1], s_red[2]), s_red[3]);
        s_rec.premax = mx * s_rec.den_scale + s_rec.den_off;      // two roundings (contraction is off), as in aug_prepare_geometry
    }
    __syncthreads();
    // The labels are shared out over the crop's S workgroups: in each, wave 0 transforms the joints (one lane per joint: every workgroup
    // needs the whole label) and projects ITS rounds of the PCA outputs (8 outputs per round, round r to workgroup r % S) while waves
    // 1-3 are already moving pixels; it then takes a small share of the pixels itself (an eighth of the workgroup's, against its normal
    // quarter).  With the whole projection on split 0 that workgroup's wave 0 spent 10-14 us there (tools/augment_phase.py) and was
    // what the launch waited for.
    const int chunk = npix / S;                          // host guarantees npix % (4 * S) == 0 and dsz % 4 == 0
    const int Q = chunk >> 2;                            // pixel quads of this workgroup
    int q0 = 0, q1 = Q, t = tid, nt = DPP_THREADS;
    {
        const int Qw = Q >= 8 * DPP_WAVE ? ((Q / 8) & ~(DPP_WAVE - 1)) : 0;
        if (vt < DPP_WAVE) {
            DPP_SETPRIO(3);                              // (the label wave starts its pixels late: it issues first until then)
            if (vt < a.J) aug_label_joint(a, b, s_lc, vt, s_label);
            DPP_WAVE_SYNC();
            aug_project_label(a, b, s_label, vt, DPP_WAVE, split, S);
            if (split == 0 && a.rec != nullptr && vt == 0) a.rec[b] = s_rec;
            DPP_SETPRIO(0);
            q1 = Qw; t = vt; nt = DPP_WAVE;
        } else {
            q0 = Qw; t = vt - DPP_WAVE; nt = DPP_THREADS - DPP_WAVE;
        }
    }
    dpp_stamp(a.prof, 3);                              // labels + projection (split 0, wave 0)
    const AugRec r = s_rec;
    float* o = out_x + (size_t)b * npix;
    if (r.warp == WARP_NONE) {
        for (int q = q0 + t; q < q1; q += nt) {
            const int p = split * chunk + q * 4;
            const int y = p / a.dsz, x = p - y * a.dsz;
            *reinterpret_cast<float4*>(o + p) = aug_warp_quad(r, im, a.dsz, x, y);        // p, dsz multiples of 4: x % 4 == 0
        }
    } else {
        // Round 6: the gathers of AUG_NB quads (24 pixels) of a thread are all issued before the first is used.  One quad per iteration
        // made the pass 5-6 dependent round trips to the L2 per thread (the indices of an iteration need the float64 map, its four
        // gathers were consumed on the spot): 12-15 us per workgroup (tools/augment_phase.py) for 64 KB of pixels.  Unconditional
        // loads (pixels outside the source crop and the lanes past the end read element 0 and are masked afterwards): a branch
        // between a load and its use would drain the queue again.
        constexpr int AUG_NB = 6;
        for (int q = q0 + t; q < q1; q += AUG_NB * nt) {
            int idx[AUG_NB][4];
            float raw[AUG_NB][4];
#pragma unroll
            for (int u = 0; u < AUG_NB; ++u) {
                const int qq = q + u * nt;
                const int p = split * chunk + (qq < q1 ? qq : q) * 4;
                const int y = p / a.dsz, x = p - y * a.dsz;
                aug_quad_index(r, a.dsz, x, y, idx[u]);
            }
#pragma unroll
            for (int u = 0; u < AUG_NB; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) raw[u][j] = im[idx[u][j] >= 0 ? idx[u][j] : 0];
#pragma unroll
            for (int u = 0; u < AUG_NB; ++u) {
                const int qq = q + u * nt;
                if (qq < q1) *reinterpret_cast<float4*>(o + split * chunk + qq * 4) = aug_quad_value(r, raw[u], idx[u]);
            }
        }
    }
    dpp_stamp(a.prof, 4);
    aug_take_ticket(counter_rw, ticket);
}


// ---- initial crop: HandDetector.cropArea3D (docom = False) + Dataset.imgStackDepthOnly ------------------------------
// /root/reference/src/util/handdetector.py:53-68 (depth range of the detector), :204-226 (comToBounds), :260-296 (getCrop),
// :382-490 (cropArea3D), cv2.resize INTER_NEAREST (OpenCV 2.4 resizeNN), /root/reference/src/data/dataset.py:97-103.
// Two launches per batch of full depth frames, like the augmentation: crop_prepare (one workgroup per frame: min / max of
// the frame -> the detector's valid depth range, then lane 0 does the bounds / resize geometry in f64) and crop_warp (one
// thread per output pixel: gather through the nearest-neighbour resize map, range clamp, z-threshold, background,
// optional normalisation to [-1, 1]).
struct CropRec {
    int xstart, ystart, cw, ch;    // crop window in the frame (may leave the frame: zero padding)
    int szw, szh, xs, ys;          // resized size and paste offset inside the dsz x dsz output
    double ifx, ify;               // resizeNN: source index = min(floor(x * ifx), cw - 1)
    float min_depth, max_depth;    // detector range: outside -> 0
    float zstart, zend;
    float far_v, norm_off, norm_div;
};

__global__ __launch_bounds__(DPP_THREADS) void crop_prepare_kernel(const float* __restrict__ frames, int H, int W,
                                                                   const float* __restrict__ com, const float* __restrict__ cube,
                                                                   double fx, double fy, int dsz, int stretch,
                                                                   CropRec* __restrict__ rec, float* __restrict__ M_out) {
    __shared__ float s_mn[DPP_THREADS / DPP_WAVE], s_mx[DPP_THREADS / DPP_WAVE];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* f = frames + (size_t)b * H * W;
    float mn = 3.4e38f, mx = -3.4e38f;
    const int npx = H * W;
    int i0 = 0;
    if ((npx & 3) == 0 && (reinterpret_cast<uintptr_t>(f) & 15) == 0) {          // 16-byte loads, 4 independent chains
        const float4* f4 = reinterpret_cast<const float4*>(f);
        const int n4 = npx >> 2;
        float4 lo = make_float4(mn, mn, mn, mn), hi = make_float4(mx, mx, mx, mx);
#pragma unroll 4
        for (int i = tid; i < n4; i += DPP_THREADS) {
            float4 v = f4[i];
            lo.x = fminf(lo.x, v.x); lo.y = fminf(lo.y, v.y); lo.z = fminf(lo.z, v.z); lo.w = fminf(lo.w, v.w);
            hi.x = fmaxf(hi.x, v.x); hi.y = fmaxf(hi.y, v.y); hi.z = fmaxf(hi.z, v.z); hi.w = fmaxf(hi.w, v.w);
        }
        mn = fminf(fminf(lo.x, lo.y), fminf(lo.z, lo.w));
        mx = fmaxf(fmaxf(hi.x, hi.y), fmaxf(hi.z, hi.w));
        i0 = npx;
    }
    for (int i = i0 + tid; i < npx; i += DPP_THREADS) { float v = f[i]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
    for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
    if ((tid & 63) == 0) { s_mn[tid >> 6] = mn; s_mx[tid >> 6] = mx; }
    __syncthreads();
    if (tid != 0) return;
    for (int w = 1; w < DPP_THREADS / DPP_WAVE; ++w) { mn = fminf(mn, s_mn[w]); mx = fmaxf(mx, s_mx[w]); }
    CropRec r;
    r.max_depth = fminf(1500.0f, mx);                          // handdetector.py:60-61
    r.min_depth = fmaxf(10.0f, mn);
    const float c[3] = {com[b * 3], com[b * 3 + 1], com[b * 3 + 2]};
    const double size[3] = {(double)cube[b * 3], (double)cube[b * 3 + 1], (double)cube[b * 3 + 2]};
    int bd[4];
    com_to_bounds(c, size, fx, fy, bd);
    r.xstart = bd[0]; r.ystart = bd[2];
    const int wb = bd[1] - bd[0], hb = bd[3] - bd[2];
    r.cw = wb; r.ch = hb;
    r.zstart = (float)((double)c[2] - size[2] / 2.);
    r.zend = (float)((double)c[2] + size[2] / 2.);
    long long sz0, sz1;                                         // (width, height) of the resized crop
    if (wb > hb) { sz0 = dsz; sz1 = floordiv((long long)hb * dsz, wb); }
    else { sz0 = floordiv((long long)wb * dsz, hb); sz1 = dsz; }
    if (stretch) { sz0 = dsz; sz1 = dsz; }                      // resizeCrop(cropped, dsize): the refinement net's input, handdetector.py:430
    r.szw = (int)sz0; r.szh = (int)sz1;
    const double sc = (hb > wb) ? (double)sz1 / (double)hb : (double)sz0 / (double)wb;     // cropped.shape = (hb, wb)
    r.ifx = 1. / ((double)sz0 / (double)wb);
    r.ify = 1. / ((double)sz1 / (double)hb);
    r.xs = (int)floor(dsz / 2. - (double)sz0 / 2.);
    r.ys = (int)floor(dsz / 2. - (double)sz1 / 2.);
    r.far_v = c[2] + (float)(size[2] / 2.);
    r.norm_off = c[2];
    r.norm_div = (float)(size[2] / 2.);
    rec[b] = r;
    if (M_out) {
        float* M = M_out + (size_t)b * 9;
        M[0] = (float)sc; M[1] = 0.f; M[2] = (float)(sc * (double)(-bd[0]) + (double)r.xs);
        M[3] = 0.f; M[4] = (float)sc; M[5] = (float)(sc * (double)(-bd[2]) + (double)r.ys);
        M[6] = 0.f; M[7] = 0.f; M[8] = 1.f;
    }
}

// value of getCrop's window at window coordinates (sx, sy): zero padding outside the frame, the detector's valid depth
// range, then the z-threshold (handdetector.py:260-296)
__device__ __forceinline__ float crop_window_value(const float* __restrict__ frame, int H, int W, const CropRec& r, long long sx, long long sy) {
    const long long gx = r.xstart + sx, gy = r.ystart + sy;
    float v = 0.0f;
    if (gx >= 0 && gx < W && gy >= 0 && gy < H) {
        v = frame[(size_t)gy * W + gx];
        if (v > r.max_depth || v < r.min_depth) v = 0.0f;
    }
    if (v != 0.0f) {
        if (v < r.zstart) v = r.zstart;
        else if (v > r.zend) v = 0.0f;
    }
    return v;
}

constexpr int COM_BANDS = 16;           // row bands of a crop window, one workgroup each

// One band of window rows: (sum x, sum y, sum depth, count) of its valid pixels in f64 -> partial[b][band][4].  A wave walks whole
// rows (lane = column), so there is no division per pixel and a row's loads are contiguous.
__global__ __launch_bounds__(DPP_THREADS) void crop_com_partial_kernel(const float* __restrict__ frames, int H, int W,
                                                                       const CropRec* __restrict__ rec, double* __restrict__ partial) {
    __shared__ double s_red[4][DPP_THREADS / DPP_WAVE];
    const int b = blockIdx.y, band = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const CropRec r = rec[b];
    const float* f = frames + (size_t)b * H * W;
    const int rows = (r.ch + COM_BANDS - 1) / COM_BANDS;
    const int y0 = band * rows, y1 = (y0 + rows < r.ch) ? y0 + rows : r.ch;
    double sx = 0.0, sy = 0.0, sd = 0.0, cnt = 0.0;
    for (int y = y0 + wave; y < y1; y += DPP_THREADS / DPP_WAVE) {
        double rs = 0.0, rc = 0.0, rx = 0.0;
        for (int x = lane; x < r.cw; x += DPP_WAVE) {
            float v = crop_window_value(f, H, W, r, x, y);
            if (v < r.min_depth || v > r.max_depth) v = 0.0f;       // calculateCoM's own range test
            if (v > 0.0f) { rx += x; rs += (double)v; rc += 1.0; }
        }
        sx += rx; sd += rs; cnt += rc; sy += rc * (double)y;
    }
    for (int o = 32; o > 0; o >>= 1) {
        sx += __shfl_xor(sx, o); sy += __shfl_xor(sy, o); sd += __shfl_xor(sd, o); cnt += __shfl_xor(cnt, o);
    }
    if (lane == 0) { s_red[0][wave] = sx; s_red[1][wave] = sy; s_red[2][wave] = sd; s_red[3][wave] = cnt; }
    __syncthreads();
    if (tid != 0) return;
    for (int w = 1; w < DPP_THREADS / DPP_WAVE; ++w) { sx += s_red[0][w]; sy += s_red[1][w]; sd += s_red[2][w]; cnt += s_red[3][w]; }
    double* out = partial + ((size_t)b * COM_BANDS + band) * 4;
    out[0] = sx; out[1] = sy; out[2] = sd; out[3] = cnt;
}

// calculateCoM of the crop window (handdetector.py:91-108, as called by cropArea3D with docom=True, :413-421): mean
// column, mean row and mean depth of the pixels inside the detector's range, moved back to frame coordinates; an empty
// window falls back to the depth of its centre pixel, then to 300 mm.  One thread per frame sums the bands in order.
__global__ __launch_bounds__(DPP_THREADS) void crop_com_finish_kernel(const float* __restrict__ frames, int B, int H, int W,
                                                                      const CropRec* __restrict__ rec, const double* __restrict__ partial,
                                                                      float* __restrict__ com_out) {
    const int b = blockIdx.x * DPP_THREADS + threadIdx.x;
    if (b >= B) return;
    const CropRec r = rec[b];
    const float* f = frames + (size_t)b * H * W;
    double sx = 0.0, sy = 0.0, sd = 0.0, cnt = 0.0;
    for (int k = 0; k < COM_BANDS; ++k) {
        const double* p = partial + ((size_t)b * COM_BANDS + k) * 4;
        sx += p[0]; sy += p[1]; sd += p[2]; cnt += p[3];
    }
    double c0 = 0.0, c1 = 0.0, c2 = 0.0;
    if (cnt > 0.0) { c0 = sx / cnt; c1 = sy / cnt; c2 = sd / cnt; }
    if (fabs(c0) <= 1e-8 && fabs(c1) <= 1e-8 && fabs(c2) <= 1e-8) {       // numpy.allclose(com, 0.)
        c2 = (double)crop_window_value(f, H, W, r, r.cw / 2, r.ch / 2);
        if (fabs(c2) <= 1e-8) c2 = 300.0;
    }
    com_out[b * 3 + 0] = (float)(c0 + (double)r.xstart);
    com_out[b * 3 + 1] = (float)(c1 + (double)r.ystart);
    com_out[b * 3 + 2] = (float)c2;
}

__global__ __launch_bounds__(DPP_THREADS) void crop_warp_kernel(const float* __restrict__ frames, int H, int W, const CropRec* __restrict__ rec,
                                                                int dsz, int normalize, float nd_value, float* __restrict__ out) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * DPP_THREADS + threadIdx.x;
    if (p >= dsz * dsz) return;
    const CropRec r = rec[b];
    const int y = p / dsz, x = p - y * dsz;
    float v = nd_value;
    const int rx = x - r.xs, ry = y - r.ys;
    if (rx >= 0 && rx < r.szw && ry >= 0 && ry < r.szh) {
        long long sx = (long long)floor((double)rx * r.ifx), sy = (long long)floor((double)ry * r.ify);
        if (sx > r.cw - 1) sx = r.cw - 1;
        if (sy > r.ch - 1) sy = r.ch - 1;
        v = crop_window_value(frames + (size_t)b * H * W, H, W, r, sx, sy);
    }
    if (normalize) {
        if (v == 0.0f) v = r.far_v;                              // dataset.py:98-100
        v = (v - r.norm_off) / r.norm_div;
    }
    out[(size_t)b * dsz * dsz + p] = v;
}

// The CoM-refinement step of cropArea3D(docom=True) with a refineNet (handdetector.py:429-440, refineCoM :634-676), batched:
//   newCom3D = net_out * (cube_z / 2) + jointImgTo3D(com);  com' = joint3DToImg(newCom3D);
//   allclose(com', 0) -> com'_z = centre pixel of the (re-centred) crop window
// one workgroup per frame.  With gt3d_orig it also forms what the importers keep per frame for the NEXT crop (importers.py:388-392,
// dataset.py:103): gt3Dcrop = gt3Dorig - jointImgTo3D(com') and the training label gt3Dcrop / (cube_z / 2), optionally projected
// onto the PCA prior (poseregnettrainer.py:262) -- so that a refine -> re-crop -> regress cascade needs no host step.
__global__ __launch_bounds__(DPP_THREADS) void crop_refine_kernel(const float* __restrict__ frames, int H, int W,
                                                                  const CropRec* __restrict__ rec, const float* __restrict__ com_in,
                                                                  const float* __restrict__ cube, const float* __restrict__ net_out,
                                                                  AugCam cam, const float* __restrict__ gt3d_orig, int J,
                                                                  const float* __restrict__ pca_mean, const float* __restrict__ pca_comp,
                                                                  int E, float* __restrict__ com_out, float* __restrict__ com3d_out,
                                                                  float* __restrict__ gt3d_crop, float* __restrict__ out_y) {
    __shared__ float s_c3[3];
    __shared__ float s_label[MAXJ3];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        const float half = (float)((double)cube[b * 3 + 2] / 2.);              // size[2] / 2. as a floatX constant
        float c3[3], n3[3], c2[3];
        to3d(cam, com_in[b * 3], com_in[b * 3 + 1], com_in[b * 3 + 2], c3);
        for (int d = 0; d < 3; ++d) n3[d] = net_out[b * 3 + d] * half + c3[d];   // float32 arrays: two roundings
        toimg(cam, n3[0], n3[1], n3[2], true, c2);
        if (fabs((double)c2[0]) <= 1e-8 && fabs((double)c2[1]) <= 1e-8 && fabs((double)c2[2]) <= 1e-8) {
            const CropRec r = rec[b];
            c2[2] = crop_window_value(frames + (size_t)b * H * W, H, W, r, r.cw / 2, r.ch / 2);
        }
        for (int d = 0; d < 3; ++d) com_out[b * 3 + d] = c2[d];
        float q3[3];
        to3d(cam, c2[0], c2[1], c2[2], q3);
        for (int d = 0; d < 3; ++d) { s_c3[d] = q3[d]; if (com3d_out) com3d_out[b * 3 + d] = q3[d]; }
    }
    if (gt3d_orig == nullptr) return;
    __syncthreads();
    const float half = (float)((double)cube[b * 3 + 2] / 2.);
    for (int i = tid; i < J * 3; i += DPP_THREADS) {
        const float g = gt3d_orig[(size_t)b * J * 3 + i] - s_c3[i % 3];
        if (gt3d_crop) gt3d_crop[(size_t)b * J * 3 + i] = g;
        s_label[i] = g / half;
    }
    __syncthreads();
    if (out_y == nullptr) return;
    const int D = J * 3;
    if (pca_comp) {
        for (int e = tid; e < E; e += DPP_THREADS) {
            double s = 0.0;
            for (int d = 0; d < D; ++d) s += ((double)s_label[d] - (double)pca_mean[d]) * (double)pca_comp[(size_t)e * D + d];
            out_y[(size_t)b * E + e] = (float)s;
        }
    } else {
        for (int d = tid; d < D; d += DPP_THREADS) out_y[(size_t)b * D + d] = s_label[d];
    }
}

}  // namespace

extern "C" size_t dpp_augment_record_bytes(void) { return sizeof(AugRec); }

extern "C" int dpp_augment_prepare(const float* img, const float* com3d, const float* cube, const float* Mcrop, const float* gt3d,
                                   int B, int J, int dsz, const int* mode, const double* off, const double* rot, const double* sc,
                                   const int* mode_table, int n_modes, unsigned long long seed, unsigned long long counter,
                                   double sigma_com, double sigma_sc, double rot_range, double fx, double fy, double ux, double uy,
                                   int flip_y, int norm_zero_one, const float* pca_mean, const float* pca_comp, int E, void* records,
                                   float* out_y, int* out_mode, const unsigned long long* counter_dev, dpp_stream_t stream) {
    if (!img || !com3d || !cube || !Mcrop || !gt3d || !records || !out_y || B < 1 || J < 1 || J * 3 > MAXJ3 || dsz < 1) return DPP_E_BADARG;
    if (!mode && (!mode_table || n_modes < 1)) return DPP_E_BADARG;
    if (mode && (!off || !rot || !sc)) return DPP_E_BADARG;
    if (pca_comp && (!pca_mean || E < 1)) return DPP_E_BADARG;
    PrepArgs a;
    a.img = img; a.com3d = com3d; a.cube = cube; a.Mcrop = Mcrop; a.gt3d = gt3d;
    a.mode = mode; a.off = off; a.rot = rot; a.sc = sc; a.mode_table = mode_table; a.n_modes = n_modes;
    a.seed = seed; a.counter = counter; a.counter_dev = counter_dev; a.sigma_com = sigma_com; a.sigma_sc = sigma_sc; a.rot_range = rot_range;
    a.cam.fx = fx; a.cam.fy = fy; a.cam.ux = ux; a.cam.uy = uy; a.cam.flip_y = flip_y;
    a.B = B; a.J = J; a.dsz = dsz; a.norm01 = norm_zero_one & 1; a.binarize = (norm_zero_one >> 1) & 1; a.pca_mean = pca_mean; a.pca_comp = pca_comp; a.E = E;
    a.rec = static_cast<AugRec*>(records); a.out_y = out_y; a.out_mode = out_mode;
    a.sample0 = 0; a.gbatch = (unsigned long long)B; a.prof = nullptr;
    DPP_LAUNCH(augment_prepare_kernel, dim3(B), dim3(DPP_THREADS), 0, static_cast<hipStream_t>(stream), a);
    return dpp_launch_status();
}

extern "C" int dpp_augment(const float* img, const float* com3d, const float* cube, const float* Mcrop, const float* gt3d,
                           int B, int J, int dsz, const int* mode, const double* off, const double* rot, const double* sc,
                           const int* mode_table, int n_modes, unsigned long long seed, unsigned long long counter,
                           double sigma_com, double sigma_sc, double rot_range, double fx, double fy, double ux, double uy,
                           int flip_y, int norm_zero_one, const float* pca_mean, const float* pca_comp, int E, void* records,
                           float* out_x, float* out_y, int* out_mode, unsigned long long* counter_dev, unsigned* ticket,
                           unsigned long long sample0, unsigned long long global_batch, int splits, dpp_stream_t stream) {
    if (!img || !com3d || !cube || !Mcrop || !gt3d || !out_x || !out_y || B < 1 || J < 1 || J * 3 > MAXJ3 || dsz < 4 || (dsz & 3) || img == out_x)
        return DPP_E_BADARG;
    if (!mode && (!mode_table || n_modes < 1)) return DPP_E_BADARG;
    if (mode && (!off || !rot || !sc)) return DPP_E_BADARG;
    if (pca_comp && (!pca_mean || E < 1)) return DPP_E_BADARG;
    if (ticket && !counter_dev) return DPP_E_BADARG;
    if (global_batch < 1) global_batch = (unsigned long long)B;
    int S = splits;
    if (S <= 0) {                                   // enough workgroups for the chip: ~1024 at training batch sizes
        S = 1;
        while (S < 16 && B * S < 1024 && (dsz * dsz) % (8 * S) == 0 && dsz * dsz / (2 * S) >= DPP_THREADS * 4) S *= 2;
    }
    if (S < 1 || (S & (S - 1)) || (dsz * dsz) % (4 * S)) return DPP_E_BADARG;
    PrepArgs a;
    a.img = img; a.com3d = com3d; a.cube = cube; a.Mcrop = Mcrop; a.gt3d = gt3d;
    a.mode = mode; a.off = off; a.rot = rot; a.sc = sc; a.mode_table = mode_table; a.n_modes = n_modes;
    a.seed = seed; a.counter = counter; a.counter_dev = counter_dev; a.sigma_com = sigma_com; a.sigma_sc = sigma_sc; a.rot_range = rot_range;
    a.cam.fx = fx; a.cam.fy = fy; a.cam.ux = ux; a.cam.uy = uy; a.cam.flip_y = flip_y;
    a.B = B; a.J = J; a.dsz = dsz; a.norm01 = norm_zero_one & 1; a.binarize = (norm_zero_one >> 1) & 1; a.pca_mean = pca_mean; a.pca_comp = pca_comp; a.E = E;
    a.rec = static_cast<AugRec*>(records); a.out_y = out_y; a.out_mode = out_mode;
    a.sample0 = sample0; a.gbatch = global_batch; a.prof = dpp_prof_buffer;
    const int groups = dpp_cdiv(B, 8);             // crops are dealt to XCDs round-robin: 8 per group
    DPP_LAUNCH(augment_fused_kernel, dim3(groups * S * 8), dim3(DPP_THREADS), 0, static_cast<hipStream_t>(stream), a, out_x, S,
               counter_dev, ticket);
    return dpp_launch_status();
}

namespace {
__global__ void counter_add_kernel(unsigned long long* c, unsigned long long inc) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *c += inc;
}
}  // namespace

extern "C" int dpp_counter_add(unsigned long long* counter, unsigned long long inc, dpp_stream_t stream) {
    if (!counter) return DPP_E_BADARG;
    DPP_LAUNCH(counter_add_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), counter, inc);
    return dpp_launch_status();
}

extern "C" int dpp_augment_warp(const float* img, const void* records, int B, int dsz, float* out, dpp_stream_t stream) {
    if (!img || !records || !out || B < 1 || dsz < 1 || img == out) return DPP_E_BADARG;
    dim3 grid(dpp_cdiv(dsz * dsz, DPP_THREADS), B);
    DPP_LAUNCH(augment_warp_kernel, grid, dim3(DPP_THREADS), 0, static_cast<hipStream_t>(stream), img,
                       static_cast<const AugRec*>(records), dsz, out);
    return dpp_launch_status();
}

extern "C" size_t dpp_crop_record_bytes(void) { return sizeof(CropRec); }

extern "C" int dpp_crop_prepare(const float* frames, int B, int H, int W, const float* com, const float* cube, double fx, double fy,
                                int dsz, int stretch, void* records, float* M_out, dpp_stream_t stream) {
    if (!frames || !com || !cube || !records || B < 1 || H < 1 || W < 1 || dsz < 1 || fx == 0.0 || fy == 0.0) return DPP_E_BADARG;
    DPP_LAUNCH(crop_prepare_kernel, dim3(B), dim3(DPP_THREADS), 0, static_cast<hipStream_t>(stream), frames, H, W, com, cube,
                       fabs(fx), fabs(fy), dsz, stretch, static_cast<CropRec*>(records), M_out);
    return dpp_launch_status();
}

extern "C" int dpp_crop_warp(const float* frames, const void* records, int B, int H, int W, int dsz, int normalize, float nd_value,
                             float* out, dpp_stream_t stream) {
    if (!frames || !records || !out || B < 1 || H < 1 || W < 1 || dsz < 1) return DPP_E_BADARG;
    dim3 grid(dpp_cdiv(dsz * dsz, DPP_THREADS), B);
    DPP_LAUNCH(crop_warp_kernel, grid, dim3(DPP_THREADS), 0, static_cast<hipStream_t>(stream), frames, H, W,
                       static_cast<const CropRec*>(records), dsz, normalize, nd_value, out);
    return dpp_launch_status();
}

extern "C" size_t dpp_crop_com_workspace_bytes(int B) { return (size_t)(B > 0 ? B : 0) * COM_BANDS * 4 * sizeof(double); }

extern "C" int dpp_crop_com(const float* frames, const void* records, int B, int H, int W, void* workspace, float* com_out,
                            dpp_stream_t stream) {
    if (!frames || !records || !workspace || !com_out || B < 1 || H < 1 || W < 1) return DPP_E_BADARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    DPP_LAUNCH(crop_com_partial_kernel, dim3(COM_BANDS, B), dim3(DPP_THREADS), 0, st, frames, H, W, static_cast<const CropRec*>(records),
               static_cast<double*>(workspace));
    DPP_LAUNCH(crop_com_finish_kernel, dim3(dpp_cdiv(B, DPP_THREADS)), dim3(DPP_THREADS), 0, st, frames, B, H, W,
               static_cast<const CropRec*>(records), static_cast<const double*>(workspace), com_out);
    return dpp_launch_status();
}

extern "C" int dpp_crop_refine(const float* frames, const void* records, int B, int H, int W, const float* com_in, const float* cube,
                               const float* net_out, double fx, double fy, double ux, double uy, int flip_y, const float* gt3d_orig, int J,
                               const float* pca_mean, const float* pca_comp, int E, float* com_out, float* com3d_out, float* gt3d_crop,
                               float* out_y, dpp_stream_t stream) {
    if (!frames || !records || !com_in || !cube || !net_out || !com_out || B < 1 || H < 1 || W < 1 || fx == 0.0 || fy == 0.0) return DPP_E_BADARG;
    if (gt3d_orig && (J < 1 || J * 3 > MAXJ3)) return DPP_E_BADARG;
    if ((gt3d_crop || out_y) && !gt3d_orig) return DPP_E_BADARG;
    if (pca_comp && (!pca_mean || E < 1)) return DPP_E_BADARG;
    AugCam cam;
    cam.fx = fx; cam.fy = fy; cam.ux = ux; cam.uy = uy; cam.flip_y = flip_y;
    DPP_LAUNCH(crop_refine_kernel, dim3(B), dim3(DPP_THREADS), 0, static_cast<hipStream_t>(stream), frames, H, W,
               static_cast<const CropRec*>(records), com_in, cube, net_out, cam, gt3d_orig, J, pca_mean, pca_comp, E, com_out, com3d_out,
               gt3d_crop, out_y);
    return dpp_launch_status();
}
