// Shared by the solver translation units (solver.hip: float kernels, Lasso, the host entry points; solver_l1_gram.hip /
// solver_l1_gg.hip: the two bit-packed L1-logistic kernels): launch shape constants, lane-0 loads, additions under a lane
// mask held in SGPRs, and the launch record the host hands to the bit-packed kernels.
#pragma once
#include <type_traits>

#include "dev_utils.h"
#include "psk_internal.h"

namespace {

// One WAVE per fit (64 threads, lane l owns samples l, l+64, ...): coordinate descent is a serial chain
// of small reductions, so wave-level DPP sums (no LDS pipe, no workgroup barrier) cut the per-coordinate
// latency from ~5 us (r01 block version) to well under 1 us; all fits of a grid search still run in one
// launch, one wave per CU.
constexpr int SV_THREADS = 64;
#ifndef PSK_SV_WAVES
#define PSK_SV_WAVES 4
#endif
constexpr int SV_COOP_WAVES = PSK_SV_WAVES;                      // waves of a fit in the register form of the descent (cd_coop)
constexpr int SV_COOP_THREADS = 64 * SV_COOP_WAVES;
constexpr int SV_LDS_N = 4096;   // samples whose per-fit state fits the 64 KiB of dynamic LDS

// Per-fit scalars that lane 0 writes to global memory (w[j], column means, norms) are read back by
// lane 0 only and broadcast: a same-thread store -> load pair is always coherent, other lanes' loads
// could be served from a stale L1 line.
__device__ __forceinline__ double lane0_load(const double *p, int lane)
{
    const double v = (lane == 0) ? *p : 0.0;
    return psk_readlane_f64(v, 0);
}

// ---- additions under a lane mask held in SGPRs (register form of the descent, cd_coop) -----------------------------------
// `if (bit) g += p` compiles to v_and + v_cmp + v_add_f64 + 2 v_cndmask with the add and the selects on one dependent
// chain: a lone wave on its SIMD pays their latencies 32 times per coordinate step.  Here the condition is a wave mask
// in an SGPR pair (one v_bfe + v_cmp per word, shared by the gradient pass and the update pass of the step) that becomes
// EXEC for ONE v_add_f64; lanes outside the mask keep their value.  Same operations in the same order as the plain
// form.  EXEC is saved and restored around each group.
#define PSK_MASKED_STEP(i) "s_mov_b64 exec, %[m" #i "]\n\tv_add_f64 %[g], %[g], %[p" #i "]\n\t"
__device__ __forceinline__ void masked_sum8(double &g, const uint64_t *m, const double *p)
{
    uint64_t sv;
    asm volatile("s_mov_b64 %[sv], exec\n\t" PSK_MASKED_STEP(0) PSK_MASKED_STEP(1) PSK_MASKED_STEP(2) PSK_MASKED_STEP(3)
                 PSK_MASKED_STEP(4) PSK_MASKED_STEP(5) PSK_MASKED_STEP(6) PSK_MASKED_STEP(7) "s_mov_b64 exec, %[sv]"
                 : [g] "+v"(g), [sv] "=&s"(sv)
                 : [m0] "s"(m[0]), [m1] "s"(m[1]), [m2] "s"(m[2]), [m3] "s"(m[3]), [m4] "s"(m[4]), [m5] "s"(m[5]), [m6] "s"(m[6]),
                   [m7] "s"(m[7]), [p0] "v"(p[0]), [p1] "v"(p[1]), [p2] "v"(p[2]), [p3] "v"(p[3]), [p4] "v"(p[4]), [p5] "v"(p[5]),
                   [p6] "v"(p[6]), [p7] "v"(p[7]));
}
__device__ __forceinline__ void masked_sum2(double &g, const uint64_t *m, const double *p)
{
    uint64_t sv;
    asm volatile("s_mov_b64 %[sv], exec\n\t" PSK_MASKED_STEP(0) PSK_MASKED_STEP(1) "s_mov_b64 exec, %[sv]"
                 : [g] "+v"(g), [sv] "=&s"(sv)
                 : [m0] "s"(m[0]), [m1] "s"(m[1]), [p0] "v"(p[0]), [p1] "v"(p[1]));
}
__device__ __forceinline__ void masked_sum4(double &g, const uint64_t *m, const double *p)
{
    uint64_t sv;
    asm volatile("s_mov_b64 %[sv], exec\n\t" PSK_MASKED_STEP(0) PSK_MASKED_STEP(1) PSK_MASKED_STEP(2) PSK_MASKED_STEP(3)
                 "s_mov_b64 exec, %[sv]"
                 : [g] "+v"(g), [sv] "=&s"(sv)
                 : [m0] "s"(m[0]), [m1] "s"(m[1]), [m2] "s"(m[2]), [m3] "s"(m[3]), [p0] "v"(p[0]), [p1] "v"(p[1]), [p2] "v"(p[2]),
                   [p3] "v"(p[3]));
}
#undef PSK_MASKED_STEP
#define PSK_MASKED_STEP(i) "s_mov_b64 exec, %[m" #i "]\n\tv_add_f64 %[x" #i "], %[x" #i "], %[z]\n\t"
__device__ __forceinline__ void masked_add2(double *x, const uint64_t *m, double z)
{
    uint64_t sv;
    asm volatile("s_mov_b64 %[sv], exec\n\t" PSK_MASKED_STEP(0) PSK_MASKED_STEP(1) "s_mov_b64 exec, %[sv]"
                 : [x0] "+v"(x[0]), [x1] "+v"(x[1]), [sv] "=&s"(sv)
                 : [m0] "s"(m[0]), [m1] "s"(m[1]), [z] "v"(z));
}
__device__ __forceinline__ void masked_add4(double *x, const uint64_t *m, double z)
{
    uint64_t sv;
    asm volatile("s_mov_b64 %[sv], exec\n\t" PSK_MASKED_STEP(0) PSK_MASKED_STEP(1) PSK_MASKED_STEP(2) PSK_MASKED_STEP(3)
                 "s_mov_b64 exec, %[sv]"
                 : [x0] "+v"(x[0]), [x1] "+v"(x[1]), [x2] "+v"(x[2]), [x3] "+v"(x[3]), [sv] "=&s"(sv)
                 : [m0] "s"(m[0]), [m1] "s"(m[1]), [m2] "s"(m[2]), [m3] "s"(m[3]), [z] "v"(z));
}
__device__ __forceinline__ void masked_add8(double *x, const uint64_t *m, double z)
{
    uint64_t sv;
    asm volatile("s_mov_b64 %[sv], exec\n\t" PSK_MASKED_STEP(0) PSK_MASKED_STEP(1) PSK_MASKED_STEP(2) PSK_MASKED_STEP(3)
                 PSK_MASKED_STEP(4) PSK_MASKED_STEP(5) PSK_MASKED_STEP(6) PSK_MASKED_STEP(7) "s_mov_b64 exec, %[sv]"
                 : [x0] "+v"(x[0]), [x1] "+v"(x[1]), [x2] "+v"(x[2]), [x3] "+v"(x[3]), [x4] "+v"(x[4]), [x5] "+v"(x[5]),
                   [x6] "+v"(x[6]), [x7] "+v"(x[7]), [sv] "=&s"(sv)
                 : [m0] "s"(m[0]), [m1] "s"(m[1]), [m2] "s"(m[2]), [m3] "s"(m[3]), [m4] "s"(m[4]), [m5] "s"(m[5]), [m6] "s"(m[6]),
                   [m7] "s"(m[7]), [z] "v"(z));
}
#undef PSK_MASKED_STEP

}  // namespace

// What psk_logreg_l1_fit hands to the bit-packed kernels (one launch = all fits of a grid search).
struct psk_l1_bits_launch {
    const uint64_t *bits, *bitsT;
    const int8_t *ypm;
    const int32_t *fold, *fit_fold;
    const double *fit_param;
    int n, p, W, n_fits, max_iter;
    double tol;
    double *coef, *icpt, *work;
    int32_t *iters, *iwork;
    int f_lds, s_lds, c_lds, q_doubles, cg_max, polish_reps, gg_sl, gg_polish_from, wmreg, all_lds;
    float *gg_q;
    size_t gg_stride, lds_bytes;
    hipStream_t stream;
};
// LDS Gram block / array forms (solver_l1_gram.hip) and the Gram matrix in global memory (solver_l1_gg.hip)
hipError_t psk_l1_bits_launch_gram(const psk_l1_bits_launch &a);
hipError_t psk_l1_bits_launch_gg(const psk_l1_bits_launch &a);

// What psk_lasso_fit hands to the covariance-form Lasso (solver_lasso.hip): the co-occurrence counts of every held-out
// fold, then one workgroup per fit.
struct psk_lasso_cov_args {
    const uint64_t *bits;      // [PP][W] column bit words (sample i = bit i & 63 of word i >> 6), zero rows beyond p
    const uint64_t *tmask;     // [n_folds][W] training samples of a fold
    const double *yc;          // [n_folds][n] y - mean over the fold's training samples, 0 for the others
    const double *fstat;       // [n_folds][4]: mean of y, sum of squares of the centred y, training samples, -
    uint16_t *C;               // [n_folds][PP / 64][PP / 64] tiles of 64 x 64 co-occurrence counts, a tile as [8][64 rows][8] (out)
    double *Dg, *q0;           // [n_folds][PP / 64][64][64] centred diagonal blocks, [n_folds][PP] X'y (out)
    const int32_t *block_fit;  // [n_blocks] the fit a workgroup solves (-1: none)
    const double *fit_param;
    const int32_t *fit_fidx;   // [n_fits] index of the fit's fold
    int n, p, PP, W, n_folds, n_blocks, max_iter;
    double tol;
    double *coef, *icpt, *gaps;
    int32_t *iters;
    hipStream_t stream;
};
hipError_t psk_lasso_cov_launch(const psk_lasso_cov_args &a);
