// a10: the two L1 estimators the reference fits behind GridSearchCV (modeling.py:994-1014,
// :1075-1085, :1208-1216), as batched HIP solvers: every (grid value, CV fold) pair and the final
// refit is an independent small problem (N <= a few thousand samples x <= ~1000 selected k-mers),
// so one workgroup solves one fit and all fits of a grid search run in ONE launch.
//
//   logreg_l1_kernel  liblinear's L1R_LR objective  ||w||_1 + |b| + C sum log(1+exp(-y(w.x+b)))
//                     (intercept = penalised constant-1 feature), cyclic coordinate descent with
//                     1-D Newton steps and an Armijo line search (CDN, Yuan et al. 2010); four step
//                     lengths are evaluated per workgroup reduction.
//   lasso_kernel      (1/2n)||y - Xw - b||^2 + alpha ||w||_1, unpenalised intercept, cyclic
//                     coordinate descent on the centred problem.
//
// Layout: XT[p][n] float (column-major: one k-mer's samples are contiguous), shared by all fits and
// L2-resident; per fit the linear predictor / residual lives in LDS (global scratch beyond 4096 samples).
// Latency-bound, f64 VALU; the roofline that matters for this stage is wall-clock, not bandwidth
// (DESIGN.md).  The host removes duplicated columns first (model.GridSearch): k-mers of one gene share
// one presence pattern, and an L1 optimum may put a pattern's weight on any one of its copies.
#include "dev_utils.h"
#include "psk_internal.h"

namespace {

// One WAVE per fit (64 threads, lane l owns samples l, l+64, ...): coordinate descent is a serial chain
// of small reductions, so wave-level DPP sums (no LDS pipe, no workgroup barrier) cut the per-coordinate
// latency from ~5 us (r01 block version) to well under 1 us; all fits of a grid search still run in one
// launch, one wave per CU.
constexpr int SV_THREADS = 64;
constexpr int SV_LDS_N = 4096;   // samples whose per-fit state fits the 64 KiB of dynamic LDS

// Per-fit scalars that lane 0 writes to global memory (w[j], column means, norms) are read back by
// lane 0 only and broadcast: a same-thread store -> load pair is always coherent, other lanes' loads
// could be served from a stale L1 line.
__device__ __forceinline__ double lane0_load(const double *p, int lane)
{
    const double v = (lane == 0) ? *p : 0.0;
    return psk_readlane_f64(v, 0);
}

__device__ __forceinline__ double log1pexp_from_e(double e) { return (e > 1e300) ? log(e) : log1p(e); }

// z = linear predictor, e = exp(-y z) per sample.  With e cached, the gradient/curvature pass needs no
// transcendental at all (s = 1/(1+e)); only samples touched by an accepted step recompute e.
__global__ __launch_bounds__(SV_THREADS) void logreg_l1_kernel(const float *__restrict__ XT, const int8_t *__restrict__ ypm,
                                                                const int32_t *__restrict__ fold, int n, int p,
                                                                const double *__restrict__ fit_param,
                                                                const int32_t *__restrict__ fit_fold, double tol,
                                                                int max_iter, double *__restrict__ coef,
                                                                double *__restrict__ icpt, int32_t *__restrict__ iters,
                                                                double *__restrict__ work, const int use_lds)
{
    extern __shared__ double sm[];
    const int fit = blockIdx.x, lane = threadIdx.x;
    const double C = fit_param[fit];
    const int tf = fit_fold[fit];
    double *w = coef + (size_t)fit * p;
    double *z = use_lds ? sm : work + (size_t)fit * 2 * n;
    double *E = z + n;

    for (int j = lane; j < p; j += SV_THREADS) w[j] = 0.0;
    double npos = 0, nneg = 0;
    for (int i = lane; i < n; i += SV_THREADS) {
        z[i] = 0.0;
        E[i] = 1.0;  // exp(-y * 0)
        if (fold[i] != tf) { if (ypm[i] > 0) npos += 1; else nneg += 1; }
    }
    npos = psk_wave_sum_f64_dpp(npos);
    nneg = psk_wave_sum_f64_dpp(nneg);
    const double ntrain = npos + nneg;
    double mn = npos < nneg ? npos : nneg;
    if (mn < 1.0) mn = 1.0;
    const double eps = tol * mn / (ntrain > 0 ? ntrain : 1.0);  // liblinear's primal_solver_tol
    double wb = 0.0;
    double gnorm_init = -1.0;
    const double sigma = 0.01;
    int sweep = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();

    for (sweep = 0; sweep < max_iter; sweep++) {
        double gnorm = 0.0;
        for (int j = 0; j <= p; j++) {
            const float *col = XT + (size_t)j * n;
            double g = 0.0, h = 0.0;
            for (int i = lane; i < n; i += SV_THREADS) {
                if (fold[i] == tf) continue;
                const double x = (j < p) ? (double)col[i] : 1.0;
                if (x == 0.0) continue;
                const double y = (double)ypm[i];
                const double s = 1.0 / (1.0 + E[i]);  // sigma(y z)
                g += (s - 1.0) * y * x;
                h += x * x * s * (1.0 - s);
            }
            g = C * psk_wave_sum_f64_dpp(g);
            h = C * psk_wave_sum_f64_dpp(h) + 1e-12;
            const double wj = (j < p) ? lane0_load(&w[j], lane) : wb;
            double v;
            if (wj > 0) v = fabs(g + 1.0);
            else if (wj < 0) v = fabs(g - 1.0);
            else { v = 0.0; if (g - 1.0 > v) v = g - 1.0; if (-1.0 - g > v) v = -1.0 - g; }
            gnorm += v;
            double d;
            if (g + 1.0 <= h * wj) d = -(g + 1.0) / h;
            else if (g - 1.0 >= h * wj) d = -(g - 1.0) / h;
            else d = -wj;
            if (v < 1e-16 || d == 0.0) continue;  // wave-uniform
            const double delta = g * d + fabs(wj + d) - fabs(wj);
            // Armijo backtracking; the full Newton step is accepted almost always
            double lam = 1.0;
            bool found = false;
            for (int trial = 0; trial < 40; trial++) {
                double dl = 0.0;
                for (int i = lane; i < n; i += SV_THREADS) {
                    if (fold[i] == tf) continue;
                    const double x = (j < p) ? (double)col[i] : 1.0;
                    if (x == 0.0) continue;
                    const double y = (double)ypm[i];
                    const double e_new = exp(-y * (z[i] + lam * d * x));
                    dl += log1pexp_from_e(e_new) - log1pexp_from_e(E[i]);
                }
                dl = psk_wave_sum_f64_dpp(dl);
                const double diff = fabs(wj + lam * d) - fabs(wj) + C * dl;
                if (diff <= sigma * lam * delta) { found = true; break; }
                lam *= 0.5;
            }
            if (!found) continue;
            const double dw = lam * d;
            if (j < p) { if (lane == 0) w[j] = wj + dw; } else wb = wj + dw;
            for (int i = lane; i < n; i += SV_THREADS) {
                if (fold[i] == tf) continue;
                const double x = (j < p) ? (double)col[i] : 1.0;
                if (x != 0.0) {
                    const double zi = z[i] + dw * x;
                    z[i] = zi;
                    E[i] = exp(-(double)ypm[i] * zi);
                }
            }
        }
        if (gnorm_init < 0) gnorm_init = gnorm;
        if (gnorm <= eps * gnorm_init || gnorm == 0.0) { sweep++; break; }
    }
    if (lane == 0) { icpt[fit] = wb; iters[fit] = sweep; }
}

__global__ __launch_bounds__(SV_THREADS) void lasso_kernel(const float *__restrict__ XT, const double *__restrict__ y,
                                                            const int32_t *__restrict__ fold, int n, int p,
                                                            const double *__restrict__ fit_param,
                                                            const int32_t *__restrict__ fit_fold, double tol, int max_iter,
                                                            double *__restrict__ coef, double *__restrict__ icpt,
                                                            int32_t *__restrict__ iters, double *__restrict__ work,
                                                            const int use_lds)
{
    extern __shared__ double sm[];
    const int fit = blockIdx.x, lane = threadIdx.x;
    const double alpha = fit_param[fit];
    const int tf = fit_fold[fit];
    double *w = coef + (size_t)fit * p;
    double *xm = work + (size_t)fit * (2 * (size_t)p + n);  // column means over the training rows
    double *nrm = xm + p;                                   // centred squared norms
    double *r = use_lds ? sm : nrm + p;                     // residual of the centred problem

    double sy = 0, cnt = 0;
    for (int i = lane; i < n; i += SV_THREADS)
        if (fold[i] != tf) { sy += y[i]; cnt += 1; }
    sy = psk_wave_sum_f64_dpp(sy);
    const double ntrain = psk_wave_sum_f64_dpp(cnt);
    const double ym = sy / ntrain;
    for (int i = lane; i < n; i += SV_THREADS) r[i] = (fold[i] != tf) ? (y[i] - ym) : 0.0;
    for (int j = 0; j < p; j++) {
        const float *col = XT + (size_t)j * n;
        double s1 = 0, s2 = 0;
        for (int i = lane; i < n; i += SV_THREADS)
            if (fold[i] != tf) { const double x = col[i]; s1 += x; s2 += x * x; }
        s1 = psk_wave_sum_f64_dpp(s1);
        s2 = psk_wave_sum_f64_dpp(s2);
        if (lane == 0) {
            const double m = s1 / ntrain;
            xm[j] = m;
            nrm[j] = s2 - ntrain * m * m;  // sum (x - m)^2
            w[j] = 0.0;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    int sweep = 0;
    for (sweep = 0; sweep < max_iter; sweep++) {
        double dmax = 0.0, wmax = 0.0;
        for (int j = 0; j < p; j++) {
            const double nj = lane0_load(&nrm[j], lane);
            if (!(nj > 1e-12)) continue;
            const float *col = XT + (size_t)j * n;
            const double m = lane0_load(&xm[j], lane), wj = lane0_load(&w[j], lane);
            double s = 0.0;
            for (int i = lane; i < n; i += SV_THREADS)
                if (fold[i] != tf) s += ((double)col[i] - m) * r[i];
            const double rho = psk_wave_sum_f64_dpp(s) + nj * wj;
            const double mag = fabs(rho) - alpha * ntrain;
            const double nw = (mag > 0.0) ? ((rho > 0 ? mag : -mag) / nj) : 0.0;
            const double dd = nw - wj;
            if (dd != 0.0) {
                for (int i = lane; i < n; i += SV_THREADS)
                    if (fold[i] != tf) r[i] -= dd * ((double)col[i] - m);
                if (lane == 0) w[j] = nw;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
            }
            if (fabs(dd) > dmax) dmax = fabs(dd);
            if (fabs(nw) > wmax) wmax = fabs(nw);
        }
        if (dmax == 0.0 || dmax <= tol * (wmax > 1e-300 ? wmax : 1e-300)) { sweep++; break; }
    }
    double acc = 0.0;
    for (int j = 0; j < p; j++) acc += lane0_load(&xm[j], lane) * lane0_load(&w[j], lane);
    if (lane == 0) { icpt[fit] = ym - acc; iters[fit] = sweep; }
}

// host: transpose X[n][p] -> XT[p+1][n]
void transpose_f32(const float *X, int n, int p, std::vector<float> &XT)
{
    XT.assign((size_t)(p + 1) * n, 1.0f);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < p; j++) XT[(size_t)j * n + i] = X[(size_t)i * p + j];
}

struct SolverBufs {
    void *xt = nullptr, *y = nullptr, *fold = nullptr, *param = nullptr, *ffold = nullptr, *coef = nullptr,
         *icpt = nullptr, *iters = nullptr, *work = nullptr;
    ~SolverBufs()
    {
        void *ps[] = {xt, y, fold, param, ffold, coef, icpt, iters, work};
        for (void *q : ps) if (q) (void)hipFree(q);
    }
};

int check_fit_args(psk_ctx *ctx, const void *X, const void *y, int n, int p, const int32_t *fold, const double *fit_param,
                   const int32_t *fit_fold, int n_fits, double *coef_out, double *icpt_out)
{
    if (!ctx) return PSK_EINVAL;
    if (!X || !y || !fold || !fit_param || !fit_fold || !coef_out || !icpt_out)
        return psk_fail(ctx, PSK_EINVAL, "null buffer");
    if (n < 2 || p < 1 || n_fits < 1) return psk_fail(ctx, PSK_EINVAL, "bad problem shape n=%d p=%d fits=%d", n, p, n_fits);
    return PSK_OK;
}

}  // namespace

#define SV_ALLOC(ptr, bytes) PSK_HIP(ctx, hipMalloc(&(ptr), (bytes) ? (bytes) : 8))

extern "C" int psk_logreg_l1_fit(psk_ctx *ctx, const float *X, const int32_t *y01, int n, int p, const int32_t *fold,
                                 const double *fit_param, const int32_t *fit_fold, int n_fits, double tol, int max_iter,
                                 double *coef_out, double *icpt_out, int32_t *iters_out)
{
    PSK_TRY(check_fit_args(ctx, X, y01, n, p, fold, fit_param, fit_fold, n_fits, coef_out, icpt_out));
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<float> XT;
    transpose_f32(X, n, p, XT);
    std::vector<int8_t> ypm(n);
    for (int i = 0; i < n; i++) ypm[i] = y01[i] ? 1 : -1;
    SolverBufs b;
    SV_ALLOC(b.xt, XT.size() * sizeof(float));
    SV_ALLOC(b.y, (size_t)n);
    SV_ALLOC(b.fold, (size_t)n * 4);
    SV_ALLOC(b.param, (size_t)n_fits * 8);
    SV_ALLOC(b.ffold, (size_t)n_fits * 4);
    SV_ALLOC(b.coef, (size_t)n_fits * p * 8);
    SV_ALLOC(b.icpt, (size_t)n_fits * 8);
    SV_ALLOC(b.iters, (size_t)n_fits * 4);
    const int use_lds = n <= SV_LDS_N ? 1 : 0;
    const size_t lds = use_lds ? (size_t)2 * n * sizeof(double) : 0;
    SV_ALLOC(b.work, use_lds ? 8 : (size_t)n_fits * 2 * n * 8);
    PSK_HIP(ctx, hipMemcpyAsync(b.xt, XT.data(), XT.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(b.y, ypm.data(), n, hipMemcpyHostToDevice, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(b.fold, fold, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(b.param, fit_param, (size_t)n_fits * 8, hipMemcpyHostToDevice, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(b.ffold, fit_fold, (size_t)n_fits * 4, hipMemcpyHostToDevice, ctx->stream));
    logreg_l1_kernel<<<n_fits, SV_THREADS, lds, ctx->stream>>>(
        (const float *)b.xt, (const int8_t *)b.y, (const int32_t *)b.fold, n, p, (const double *)b.param,
        (const int32_t *)b.ffold, tol, max_iter, (double *)b.coef, (double *)b.icpt, (int32_t *)b.iters,
        (double *)b.work, use_lds);
    PSK_HIP(ctx, hipGetLastError());
    PSK_HIP(ctx, hipMemcpyAsync(coef_out, b.coef, (size_t)n_fits * p * 8, hipMemcpyDeviceToHost, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(icpt_out, b.icpt, (size_t)n_fits * 8, hipMemcpyDeviceToHost, ctx->stream));
    std::vector<int32_t> it(n_fits);
    PSK_HIP(ctx, hipMemcpyAsync(it.data(), b.iters, (size_t)n_fits * 4, hipMemcpyDeviceToHost, ctx->stream));
    PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (iters_out) memcpy(iters_out, it.data(), (size_t)n_fits * 4);
    return PSK_OK;
}

extern "C" int psk_lasso_fit(psk_ctx *ctx, const float *X, const double *y, int n, int p, const int32_t *fold,
                             const double *fit_param, const int32_t *fit_fold, int n_fits, double tol, int max_iter,
                             double *coef_out, double *icpt_out, int32_t *iters_out)
{
    PSK_TRY(check_fit_args(ctx, X, y, n, p, fold, fit_param, fit_fold, n_fits, coef_out, icpt_out));
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<float> XT;
    transpose_f32(X, n, p, XT);
    SolverBufs b;
    SV_ALLOC(b.xt, XT.size() * sizeof(float));
    SV_ALLOC(b.y, (size_t)n * 8);
    SV_ALLOC(b.fold, (size_t)n * 4);
    SV_ALLOC(b.param, (size_t)n_fits * 8);
    SV_ALLOC(b.ffold, (size_t)n_fits * 4);
    SV_ALLOC(b.coef, (size_t)n_fits * p * 8);
    SV_ALLOC(b.icpt, (size_t)n_fits * 8);
    SV_ALLOC(b.iters, (size_t)n_fits * 4);
    const int use_lds = n <= 2 * SV_LDS_N ? 1 : 0;   // one residual array: 8 KiB samples fit
    const size_t lds = use_lds ? (size_t)n * sizeof(double) : 0;
    SV_ALLOC(b.work, (size_t)n_fits * (2 * (size_t)p + n) * 8);
    PSK_HIP(ctx, hipMemcpyAsync(b.xt, XT.data(), XT.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(b.y, y, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(b.fold, fold, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(b.param, fit_param, (size_t)n_fits * 8, hipMemcpyHostToDevice, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(b.ffold, fit_fold, (size_t)n_fits * 4, hipMemcpyHostToDevice, ctx->stream));
    lasso_kernel<<<n_fits, SV_THREADS, lds, ctx->stream>>>((const float *)b.xt, (const double *)b.y,
                                                           (const int32_t *)b.fold, n, p, (const double *)b.param,
                                                           (const int32_t *)b.ffold, tol, max_iter, (double *)b.coef,
                                                           (double *)b.icpt, (int32_t *)b.iters, (double *)b.work,
                                                           use_lds);
    PSK_HIP(ctx, hipGetLastError());
    PSK_HIP(ctx, hipMemcpyAsync(coef_out, b.coef, (size_t)n_fits * p * 8, hipMemcpyDeviceToHost, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(icpt_out, b.icpt, (size_t)n_fits * 8, hipMemcpyDeviceToHost, ctx->stream));
    std::vector<int32_t> it(n_fits);
    PSK_HIP(ctx, hipMemcpyAsync(it.data(), b.iters, (size_t)n_fits * 4, hipMemcpyDeviceToHost, ctx->stream));
    PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (iters_out) memcpy(iters_out, it.data(), (size_t)n_fits * 4);
    return PSK_OK;
}
