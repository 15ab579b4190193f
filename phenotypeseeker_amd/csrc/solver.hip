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
// L2-resident; per fit the linear predictor / residual lives in LDS.  Latency-bound, f64 VALU; the
// roofline that matters for this stage is wall-clock, not bandwidth.
#include "dev_utils.h"
#include "psk_internal.h"

namespace {

constexpr int SV_THREADS = 256;
constexpr int SV_MAXN = 8192;  // samples per fit (LDS: 8 B each)

struct Red4 { double a, b, c, d; };

__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += psk_shfl_xor_f64(v, d);
    return v;
}

// sums four doubles over the workgroup; result valid in every thread.  lds: 4 * (SV_THREADS/64) doubles
__device__ __forceinline__ Red4 block_sum4(Red4 v, double *lds)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    v.a = wave_sum_f64(v.a); v.b = wave_sum_f64(v.b); v.c = wave_sum_f64(v.c); v.d = wave_sum_f64(v.d);
    __syncthreads();
    if (lane == 0) { lds[wid * 4 + 0] = v.a; lds[wid * 4 + 1] = v.b; lds[wid * 4 + 2] = v.c; lds[wid * 4 + 3] = v.d; }
    __syncthreads();
    Red4 r{0, 0, 0, 0};
#pragma unroll
    for (int w = 0; w < SV_THREADS / 64; w++) {
        r.a += lds[w * 4 + 0]; r.b += lds[w * 4 + 1]; r.c += lds[w * 4 + 2]; r.d += lds[w * 4 + 3];
    }
    return r;
}

__device__ __forceinline__ double log1pexp(double x)
{
    if (x > 35.0) return x;
    if (x < -35.0) return exp(x);
    return log1p(exp(x));
}

__global__ __launch_bounds__(SV_THREADS) void logreg_l1_kernel(const float *__restrict__ XT, const int8_t *__restrict__ ypm,
                                                                const int32_t *__restrict__ fold, int n, int p,
                                                                const double *__restrict__ fit_param,
                                                                const int32_t *__restrict__ fit_fold, double tol,
                                                                int max_iter, double *__restrict__ coef,
                                                                double *__restrict__ icpt, int32_t *__restrict__ iters)
{
    __shared__ double z[SV_MAXN];
    __shared__ double red[4 * (SV_THREADS / 64)];
    const int fit = blockIdx.x, tid = threadIdx.x;
    const double C = fit_param[fit];
    const int tf = fit_fold[fit];
    double *w = coef + (size_t)fit * p;

    for (int j = tid; j < p; j += SV_THREADS) w[j] = 0.0;
    double npos = 0, nneg = 0;
    for (int i = tid; i < n; i += SV_THREADS) {
        z[i] = 0.0;
        if (fold[i] != tf) { if (ypm[i] > 0) npos += 1; else nneg += 1; }
    }
    Red4 cnt = block_sum4(Red4{npos, nneg, 0, 0}, red);
    const double ntrain = cnt.a + cnt.b;
    double mn = cnt.a < cnt.b ? cnt.a : cnt.b;
    if (mn < 1.0) mn = 1.0;
    const double eps = tol * mn / (ntrain > 0 ? ntrain : 1.0);  // liblinear's primal_solver_tol
    double wb = 0.0;  // intercept weight (same value in every thread)
    double gnorm_init = -1.0;
    const double sigma = 0.01;
    int sweep = 0;
    __syncthreads();

    for (sweep = 0; sweep < max_iter; sweep++) {
        double gnorm = 0.0;
        for (int j = 0; j <= p; j++) {
            const float *col = XT + (size_t)j * n;
            // gradient / curvature of the loss along coordinate j over the training rows
            double g = 0.0, h = 0.0;
            for (int i = tid; i < n; i += SV_THREADS) {
                if (fold[i] == tf) continue;
                const double x = (j < p) ? (double)col[i] : 1.0;
                if (x == 0.0) continue;
                const double y = (double)ypm[i];
                const double s = 1.0 / (1.0 + exp(-y * z[i]));
                g += (s - 1.0) * y * x;
                h += x * x * s * (1.0 - s);
            }
            Red4 r = block_sum4(Red4{g, h, 0, 0}, red);
            g = C * r.a;
            h = C * r.b + 1e-12;
            const double wj = (j < p) ? w[j] : wb;
            double v;
            if (wj > 0) v = fabs(g + 1.0);
            else if (wj < 0) v = fabs(g - 1.0);
            else { v = 0.0; if (g - 1.0 > v) v = g - 1.0; if (-1.0 - g > v) v = -1.0 - g; }
            gnorm += v;
            double d;
            if (g + 1.0 <= h * wj) d = -(g + 1.0) / h;
            else if (g - 1.0 >= h * wj) d = -(g - 1.0) / h;
            else d = -wj;
            if (v < 1e-16 || d == 0.0) continue;  // uniform across the workgroup
            const double delta = g * d + fabs(wj + d) - fabs(wj);
            // Armijo search, four step lengths per reduction
            double lam = 1.0, step = 0.0;
            bool found = false;
            for (int round = 0; round < 16 && !found; round++) {
                double l0 = 0, l1 = 0, l2 = 0, l3 = 0;
                for (int i = tid; i < n; i += SV_THREADS) {
                    if (fold[i] == tf) continue;
                    const double x = (j < p) ? (double)col[i] : 1.0;
                    if (x == 0.0) continue;
                    const double y = (double)ypm[i], zi = z[i];
                    const double base = log1pexp(-y * zi);
                    const double dx = d * x;
                    l0 += log1pexp(-y * (zi + lam * dx)) - base;
                    l1 += log1pexp(-y * (zi + 0.5 * lam * dx)) - base;
                    l2 += log1pexp(-y * (zi + 0.25 * lam * dx)) - base;
                    l3 += log1pexp(-y * (zi + 0.125 * lam * dx)) - base;
                }
                Red4 q = block_sum4(Red4{l0, l1, l2, l3}, red);
                const double ls[4] = {q.a, q.b, q.c, q.d};
                double t = lam;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const double diff = fabs(wj + t * d) - fabs(wj) + C * ls[c];
                    if (!found && diff <= sigma * t * delta) { found = true; step = t; }
                    t *= 0.5;
                }
                lam *= 0.0625;
            }
            if (!found) continue;
            const double dw = step * d;
            if (j < p) { if (tid == 0) w[j] = wj + dw; } else wb = wj + dw;
            for (int i = tid; i < n; i += SV_THREADS) {
                const double x = (j < p) ? (double)col[i] : 1.0;
                if (x != 0.0) z[i] += dw * x;  // test rows too: z then also serves prediction
            }
            __syncthreads();
        }
        if (gnorm_init < 0) gnorm_init = gnorm;
        if (gnorm <= eps * gnorm_init || gnorm == 0.0) { sweep++; break; }
    }
    __syncthreads();
    if (tid == 0) { icpt[fit] = wb; iters[fit] = sweep; }
}

__global__ __launch_bounds__(SV_THREADS) void lasso_kernel(const float *__restrict__ XT, const double *__restrict__ y,
                                                            const int32_t *__restrict__ fold, int n, int p,
                                                            const double *__restrict__ fit_param,
                                                            const int32_t *__restrict__ fit_fold, double tol, int max_iter,
                                                            double *__restrict__ coef, double *__restrict__ icpt,
                                                            int32_t *__restrict__ iters, double *__restrict__ work)
{
    __shared__ double r[SV_MAXN];
    __shared__ double red[4 * (SV_THREADS / 64)];
    const int fit = blockIdx.x, tid = threadIdx.x;
    const double alpha = fit_param[fit];
    const int tf = fit_fold[fit];
    double *w = coef + (size_t)fit * p;
    double *xm = work + (size_t)fit * 2 * p;  // column means over the training rows
    double *nrm = xm + p;                     // centred squared norms

    double sy = 0, cnt = 0;
    for (int i = tid; i < n; i += SV_THREADS)
        if (fold[i] != tf) { sy += y[i]; cnt += 1; }
    Red4 q = block_sum4(Red4{sy, cnt, 0, 0}, red);
    const double ntrain = q.b;
    const double ym = q.a / ntrain;
    for (int i = tid; i < n; i += SV_THREADS) r[i] = (fold[i] != tf) ? (y[i] - ym) : 0.0;
    for (int j = 0; j < p; j++) {
        const float *col = XT + (size_t)j * n;
        double s1 = 0, s2 = 0;
        for (int i = tid; i < n; i += SV_THREADS)
            if (fold[i] != tf) { const double x = col[i]; s1 += x; s2 += x * x; }
        Red4 t = block_sum4(Red4{s1, s2, 0, 0}, red);
        if (tid == 0) {
            const double m = t.a / ntrain;
            xm[j] = m;
            nrm[j] = t.b - ntrain * m * m;  // sum (x - m)^2
            w[j] = 0.0;
        }
    }
    __syncthreads();
    int sweep = 0;
    for (sweep = 0; sweep < max_iter; sweep++) {
        double dmax = 0.0, wmax = 0.0;
        for (int j = 0; j < p; j++) {
            const double nj = nrm[j];
            if (!(nj > 1e-12)) continue;
            const float *col = XT + (size_t)j * n;
            const double m = xm[j], wj = w[j];
            double s = 0.0;
            for (int i = tid; i < n; i += SV_THREADS)
                if (fold[i] != tf) s += ((double)col[i] - m) * r[i];
            Red4 t = block_sum4(Red4{s, 0, 0, 0}, red);
            const double rho = t.a + nj * wj;
            const double mag = fabs(rho) - alpha * ntrain;
            const double nw = (mag > 0.0) ? ((rho > 0 ? mag : -mag) / nj) : 0.0;
            const double dd = nw - wj;
            if (dd != 0.0) {
                for (int i = tid; i < n; i += SV_THREADS)
                    if (fold[i] != tf) r[i] -= dd * ((double)col[i] - m);
                __syncthreads();
                if (tid == 0) w[j] = nw;
            }
            if (fabs(dd) > dmax) dmax = fabs(dd);
            if (fabs(nw) > wmax) wmax = fabs(nw);
        }
        __syncthreads();
        if (dmax == 0.0 || dmax <= tol * (wmax > 1e-300 ? wmax : 1e-300)) { sweep++; break; }
    }
    __syncthreads();
    // intercept = ym - xm . w
    double acc = 0.0;
    for (int j = tid; j < p; j += SV_THREADS) acc += xm[j] * w[j];
    Red4 t = block_sum4(Red4{acc, 0, 0, 0}, red);
    if (tid == 0) { icpt[fit] = ym - t.a; iters[fit] = sweep; }
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
    if (n > SV_MAXN) return psk_fail(ctx, PSK_ERANGE, "at most %d samples per fit (got %d)", SV_MAXN, n);
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
    PSK_HIP(ctx, hipMemcpyAsync(b.xt, XT.data(), XT.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(b.y, ypm.data(), n, hipMemcpyHostToDevice, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(b.fold, fold, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(b.param, fit_param, (size_t)n_fits * 8, hipMemcpyHostToDevice, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(b.ffold, fit_fold, (size_t)n_fits * 4, hipMemcpyHostToDevice, ctx->stream));
    logreg_l1_kernel<<<n_fits, SV_THREADS, 0, ctx->stream>>>(
        (const float *)b.xt, (const int8_t *)b.y, (const int32_t *)b.fold, n, p, (const double *)b.param,
        (const int32_t *)b.ffold, tol, max_iter, (double *)b.coef, (double *)b.icpt, (int32_t *)b.iters);
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
    SV_ALLOC(b.work, (size_t)n_fits * 2 * p * 8);
    PSK_HIP(ctx, hipMemcpyAsync(b.xt, XT.data(), XT.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(b.y, y, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(b.fold, fold, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(b.param, fit_param, (size_t)n_fits * 8, hipMemcpyHostToDevice, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(b.ffold, fit_fold, (size_t)n_fits * 4, hipMemcpyHostToDevice, ctx->stream));
    lasso_kernel<<<n_fits, SV_THREADS, 0, ctx->stream>>>((const float *)b.xt, (const double *)b.y,
                                                         (const int32_t *)b.fold, n, p, (const double *)b.param,
                                                         (const int32_t *)b.ffold, tol, max_iter, (double *)b.coef,
                                                         (double *)b.icpt, (int32_t *)b.iters, (double *)b.work);
    PSK_HIP(ctx, hipGetLastError());
    PSK_HIP(ctx, hipMemcpyAsync(coef_out, b.coef, (size_t)n_fits * p * 8, hipMemcpyDeviceToHost, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(icpt_out, b.icpt, (size_t)n_fits * 8, hipMemcpyDeviceToHost, ctx->stream));
    std::vector<int32_t> it(n_fits);
    PSK_HIP(ctx, hipMemcpyAsync(it.data(), b.iters, (size_t)n_fits * 4, hipMemcpyDeviceToHost, ctx->stream));
    PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (iters_out) memcpy(iters_out, it.data(), (size_t)n_fits * 4);
    return PSK_OK;
}
