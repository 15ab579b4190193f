// L2-penalised estimators of the reference's `--penalty L2` branch:
//   continuous phenotype: sklearn Ridge(alpha)               set_model, modeling.py:1001-1002
//   binary phenotype:     LogisticRegression(penalty='l2')   set_model, modeling.py:1015-1019
// Both objectives are strictly convex, so the optimum is unique and any solver converged tightly
// reproduces scikit-learn's coefficients.  One workgroup per (grid value, fold) fit, all fits of a grid
// search in one launch (as in solver.hip); vectors live in a per-fit global scratch slice that stays in
// L2, the design matrix is kept in both orientations so that X v and X' u are coalesced.
//   Ridge:    conjugate gradients on (Xc'Xc + alpha I) w = Xc' yc with the training-fold centring sklearn
//             applies for fit_intercept=True (rank(Xc'Xc) <= n, so CG needs at most n+1 steps);
//   logistic: Newton steps, each solved by CG on the Hessian, Armijo backtracking.  The intercept is
//             unpenalised for lbfgs/newton-cg/sag/saga and a penalised constant-1 feature for liblinear
//             (get_logreg_solver, modeling.py:245-264).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <vector>

#include "dev_utils.h"
#include "psk_internal.h"

namespace {

constexpr int L2_THREADS = 256;

__device__ __forceinline__ double block_sum(double v, double *red)
{
    v = psk_wave_sum_f64_dpp(v);
    const int wave = threadIdx.x >> 6;
    __syncthreads();  // `red` may still be read from the previous reduction
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ __forceinline__ double block_dot(const double *a, const double *b, int len, double *red)
{
    double s = 0.0;
    for (int j = threadIdx.x; j < len; j += L2_THREADS) s += a[j] * b[j];
    return block_sum(s, red);
}

// out[i] = sum_j X[i][j] v[j] + vb for every sample i (XT is the [p][n] orientation)
__device__ __forceinline__ void mat_xv(const float *__restrict__ XT, int n, int p, const double *v, double vb, double *out)
{
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += L2_THREADS) {
        double s0 = vb, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int j = 0;
        for (; j + 3 < p; j += 4) {
            s0 += (double)XT[(size_t)j * n + i] * v[j];
            s1 += (double)XT[(size_t)(j + 1) * n + i] * v[j + 1];
            s2 += (double)XT[(size_t)(j + 2) * n + i] * v[j + 2];
            s3 += (double)XT[(size_t)(j + 3) * n + i] * v[j + 3];
        }
        for (; j < p; j++) s0 += (double)XT[(size_t)j * n + i] * v[j];
        out[i] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
}

// out[j] = sum_i X[i][j] u[i] for every feature j (X is the [n][p] orientation)
__device__ __forceinline__ void mat_xtu(const float *__restrict__ X, int n, int p, const double *u, double *out)
{
    __syncthreads();
    for (int j = threadIdx.x; j < p; j += L2_THREADS) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int i = 0;
        for (; i + 3 < n; i += 4) {
            s0 += (double)X[(size_t)i * p + j] * u[i];
            s1 += (double)X[(size_t)(i + 1) * p + j] * u[i + 1];
            s2 += (double)X[(size_t)(i + 2) * p + j] * u[i + 2];
            s3 += (double)X[(size_t)(i + 3) * p + j] * u[i + 3];
        }
        for (; i < n; i++) s0 += (double)X[(size_t)i * p + j] * u[i];
        out[j] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------
// Ridge: min ||y - Xw - b||^2 + alpha ||w||^2 over the training rows (sklearn Ridge, fit_intercept=True)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(L2_THREADS) void ridge_cg_kernel(
    const float *__restrict__ X, const float *__restrict__ XT, const double *__restrict__ y,
    const int32_t *__restrict__ fold, int n, int p, const double *__restrict__ fit_param,
    const int32_t *__restrict__ fit_fold, double *__restrict__ coef, double *__restrict__ icpt,
    int32_t *__restrict__ iters, double *__restrict__ work)
{
    __shared__ double red[4];
    const int fit = blockIdx.x, tid = threadIdx.x;
    const double alpha = fit_param[fit];
    const int tf = fit_fold[fit];
    double *mu = work + (size_t)fit * (5 * (size_t)p + 2 * (size_t)n);
    double *w = mu + p, *r = w + p, *d = r + p, *q = d + p, *u = q + p, *yc = u + n;

    double cnt = 0.0, ys = 0.0;
    for (int i = tid; i < n; i += L2_THREADS) {
        const double m = fold[i] != tf ? 1.0 : 0.0;
        u[i] = m;
        cnt += m;
        ys += m * y[i];
    }
    const double ntr = block_sum(cnt, red);
    const double ybar = block_sum(ys, red) / (ntr > 0 ? ntr : 1.0);
    mat_xtu(X, n, p, u, mu);
    for (int j = tid; j < p; j += L2_THREADS) { mu[j] /= (ntr > 0 ? ntr : 1.0); w[j] = 0.0; }
    for (int i = tid; i < n; i += L2_THREADS) yc[i] = fold[i] != tf ? y[i] - ybar : 0.0;
    mat_xtu(X, n, p, yc, r);  // b = Xc' yc = X' yc - mu * sum(yc), and sum(yc) = 0 over the training rows
    for (int j = tid; j < p; j += L2_THREADS) d[j] = r[j];
    __syncthreads();
    double rs = block_dot(r, r, p, red);
    const double b2 = rs;
    const int max_it = 4 * (n < p ? n : p) + 100;
    int it = 0;
    for (; it < max_it; it++) {
        if (!(rs > 1e-26 * b2)) break;
        const double mud = block_dot(mu, d, p, red);
        mat_xv(XT, n, p, d, -mud, u);  // u = Xc d
        double su = 0.0;
        for (int i = tid; i < n; i += L2_THREADS) {
            const double v = fold[i] != tf ? u[i] : 0.0;
            u[i] = v;
            su += v;
        }
        su = block_sum(su, red);
        mat_xtu(X, n, p, u, q);
        for (int j = tid; j < p; j += L2_THREADS) q[j] = q[j] - mu[j] * su + alpha * d[j];
        __syncthreads();
        const double dq = block_dot(d, q, p, red);
        if (!(dq > 0.0)) break;
        const double a = rs / dq;
        for (int j = tid; j < p; j += L2_THREADS) { w[j] += a * d[j]; r[j] -= a * q[j]; }
        __syncthreads();
        const double rs_new = block_dot(r, r, p, red);
        const double beta = rs_new / rs;
        for (int j = tid; j < p; j += L2_THREADS) d[j] = r[j] + beta * d[j];
        __syncthreads();
        rs = rs_new;
    }
    const double muw = block_dot(mu, w, p, red);
    for (int j = tid; j < p; j += L2_THREADS) coef[(size_t)fit * p + j] = w[j];
    if (tid == 0) { icpt[fit] = ybar - muw; iters[fit] = it; }
}

// ---------------------------------------------------------------------------------------------------
// L2 logistic regression: min 0.5 (w'w [+ b^2]) + C sum_i log(1 + exp(-y_i (x_i'w + b)))
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double log1pexp(double x)  // log(1 + e^x) without overflow
{
    return (x > 0 ? x : 0.0) + log1p(exp(-fabs(x)));
}

__global__ __launch_bounds__(L2_THREADS) void logreg_l2_newton_kernel(
    const float *__restrict__ X, const float *__restrict__ XT, const int8_t *__restrict__ ypm,
    const int32_t *__restrict__ fold, int n, int p, const double *__restrict__ fit_param,
    const int32_t *__restrict__ fit_fold, double tol, int max_newton, int pen_icpt, double *__restrict__ coef,
    double *__restrict__ icpt, int32_t *__restrict__ iters, double *__restrict__ work)
{
    __shared__ double red[4];
    const int fit = blockIdx.x, tid = threadIdx.x;
    const double C = fit_param[fit];
    const int tf = fit_fold[fit];
    double *w = work + (size_t)fit * (5 * (size_t)p + 4 * (size_t)n);
    double *g = w + p, *dw = g + p, *pw = dw + p, *qw = pw + p;  // gradient, Newton step, CG direction, H*direction
    double *z = qw + p, *Dv = z + n, *t = Dv + n, *xd = t + n;
    const double bpen = pen_icpt ? 1.0 : 0.0;

    for (int j = tid; j < p; j += L2_THREADS) w[j] = 0.0;
    double npos = 0.0, nneg = 0.0;
    for (int i = tid; i < n; i += L2_THREADS) {
        z[i] = 0.0;
        if (fold[i] != tf) { if (ypm[i] > 0) npos += 1.0; else nneg += 1.0; }
    }
    npos = block_sum(npos, red);
    nneg = block_sum(nneg, red);
    const double l = npos + nneg;
    double mn = npos < nneg ? npos : nneg;
    if (mn < 1.0) mn = 1.0;
    const double eps_ll = tol * mn / (l > 0 ? l : 1.0);  // liblinear's relative rule (linear.cpp, train_one)
    double b = 0.0, g0norm = 0.0;
    int newton = 0;
    for (; newton < max_newton; newton++) {
        // gradient pieces per sample: t = dloss/dz, Dv = d2loss/dz2 (held-out rows contribute nothing)
        double fsum = 0.0, gb = 0.0;
        for (int i = tid; i < n; i += L2_THREADS) {
            double ti = 0.0, di = 0.0;
            if (fold[i] != tf) {
                const double yi = (double)ypm[i];
                const double s = 1.0 / (1.0 + exp(yi * z[i]));  // probability of the wrong label
                ti = -C * yi * s;
                di = C * s * (1.0 - s);
                fsum += C * log1pexp(-yi * z[i]);
            }
            t[i] = ti;
            Dv[i] = di;
            gb += ti;
        }
        fsum = block_sum(fsum, red);
        gb = block_sum(gb, red) + bpen * b;
        mat_xtu(X, n, p, t, g);
        double g2 = 0.0, gmax = 0.0, ww = 0.0;
        for (int j = tid; j < p; j += L2_THREADS) {
            const double gj = g[j] + w[j];
            g[j] = gj;
            g2 += gj * gj;
            gmax = fmax(gmax, fabs(gj));
            ww += w[j] * w[j];
        }
        __syncthreads();
        g2 = block_sum(g2, red) + gb * gb;
        ww = block_sum(ww, red) + bpen * b * b;
        // max over the block via a sum of per-wave maxima is not a max: reduce it with a second pass
        double gm = gmax;
        for (int off = 32; off > 0; off >>= 1) gm = fmax(gm, __shfl_xor(gm, off));
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = gm;
        __syncthreads();
        gmax = fmax(fmax(fmax(red[0], red[1]), fmax(red[2], red[3])), fabs(gb));
        const double gnorm = sqrt(g2);
        if (newton == 0) g0norm = gnorm;
        const bool done = pen_icpt ? (gnorm <= eps_ll * g0norm) : (gmax <= tol * C);
        if (done || !(gnorm > 0.0)) break;
        const double f0 = 0.5 * ww + fsum;

        // CG on H d = -g,  H = I(+0 for an unpenalised intercept) + X~' D X~
        for (int j = tid; j < p; j += L2_THREADS) { dw[j] = 0.0; pw[j] = -g[j]; g[j] = -g[j]; }  // g now holds the residual
        __syncthreads();
        double db = 0.0, rb = -gb, pb = rb;
        double rs = g2;
        const int cg_max = (int)(l < p ? l : p) + 10;
        for (int cg = 0; cg < cg_max; cg++) {
            if (!(rs > 1e-20 * g2)) break;
            mat_xv(XT, n, p, pw, pb, xd);
            double sb = 0.0;
            for (int i = tid; i < n; i += L2_THREADS) {
                const double v = Dv[i] * xd[i];
                xd[i] = v;
                sb += v;
            }
            const double qb = block_sum(sb, red) + bpen * pb;
            mat_xtu(X, n, p, xd, qw);
            double pq = 0.0;
            for (int j = tid; j < p; j += L2_THREADS) {
                const double v = qw[j] + pw[j];
                qw[j] = v;
                pq += pw[j] * v;
            }
            __syncthreads();
            pq = block_sum(pq, red) + pb * qb;
            if (!(pq > 0.0)) break;
            const double a = rs / pq;
            double rn = 0.0;
            for (int j = tid; j < p; j += L2_THREADS) {
                dw[j] += a * pw[j];
                const double rj = g[j] - a * qw[j];
                g[j] = rj;
                rn += rj * rj;
            }
            db += a * pb;
            rb -= a * qb;
            __syncthreads();
            rn = block_sum(rn, red) + rb * rb;
            const double beta = rn / rs;
            for (int j = tid; j < p; j += L2_THREADS) pw[j] = g[j] + beta * pw[j];
            pb = rb + beta * pb;
            __syncthreads();
            rs = rn;
        }
        // directional derivative g'd with the ORIGINAL gradient: g_orig = w + X't (+b), recomputed cheaply as
        // -(H d + residual) is not exact, so evaluate it directly from its definition
        mat_xv(XT, n, p, dw, db, xd);  // xd = X~ d
        double gd = 0.0, wd = 0.0, dd = 0.0;
        for (int i = tid; i < n; i += L2_THREADS) gd += t[i] * xd[i];
        for (int j = tid; j < p; j += L2_THREADS) { wd += w[j] * dw[j]; dd += dw[j] * dw[j]; }
        gd = block_sum(gd, red);
        wd = block_sum(wd, red) + bpen * b * db;
        dd = block_sum(dd, red) + bpen * db * db;
        gd += wd;
        double step = 1.0;
        bool ok = false;
        for (int ls = 0; ls < 40; ls++) {
            double fs = 0.0;
            for (int i = tid; i < n; i += L2_THREADS)
                if (fold[i] != tf) fs += C * log1pexp(-(double)ypm[i] * (z[i] + step * xd[i]));
            fs = block_sum(fs, red);
            const double f1 = 0.5 * (ww + 2.0 * step * wd + step * step * dd) + fs;
            if (f1 <= f0 + 1e-4 * step * gd) { ok = true; break; }
            step *= 0.5;
        }
        if (!ok) break;
        for (int j = tid; j < p; j += L2_THREADS) w[j] += step * dw[j];
        for (int i = tid; i < n; i += L2_THREADS) z[i] += step * xd[i];
        b += step * db;
        __syncthreads();
    }
    for (int j = tid; j < p; j += L2_THREADS) coef[(size_t)fit * p + j] = w[j];
    if (tid == 0) { icpt[fit] = b; iters[fit] = newton; }
}

struct L2Bufs {
    void *x = nullptr, *xt = nullptr, *y = nullptr, *fold = nullptr, *param = nullptr, *ffold = nullptr, *coef = nullptr,
         *icpt = nullptr, *iters = nullptr, *work = nullptr;
    ~L2Bufs()
    {
        void *ps[] = {x, xt, y, fold, param, ffold, coef, icpt, iters, work};
        for (void *q : ps) if (q) (void)hipFree(q);
    }
};

int check_l2_args(psk_ctx *ctx, const void *X, const void *y, int n, int p, const int32_t *fold, const double *fit_param,
                  const int32_t *fit_fold, int n_fits, double *coef_out, double *icpt_out)
{
    if (!ctx) return PSK_EINVAL;
    if (!X || !y || !fold || !fit_param || !fit_fold || !coef_out || !icpt_out)
        return psk_fail(ctx, PSK_EINVAL, "null buffer");
    if (n < 2 || p < 1 || n_fits < 1) return psk_fail(ctx, PSK_EINVAL, "bad problem shape n=%d p=%d fits=%d", n, p, n_fits);
    return PSK_OK;
}

#define L2_ALLOC(ptr, bytes) PSK_HIP(ctx, hipMalloc(&(ptr), (bytes) ? (bytes) : 8))

// uploads the shared inputs; `work_doubles` = per-fit scratch length
int l2_upload(psk_ctx *ctx, L2Bufs &b, const float *X, int n, int p, const int32_t *fold, const double *fit_param,
              const int32_t *fit_fold, int n_fits, size_t work_doubles, std::vector<float> &XT)
{
    XT.resize((size_t)n * p);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < p; j++) XT[(size_t)j * n + i] = X[(size_t)i * p + j];
    L2_ALLOC(b.x, (size_t)n * p * 4);
    L2_ALLOC(b.xt, (size_t)n * p * 4);
    L2_ALLOC(b.fold, (size_t)n * 4);
    L2_ALLOC(b.param, (size_t)n_fits * 8);
    L2_ALLOC(b.ffold, (size_t)n_fits * 4);
    L2_ALLOC(b.coef, (size_t)n_fits * p * 8);
    L2_ALLOC(b.icpt, (size_t)n_fits * 8);
    L2_ALLOC(b.iters, (size_t)n_fits * 4);
    L2_ALLOC(b.work, (size_t)n_fits * work_doubles * 8);
    PSK_HIP(ctx, hipMemcpyAsync(b.x, X, (size_t)n * p * 4, hipMemcpyHostToDevice, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(b.xt, XT.data(), (size_t)n * p * 4, hipMemcpyHostToDevice, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(b.fold, fold, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(b.param, fit_param, (size_t)n_fits * 8, hipMemcpyHostToDevice, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(b.ffold, fit_fold, (size_t)n_fits * 4, hipMemcpyHostToDevice, ctx->stream));
    return PSK_OK;
}

int l2_download(psk_ctx *ctx, L2Bufs &b, int p, int n_fits, double *coef_out, double *icpt_out, int32_t *iters_out)
{
    PSK_HIP(ctx, hipGetLastError());
    std::vector<int32_t> it(n_fits);
    PSK_HIP(ctx, hipMemcpyAsync(coef_out, b.coef, (size_t)n_fits * p * 8, hipMemcpyDeviceToHost, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(icpt_out, b.icpt, (size_t)n_fits * 8, hipMemcpyDeviceToHost, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(it.data(), b.iters, (size_t)n_fits * 4, hipMemcpyDeviceToHost, ctx->stream));
    PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (iters_out) memcpy(iters_out, it.data(), (size_t)n_fits * 4);
    return PSK_OK;
}

}  // namespace

extern "C" int psk_ridge_fit(psk_ctx *ctx, const float *X, const double *y, int n, int p, const int32_t *fold,
                             const double *fit_param, const int32_t *fit_fold, int n_fits, double *coef_out,
                             double *icpt_out, int32_t *iters_out)
{
    PSK_TRY(check_l2_args(ctx, X, y, n, p, fold, fit_param, fit_fold, n_fits, coef_out, icpt_out));
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    L2Bufs b;
    std::vector<float> XT;
    PSK_TRY(l2_upload(ctx, b, X, n, p, fold, fit_param, fit_fold, n_fits, 5 * (size_t)p + 2 * (size_t)n, XT));
    L2_ALLOC(b.y, (size_t)n * 8);
    PSK_HIP(ctx, hipMemcpyAsync(b.y, y, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    ridge_cg_kernel<<<n_fits, L2_THREADS, 0, ctx->stream>>>(
        (const float *)b.x, (const float *)b.xt, (const double *)b.y, (const int32_t *)b.fold, n, p,
        (const double *)b.param, (const int32_t *)b.ffold, (double *)b.coef, (double *)b.icpt, (int32_t *)b.iters,
        (double *)b.work);
    return l2_download(ctx, b, p, n_fits, coef_out, icpt_out, iters_out);
}

extern "C" int psk_logreg_l2_fit(psk_ctx *ctx, const float *X, const int32_t *y01, int n, int p, const int32_t *fold,
                                 const double *fit_param, const int32_t *fit_fold, int n_fits, double tol, int max_iter,
                                 int penalise_intercept, double *coef_out, double *icpt_out, int32_t *iters_out)
{
    PSK_TRY(check_l2_args(ctx, X, y01, n, p, fold, fit_param, fit_fold, n_fits, coef_out, icpt_out));
    if (!(tol > 0.0) || max_iter < 1) return psk_fail(ctx, PSK_EINVAL, "tol must be > 0 and max_iter >= 1");
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    L2Bufs b;
    std::vector<float> XT;
    PSK_TRY(l2_upload(ctx, b, X, n, p, fold, fit_param, fit_fold, n_fits, 5 * (size_t)p + 4 * (size_t)n, XT));
    std::vector<int8_t> ypm(n);
    for (int i = 0; i < n; i++) ypm[i] = y01[i] ? 1 : -1;
    L2_ALLOC(b.y, (size_t)n);
    PSK_HIP(ctx, hipMemcpyAsync(b.y, ypm.data(), (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    logreg_l2_newton_kernel<<<n_fits, L2_THREADS, 0, ctx->stream>>>(
        (const float *)b.x, (const float *)b.xt, (const int8_t *)b.y, (const int32_t *)b.fold, n, p,
        (const double *)b.param, (const int32_t *)b.ffold, tol, max_iter, penalise_intercept ? 1 : 0, (double *)b.coef,
        (double *)b.icpt, (int32_t *)b.iters, (double *)b.work);
    return l2_download(ctx, b, p, n_fits, coef_out, icpt_out, iters_out);
}
