// a10: the two L1 estimators the reference fits behind GridSearchCV (modeling.py:994-1014,
// :1075-1085, :1208-1216), as batched HIP solvers: every (grid value, CV fold) pair and the final
// refit is an independent small problem (N <= a few thousand samples x <= ~1000 selected k-mers),
// so one workgroup solves one fit and all fits of a grid search run in ONE launch.
//
//   logreg_newglmnet_*  liblinear's L1R_LR objective  ||w||_1 + |b| + C sum log(1+exp(-y(w.x+b)))
//                     (intercept = penalised constant-1 feature) by the improved GLMNET scheme liblinear
//                     itself uses: outer Newton steps, inner coordinate descent with shrinking on the
//                     quadratic model, one line search per outer step; a bit-packed form for 0/1 designs.
//   lasso_kernel      (1/2n)||y - Xw - b||^2 + alpha ||w||_1, unpenalised intercept, cyclic
//                     coordinate descent on the centred problem.
//
// Layout: XT[p][n] float (column-major: one k-mer's samples are contiguous), shared by all fits and
// L2-resident; per fit the linear predictor / residual lives in LDS (global scratch beyond 4096 samples).
// Latency-bound, f64 VALU; the roofline that matters for this stage is wall-clock, not bandwidth
// (DESIGN.md).  The host removes duplicated columns first (model.GridSearch): k-mers of one gene share
// one presence pattern, and an L1 optimum may put a pattern's weight on any one of its copies.
#include <algorithm>
#include <cerrno>

#include "solver_common.h"

namespace {

// L1 logistic regression by the improved GLMNET scheme (Yuan, Ho & Lin, JMLR 2012 -- the method behind
// liblinear's L1R_LR solver, which is what the reference's LogisticRegression(penalty='l1',
// solver='liblinear') runs): outer Newton iterations build a quadratic model from cached
// tau_i = C/(1+exp(w.x_i)) and D_i = C exp(w.x_i)/(1+exp(w.x_i))^2; an inner cyclic coordinate descent
// with shrinking solves the model WITHOUT any transcendental per coordinate; one Armijo line search per
// outer iteration (<= 20 halvings) needs exp/log once per sample.  Same stopping rule as liblinear:
// ||violation||_1 <= tol * min(#pos, #neg)/l * ||violation at w = 0||_1.
// Per-fit state: five feature arrays (w, w+d, diag H, grad, sum of C*x over the y=-1 rows) and five sample
// arrays (exp(w.x), its trial value, tau, D, x.d), in LDS when they fit (up to 160 KiB per workgroup).
#define FLD(ptr) (f_lds ? *(ptr) : lane0_load((ptr), lane))
__global__ __launch_bounds__(SV_THREADS) void logreg_newglmnet_kernel(
    const float *__restrict__ XT, const int8_t *__restrict__ ypm, const int32_t *__restrict__ fold, int n, int p,
    const double *__restrict__ fit_param, const int32_t *__restrict__ fit_fold, double tol, int max_newton,
    double *__restrict__ coef, double *__restrict__ icpt, int32_t *__restrict__ iters, double *__restrict__ work,
    int32_t *__restrict__ iwork, const int f_lds, const int s_lds)
{
    extern __shared__ double sm[];
    const int fit = blockIdx.x, lane = threadIdx.x;
    const double C = fit_param[fit];
    const int tf = fit_fold[fit];
    const int P1 = p + 1;
    double *gw = work + (size_t)fit * (5 * (size_t)P1 + 5 * (size_t)n);
    double *F = f_lds ? sm : gw;
    double *S = s_lds ? (sm + (f_lds ? 5 * P1 : 0)) : (gw + 5 * (size_t)P1);
    double *w = F, *wpd = F + P1, *Hd = F + 2 * P1, *Gr = F + 3 * P1, *xjneg = F + 4 * P1;
    double *ewx = S, *ewxn = S + n, *tau = S + 2 * (size_t)n, *D = S + 3 * (size_t)n, *xTd = S + 4 * (size_t)n;
    int32_t *act = iwork + (size_t)fit * P1;
    const double nu = 1e-12, sigma = 0.01;

    double npos = 0, nneg = 0;
    for (int i = lane; i < n; i += SV_THREADS) {
        ewx[i] = 1.0; tau[i] = C * 0.5; D[i] = C * 0.25; xTd[i] = 0.0;
        if (fold[i] != tf) { if (ypm[i] > 0) npos += 1; else nneg += 1; }
    }
    npos = psk_wave_sum_f64_dpp(npos);
    nneg = psk_wave_sum_f64_dpp(nneg);
    const double l = npos + nneg;
    double mn = npos < nneg ? npos : nneg;
    if (mn < 1.0) mn = 1.0;
    const double eps = tol * mn / (l > 0 ? l : 1.0);
    for (int j = 0; j < P1; j++) {  // feature arrays are written by lane 0 only (see lane0_load)
        const float *col = XT + (size_t)j * n;
        double sneg = 0.0;
        for (int i = lane; i < n; i += SV_THREADS)
            if (fold[i] != tf && ypm[i] < 0) sneg += C * ((j < p) ? (double)col[i] : 1.0);
        sneg = psk_wave_sum_f64_dpp(sneg);
        if (lane == 0) { w[j] = 0.0; wpd[j] = 0.0; xjneg[j] = sneg; act[j] = j; }
    }
    double w_norm = 0.0, Gmax_old = 1e300, Gnorm1_init = -1.0, inner_eps = 1.0;
    uint32_t rng = ((uint32_t)fit + 1u) * 2654435761u | 1u;  // per-fit xorshift state of the sweep permutations (lane 0's copy counts)
    int newton = 0, floor_steps = 0;   // floor_steps: Newton steps in a row taken inside the line search's rounding noise
    for (newton = 0; newton < max_newton; newton++) {
        double Gmax_new = 0.0, Gnorm1_new = 0.0;
        int active = P1;
        for (int sidx = 0; sidx < active; sidx++) {
            const int j = __builtin_amdgcn_readfirstlane(lane == 0 ? act[sidx] : 0);
            const float *col = XT + (size_t)j * n;
            double hd = 0.0, tmp = 0.0;
            for (int i = lane; i < n; i += SV_THREADS) {
                if (fold[i] == tf) continue;
                const double x = (j < p) ? (double)col[i] : 1.0;
                if (x == 0.0) continue;
                hd += x * x * D[i];
                tmp += x * tau[i];
            }
            hd = psk_wave_sum_f64_dpp(hd) + nu;
            tmp = psk_wave_sum_f64_dpp(tmp);
            const double grad = -tmp + FLD(&xjneg[j]);
            const double wj = FLD(&w[j]);
            const double Gp = grad + 1.0, Gn = grad - 1.0;
            double viol = 0.0;
            if (wj == 0.0) {
                if (Gp < 0) viol = -Gp;
                else if (Gn > 0) viol = Gn;
                else if (Gp > Gmax_old / l && Gn < -Gmax_old / l) {  // outer-level shrinking
                    active--;
                    if (lane == 0) { const int32_t t = act[sidx]; act[sidx] = act[active]; act[active] = t; }
                    sidx--;
                    continue;
                }
            } else if (wj > 0) viol = fabs(Gp);
            else viol = fabs(Gn);
            if (lane == 0) { Hd[j] = hd; Gr[j] = grad; }
            if (viol > Gmax_new) Gmax_new = viol;
            Gnorm1_new += viol;
        }
        if (newton == 0) Gnorm1_init = Gnorm1_new;
        if (Gnorm1_new <= eps * Gnorm1_init) break;

        // inner coordinate descent on the quadratic model
        for (int i = lane; i < n; i += SV_THREADS) xTd[i] = 0.0;
        double QP_Gmax_old = 1e300;
        int QP_active = active, iter = 0;
        while (iter < 1000) {
            // liblinear visits the active coordinates in a fresh random order every sweep (solve_l1r_lr); a fixed cyclic
            // order needs hundreds of times more sweeps on correlated columns (r01: 4.3 s against liblinear's 14 ms)
            if (lane == 0) {
                for (int jj = 0; jj + 1 < QP_active; jj++) {
                    rng ^= rng << 13; rng ^= rng >> 17; rng ^= rng << 5;
                    const int ii = jj + (int)(rng % (uint32_t)(QP_active - jj));
                    const int32_t tt = act[ii]; act[ii] = act[jj]; act[jj] = tt;
                }
            }
            double QP_Gmax_new = 0.0, QP_Gnorm1_new = 0.0;
            for (int sidx = 0; sidx < QP_active; sidx++) {
                const int j = __builtin_amdgcn_readfirstlane(lane == 0 ? act[sidx] : 0);
                const float *col = XT + (size_t)j * n;
                const double H = FLD(&Hd[j]);
                const double wp = FLD(&wpd[j]);
                double G = 0.0;
                for (int i = lane; i < n; i += SV_THREADS) {
                    if (fold[i] == tf) continue;
                    const double x = (j < p) ? (double)col[i] : 1.0;
                    if (x != 0.0) G += x * D[i] * xTd[i];
                }
                G = psk_wave_sum_f64_dpp(G) + FLD(&Gr[j]) + (wp - FLD(&w[j])) * nu;
                const double Gp = G + 1.0, Gn = G - 1.0;
                double viol = 0.0;
                if (wp == 0.0) {
                    if (Gp < 0) viol = -Gp;
                    else if (Gn > 0) viol = Gn;
                    else if (Gp > QP_Gmax_old / l && Gn < -QP_Gmax_old / l) {  // inner shrinking
                        QP_active--;
                        if (lane == 0) { const int32_t t = act[sidx]; act[sidx] = act[QP_active]; act[QP_active] = t; }
                        sidx--;
                        continue;
                    }
                } else if (wp > 0) viol = fabs(Gp);
                else viol = fabs(Gn);
                if (viol > QP_Gmax_new) QP_Gmax_new = viol;
                QP_Gnorm1_new += viol;
                double z;
                if (Gp < H * wp) z = -Gp / H;
                else if (Gn > H * wp) z = -Gn / H;
                else z = -wp;
                // liblinear skips |z| < 1e-12; a coefficient that has drifted to ~1e-17 must still be allowed to land
                // on exactly 0, otherwise its +-1 subgradient term keeps the violation above the stopping threshold
                // forever (r01: 5 of 143 fits of a 256 x 138 problem spun to max_iter)
                if (fabs(z) < 1e-12 && !(z == -wp && wp != 0.0)) continue;
                z = fmin(fmax(z, -10.0), 10.0);
                if (lane == 0) wpd[j] = wp + z;
                for (int i = lane; i < n; i += SV_THREADS) {
                    if (fold[i] == tf) continue;
                    const double x = (j < p) ? (double)col[i] : 1.0;
                    if (x != 0.0) xTd[i] += x * z;
                }
            }
            iter++;
            if (QP_Gnorm1_new <= inner_eps * Gnorm1_init) {
                if (QP_active == active) break;       // inner problem solved
                QP_active = active;                   // re-check the shrunk coordinates
                QP_Gmax_old = 1e300;
                continue;
            }
            QP_Gmax_old = QP_Gmax_new;
        }

        // line search along d = wpd - w
        double delta = 0.0, w_norm_new = 0.0;
        for (int j = 0; j < P1; j++) {
            const double wp = FLD(&wpd[j]);
            delta += FLD(&Gr[j]) * (wp - FLD(&w[j]));
            w_norm_new += fabs(wp);
        }
        delta += (w_norm_new - w_norm);
        double negsum = 0.0;
        for (int i = lane; i < n; i += SV_THREADS)
            if (fold[i] != tf && ypm[i] < 0) negsum += C * xTd[i];
        negsum = psk_wave_sum_f64_dpp(negsum);
        bool accepted = false, floor_rebuild = false;
        for (int ls = 0; ls < 20; ls++) {
            double cs = 0.0;
            for (int i = lane; i < n; i += SV_THREADS) {
                if (fold[i] == tf) continue;
                const double ex = exp(xTd[i]);
                const double en = ewx[i] * ex;
                ewxn[i] = en;
                cs += C * log((1.0 + en) / (ex + en));
            }
            const double cond = w_norm_new - w_norm + negsum - sigma * delta + psk_wave_sum_f64_dpp(cs);
            // liblinear accepts when cond <= 0.  Close to the optimum the decrease the model predicts sinks below what the sum of
            // l logarithms can resolve in doubles (~C l 2^-52): the sign of cond is then noise, twenty halvings only make the
            // step smaller, and a fit run at a tolerance near that floor repeats the same rejected step until max_iter (r04:
            // fits at tol = 1e-12 either stopped after ~100 Newton steps or never).  A step whose cond is inside the noise is
            // taken whole -- near the optimum the quadratic model is the better judge -- and exp(w.x) is then rebuilt from w
            // (below: accepting such steps without it, the multiplicatively updated copy drifted over thousands of them and one
            // fit "converged" 26 % away); three such steps in a row end the fit: it is where doubles can take it.  At the
            // tolerances the reference runs at |cond| is many orders above the noise and nothing changes.
            const bool in_noise = cond > 0.0 && cond <= 4.0 * 2.220446049250313e-16 * C * l;
            floor_steps = in_noise ? floor_steps + 1 : (cond <= 0.0 ? 0 : floor_steps);
            if (cond <= 0.0 || in_noise) {
                w_norm = w_norm_new;
                for (int j = 0; j < P1; j++) { if (lane == 0) w[j] = wpd[j]; }
                for (int i = lane; i < n; i += SV_THREADS) {
                    if (fold[i] == tf) continue;
                    const double en = ewxn[i];
                    const double tt = 1.0 / (1.0 + en);
                    ewx[i] = en; tau[i] = C * tt; D[i] = C * en * tt * tt;
                }
                accepted = true;
                floor_rebuild = in_noise;
                break;
            }
            w_norm_new = 0.0;
            for (int j = 0; j < P1; j++) {
                const double v = 0.5 * (FLD(&w[j]) + FLD(&wpd[j]));
                if (lane == 0) wpd[j] = v;
                w_norm_new += fabs(v);
            }
            delta *= 0.5;
            negsum *= 0.5;
            for (int i = lane; i < n; i += SV_THREADS) xTd[i] *= 0.5;
        }
        if (!accepted || floor_rebuild) {
            // the step was rejected 20 times: fall back to the current w and, as liblinear does after "too many
            // line search steps", rebuild exp(w.x) from w -- the multiplicatively updated copy has drifted and
            // the gradient computed from it would reject every further step (r01: fits spinning to max_iter)
            for (int j = 0; j < P1; j++) { if (lane == 0) wpd[j] = w[j]; }
            for (int i = lane; i < n; i += SV_THREADS) xTd[i] = 0.0;
            for (int j = 0; j < P1; j++) {
                const double wj = FLD(&w[j]);
                if (wj == 0.0) continue;
                const float *col = XT + (size_t)j * n;
                for (int i = lane; i < n; i += SV_THREADS) xTd[i] += wj * (double)col[i];
            }
            for (int i = lane; i < n; i += SV_THREADS) {
                if (fold[i] == tf) continue;
                const double en = exp(xTd[i]);
                const double tt = 1.0 / (1.0 + en);
                ewx[i] = en; tau[i] = C * tt; D[i] = C * en * tt * tt;
            }
        }
        if (iter == 1) inner_eps *= 0.25;
        if (floor_steps >= 3) { newton++; break; }   // at the floor of what doubles resolve (see the line search)
        Gmax_old = Gmax_new;
    }
    for (int j = 0; j < p; j++) { if (lane == 0) coef[(size_t)fit * p + j] = w[j]; }
    if (lane == 0) { icpt[fit] = w[p]; iters[fit] = newton; }
}
#undef FLD

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
    double yy = 0.0;   // y'y of the centred problem: scikit-learn's gap tolerance is tol y'y
    for (int i = lane; i < n; i += SV_THREADS) {
        const double d = (fold[i] != tf) ? (y[i] - ym) : 0.0;
        r[i] = d;
        yy += d * d;
    }
    yy = psk_wave_sum_f64_dpp(yy);
    const double an = alpha * ntrain, tol_s = tol * yy;
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
            const double mag = fabs(rho) - an;
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
        // scikit-learn's stop (enet_coordinate_descent): after a sweep whose largest step is below tol x the largest
        // coefficient -- or the last sweep allowed -- the duality gap is evaluated and the descent ends when gap < tol y'y
        if (wmax == 0.0 || dmax / wmax < tol || sweep == max_iter - 1) {
            double dual = 0.0, l1 = 0.0;
            for (int j = 0; j < p; j++) {
                const float *col = XT + (size_t)j * n;
                const double m = lane0_load(&xm[j], lane);
                double s = 0.0;
                for (int i = lane; i < n; i += SV_THREADS)
                    if (fold[i] != tf) s += ((double)col[i] - m) * r[i];
                s = psk_wave_sum_f64_dpp(s);
                dual = fmax(dual, fabs(s));
                l1 += fabs(lane0_load(&w[j], lane));
            }
            double rr = 0.0, ry = 0.0;
            for (int i = lane; i < n; i += SV_THREADS)
                if (fold[i] != tf) { rr += r[i] * r[i]; ry += r[i] * (y[i] - ym); }
            rr = psk_wave_sum_f64_dpp(rr);
            ry = psk_wave_sum_f64_dpp(ry);
            double cst = 1.0, gap;
            if (dual > an) { cst = an / dual; gap = 0.5 * (rr + rr * (cst * cst)); }
            else gap = rr;
            gap += an * l1 - cst * ry;
            if (gap < tol_s) { sweep++; break; }
        }
    }
    double acc = 0.0;
    for (int j = 0; j < p; j++) acc += lane0_load(&xm[j], lane) * lane0_load(&w[j], lane);
    if (lane == 0) { icpt[fit] = ym - acc; iters[fit] = sweep; }
}

// Lasso on a 0/1 design (presence / absence: the default), four waves per fit, the samples in registers -- the layout and
// the EXEC-masked sums of cd_coop.  With x in {0, 1} the coordinate step of the kernel above needs no pass over all the
// samples: the residual is kept as r_i = r'_i + c (c: one scalar for the -dd * mean terms every training sample gets), so
//   sum (x - m) r  =  (S1 + c cnt) - m (R + c n),   S1 = sum of r' over the samples that have the k-mer (a masked sum),
//   R = sum of r' over the training samples (kept up to date),  cnt = their number with the k-mer,
// and the update is r' -= dd on those samples (a masked add), c += dd m, R -= dd cnt.  Same cyclic order, same soft
// threshold, same stop as lasso_kernel; column means and norms from the counts (x^2 = x).  The float kernel read a
// column of n floats from L2 per coordinate (~2.6 us at 1,024 samples); here it is one transposed word per lane, asked
// for a step ahead.  LDS: w | mean | norm | count, p doubles each.
template <int WM>
__global__ __launch_bounds__(SV_COOP_THREADS) void lasso_bits_kernel(const uint64_t *__restrict__ colT, const double *__restrict__ y,
                                                                      const int32_t *__restrict__ fold, int n, int p, int W,
                                                                      const double *__restrict__ fit_param,
                                                                      const int32_t *__restrict__ fit_fold, double tol, int max_iter,
                                                                      double *__restrict__ coef, double *__restrict__ icpt,
                                                                      int32_t *__restrict__ iters)
{
    constexpr int WQ = WM / SV_COOP_WAVES;
    extern __shared__ double sm[];
    __shared__ double s_part[2][SV_COOP_WAVES];
    __shared__ double s_tot[2][SV_COOP_WAVES];
    double *w = sm, *mean = sm + p, *nrm = sm + 2 * (size_t)p, *cntd = sm + 3 * (size_t)p;
    uint32_t *cpart = reinterpret_cast<uint32_t *>(sm + 4 * (size_t)p);   // [wave][p] counts of this wave's samples
    const int fit = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, t0 = wave * WQ;
    const double alpha = fit_param[fit];
    const int tf = fit_fold[fit];
    // this wave's training samples and their y
    double R[WQ], Yc[WQ];
    uint64_t tmask = 0;
    double sy = 0.0, cn = 0.0;
#pragma unroll
    for (int q = 0; q < WQ; q++) {
        const int t = t0 + q, i = t * 64 + lane;
        const bool tr = t < W && i < n && fold[i] != tf;
        R[q] = tr ? y[i] : 0.0;
        if (tr) { tmask |= 1ull << t; sy += R[q]; cn += 1.0; }
    }
    sy = psk_wave_sum_f64_dpp(sy);
    cn = psk_wave_sum_f64_dpp(cn);
    if (lane == 0) { s_part[0][wave] = sy; s_tot[0][wave] = cn; }
    // this wave's share of every column's count
    for (int j = 0; j < p; j++) {
        const uint64_t x = colT[(size_t)j * 64 + lane] & tmask;
        uint32_t c = (uint32_t)__popcll(x);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d, 64);
        if (lane == 0) cpart[(size_t)wave * p + j] = c;
    }
    __syncthreads();
    double ysum = s_part[0][0], ntrain = s_tot[0][0];
#pragma unroll
    for (int v = 1; v < SV_COOP_WAVES; v++) { ysum += s_part[0][v]; ntrain += s_tot[0][v]; }
    const double ym = ysum / ntrain;
    double yyw = 0.0;
#pragma unroll
    for (int q = 0; q < WQ; q++) {
        if ((tmask >> (t0 + q)) & 1) R[q] -= ym;
        Yc[q] = R[q];            // the centred y of the training samples (0 elsewhere): R'y of the duality gap
        yyw += R[q] * R[q];
    }
    yyw = psk_wave_sum_f64_dpp(yyw);
    if (lane == 0) s_tot[1][wave] = yyw;
    for (int j = threadIdx.x; j < p; j += SV_COOP_THREADS) {
        uint32_t c = 0;
#pragma unroll
        for (int v = 0; v < SV_COOP_WAVES; v++) c += cpart[(size_t)v * p + j];
        const double s1 = (double)c, m = s1 / ntrain;
        cntd[j] = s1;
        mean[j] = m;
        nrm[j] = s1 - ntrain * m * m;   // sum (x - m)^2 with x^2 = x
        w[j] = 0.0;
    }
    // R' = sum of r' over the training samples: every wave sums its own, all of them add the four parts
    double Rw = 0.0;
#pragma unroll
    for (int q = 0; q < WQ; q++) Rw += R[q];
    Rw = psk_wave_sum_f64_dpp(Rw);
    if (lane == 0) s_part[1][wave] = Rw;
    __syncthreads();
    double Rp = s_part[1][0], yy = s_tot[1][0];
#pragma unroll
    for (int v = 1; v < SV_COOP_WAVES; v++) { Rp += s_part[1][v]; yy += s_tot[1][v]; }
    __syncthreads();
    const double an = alpha * ntrain, tol_s = tol * yy;
    double c = 0.0;
    int sweep = 0, visit = 0;
    auto col_t = [&](int j) { return (colT[(size_t)j * 64 + lane] & tmask) >> t0; };
    for (sweep = 0; sweep < max_iter; sweep++) {
        double dmax = 0.0, wmax = 0.0;
        uint64_t m_next = p > 0 ? col_t(0) : 0ull;
        for (int j = 0; j < p; j++) {
            const uint64_t x = m_next;
            if (j + 1 < p) m_next = col_t(j + 1);
            const double nj = nrm[j];
            if (!(nj > 1e-12)) continue;
            const double mj = mean[j], wj = w[j], cj = cntd[j];
            uint64_t M[WQ];
#pragma unroll
            for (int q = 0; q < WQ; q++) M[q] = __ballot((x >> q) & 1ull);
            double S = 0.0;
            if (WQ == 16) { masked_sum8(S, M, R); masked_sum8(S, M + (WQ == 16 ? 8 : 0), R + (WQ == 16 ? 8 : 0)); }
            else if (WQ == 8) masked_sum8(S, M, R);
            else masked_sum4(S, M, R);
            S = psk_wave_sum_f64_dpp(S);
            const int slot = visit & 1;
            visit++;
            if (lane == 0) s_part[slot][wave] = S;
            __syncthreads();
            double S1 = s_part[slot][0];
#pragma unroll
            for (int v = 1; v < SV_COOP_WAVES; v++) S1 += s_part[slot][v];
            const double rho = ((S1 + c * cj) - mj * (Rp + c * ntrain)) + nj * wj;
            const double mag = fabs(rho) - an;
            const double nw = (mag > 0.0) ? ((rho > 0 ? mag : -mag) / nj) : 0.0;
            const double dd = nw - wj;
            if (dd != 0.0) {
                const double nd = -dd;
                if (WQ == 16) { masked_add8(R, M, nd); masked_add8(R + (WQ == 16 ? 8 : 0), M + (WQ == 16 ? 8 : 0), nd); }
                else if (WQ == 8) masked_add8(R, M, nd);
                else masked_add4(R, M, nd);
                c += dd * mj;
                Rp -= dd * cj;
                if (threadIdx.x == 0) w[j] = nw;   // read again a sweep later, many barriers from here
            }
            if (fabs(dd) > dmax) dmax = fabs(dd);
            if (fabs(nw) > wmax) wmax = fabs(nw);
        }
        __syncthreads();   // w of this sweep is in place for the next one (and for the end)
        // scikit-learn's stop (see lasso_kernel): the duality gap, when it would evaluate it.  X'R of every column is one
        // more pass of masked sums (no steps), R'R and R'y come from the registers (r = r' + c on the training samples)
        if (wmax == 0.0 || dmax / wmax < tol || sweep == max_iter - 1) {
            double dual = 0.0, l1 = 0.0;
            uint64_t mg_next = p > 0 ? col_t(0) : 0ull;
            for (int j = 0; j < p; j++) {
                const uint64_t x = mg_next;
                if (j + 1 < p) mg_next = col_t(j + 1);
                uint64_t M[WQ];
#pragma unroll
                for (int q = 0; q < WQ; q++) M[q] = __ballot((x >> q) & 1ull);
                double S = 0.0;
                if (WQ == 16) { masked_sum8(S, M, R); masked_sum8(S, M + (WQ == 16 ? 8 : 0), R + (WQ == 16 ? 8 : 0)); }
                else if (WQ == 8) masked_sum8(S, M, R);
                else masked_sum4(S, M, R);
                S = psk_wave_sum_f64_dpp(S);
                const int slot = visit & 1;
                visit++;
                if (lane == 0) s_part[slot][wave] = S;
                __syncthreads();
                double S1 = s_part[slot][0];
#pragma unroll
                for (int v = 1; v < SV_COOP_WAVES; v++) S1 += s_part[slot][v];
                dual = fmax(dual, fabs((S1 + c * cntd[j]) - mean[j] * (Rp + c * ntrain)));
                l1 += fabs(w[j]);
            }
            double rr = 0.0, ry = 0.0;
#pragma unroll
            for (int q = 0; q < WQ; q++)
                if ((tmask >> (t0 + q)) & 1) { const double rv = R[q] + c; rr += rv * rv; ry += rv * Yc[q]; }
            rr = psk_wave_sum_f64_dpp(rr);
            ry = psk_wave_sum_f64_dpp(ry);
            __syncthreads();
            if (lane == 0) { s_part[0][wave] = rr; s_part[1][wave] = ry; }
            __syncthreads();
            rr = (s_part[0][0] + s_part[0][1]) + (s_part[0][2] + s_part[0][3]);
            ry = (s_part[1][0] + s_part[1][1]) + (s_part[1][2] + s_part[1][3]);
            __syncthreads();
            double cst = 1.0, gap;
            if (dual > an) { cst = an / dual; gap = 0.5 * (rr + rr * (cst * cst)); }
            else gap = rr;
            gap += an * l1 - cst * ry;
            if (gap < tol_s) { sweep++; break; }
        }
    }
    double acc = 0.0;
    for (int j = threadIdx.x; j < p; j += SV_COOP_THREADS) {
        coef[(size_t)fit * p + j] = w[j];
        acc += mean[j] * w[j];
    }
    acc = psk_wave_sum_f64_dpp(acc);
    if (lane == 0) s_tot[1][wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = s_tot[1][0];
#pragma unroll
        for (int v = 1; v < SV_COOP_WAVES; v++) a += s_tot[1][v];
        icpt[fit] = ym - a;
        iters[fit] = sweep;
    }
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
         *icpt = nullptr, *iters = nullptr, *work = nullptr, *iwork = nullptr, *bits = nullptr, *bitsT = nullptr, *ggq = nullptr;
    ~SolverBufs()
    {
        void *ps[] = {xt, y, fold, param, ffold, coef, icpt, iters, work, iwork, bits, bitsT, ggq};
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

// The PSK_* variables that pick kernel forms (docs/KNOBS.md: A/B runs and tests): a flag counts when it is set to
// anything but "" or "0"; an integer must parse completely and lie in its range -- atoi() made 0 of garbage and took any
// number, and a forced register form narrower than the design or a negative CG count are reachable from a user's shell.
bool env_flag(const char *name)
{
    const char *v = getenv(name);
    return v && *v && strcmp(v, "0") != 0;
}
int env_int(psk_ctx *ctx, const char *name, long lo, long hi, bool *set, int *out)
{
    const char *v = getenv(name);
    *set = false;
    if (!v || !*v) return PSK_OK;
    char *end = nullptr;
    errno = 0;
    const long x = strtol(v, &end, 10);
    if (errno || end == v || *end != '\0' || x < lo || x > hi)
        return psk_fail(ctx, PSK_EINVAL, "%s=%s: expected an integer in [%ld, %ld]", name, v, lo, hi);
    *set = true;
    *out = (int)x;
    return PSK_OK;
}

}  // namespace

#define SV_ALLOC(ptr, bytes) PSK_HIP(ctx, hipMalloc(&(ptr), (bytes) ? (bytes) : 8))

extern "C" int psk_logreg_l1_fit(psk_ctx *ctx, const float *X, const int32_t *y01, int n, int p, const int32_t *fold,
                                 const double *fit_param, const int32_t *fit_fold, int n_fits, double tol, int max_iter,
                                 double *coef_out, double *icpt_out, int32_t *iters_out)
{
    PSK_TRY(check_fit_args(ctx, X, y01, n, p, fold, fit_param, fit_fold, n_fits, coef_out, icpt_out));
    // form knobs, validated before anything is allocated
    const bool knob_no_cd_regs = env_flag("PSK_NO_CD_REGS"), knob_no_gram = env_flag("PSK_NO_GRAM"),
               knob_no_gram_global = env_flag("PSK_NO_GRAM_GLOBAL");
    bool set_min_p1 = false, set_wmreg = false, set_cg = false, set_reps = false, set_from = false;
    int knob_min_p1 = 0, knob_wmreg = 0, knob_cg_max = 16, knob_polish_reps = 0, knob_polish_from = 32;
    PSK_TRY(env_int(ctx, "PSK_GG_MIN_P1", 1, 1 << 20, &set_min_p1, &knob_min_p1));
    PSK_TRY(env_int(ctx, "PSK_FORCE_WMREG", 16, 64, &set_wmreg, &knob_wmreg));
    PSK_TRY(env_int(ctx, "PSK_CG_MAX", 0, 256, &set_cg, &knob_cg_max));
    PSK_TRY(env_int(ctx, "PSK_POLISH_REPS", -4096, 4096, &set_reps, &knob_polish_reps));
    PSK_TRY(env_int(ctx, "PSK_GG_POLISH_FROM", 1, 1000, &set_from, &knob_polish_from));
    if (set_wmreg && (knob_wmreg != 16 && knob_wmreg != 32 && knob_wmreg != 64))
        return psk_fail(ctx, PSK_EINVAL, "PSK_FORCE_WMREG=%d: the register forms hold 16, 32 or 64 sample words", knob_wmreg);
    if (set_wmreg && knob_wmreg * 64 < n)
        return psk_fail(ctx, PSK_EINVAL, "PSK_FORCE_WMREG=%d holds %d samples, the design has %d", knob_wmreg, knob_wmreg * 64, n);
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
    // presence/absence design (every entry 0 or 1)?  -> bit-packed kernel
    bool binary = n <= 4096;  // the bit-packed kernel keeps a column in one register per lane (64 words)
    for (size_t q = 0; binary && q < (size_t)n * p; q++) binary = (X[q] == 0.0f || X[q] == 1.0f);
    const int W = (n + 63) / 64;
    const size_t n_state = binary ? (size_t)W * 64 : (size_t)n;
    // placement of the per-fit state: everything in LDS when it fits the 160 KiB of a CU, else the sample
    // arrays only, else global scratch
    const size_t fbytes = 5 * (size_t)(p + 1) * 8, sbytes = 5 * n_state * 8, lds_max = 160 * 1024 - 8192;   // the kernels' static LDS (cooperation state of the bit-packed kernel: ~7 KB) comes on top
    int f_lds = 0, s_lds = 0;
    if (fbytes + sbytes <= lds_max) f_lds = s_lds = 1;
    else if (sbytes <= lds_max) s_lds = 1;
    else if (fbytes <= lds_max) f_lds = 1;
    const size_t lds = (f_lds ? fbytes : 0) + (s_lds ? sbytes : 0);
    SV_ALLOC(b.work, (size_t)n_fits * (fbytes + sbytes));
    SV_ALLOC(b.iwork, (size_t)n_fits * (p + 1) * 4);
    PSK_HIP(ctx, hipMemcpyAsync(b.y, ypm.data(), n, hipMemcpyHostToDevice, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(b.fold, fold, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(b.param, fit_param, (size_t)n_fits * 8, hipMemcpyHostToDevice, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(b.ffold, fit_fold, (size_t)n_fits * 4, hipMemcpyHostToDevice, ctx->stream));
    if (binary) {
        // LDS budget of the bit-packed kernel.  The inner QP works on the Gram block, the per-feature arrays and the
        // active list only, so those come first; the five sample arrays are streamed (coalesced) once per Newton
        // step and line-search trial and move to global scratch when they do not fit beside the Gram block of the
        // problem (thousands of samples: 5 x 2048 doubles are 80 KiB); the column bit words take what is left.
        // (With the sample arrays first, a 2048-sample fit with 170 distinct patterns had room for 79 Gram columns,
        // fell back to the array-form descent and took 0.5 s instead of 0.02 s, r01.)
        const size_t fa = fbytes + (((size_t)(p + 1) + 1) / 2) * 8, cbytes = (size_t)(p + 1) * W * 8;
        // More coordinates than the LDS Gram block takes (192): the Gram matrix of a fit in global memory, f32, a column
        // per slot (gg_run).  Its LDS arrays (slot parameters, ring, D in operand order, orders, flags) take the place of
        // the Gram block; the feature arrays must be in LDS beside them.
        const int P1 = p + 1, wmreg_h = knob_no_cd_regs ? 0 : (W <= 16 ? 16 : W <= 32 ? 32 : 64);
        int gg_sl = 0;
        size_t gg_stride = 0, gg_lds = 0;
        // up to 192 columns the LDS Gram form (exact f64 Hessian, columns built sample by sample) keeps the designs with
        // fewer than 1,024 samples: its builds are cheap there and an ill-conditioned fit at a tight tolerance converges in
        // fewer Newton steps than with the f32 / bf16-split Q of the global form (256 x 150 near-duplicates at tol = 1e-7:
        // inside 300 steps against not); from 1,024 samples on the global form is 3-10 x faster (2048 x 169 grid 0.27 ->
        // 0.03 s, 2000 x 150: 0.14 -> 0.012 s)
        const int gg_min_p1 = set_min_p1 ? knob_min_p1 : (n >= 1024 ? 64 : 192);
        if (P1 > gg_min_p1 && P1 <= 1024 && wmreg_h > 0 && SV_COOP_WAVES == 4 && !knob_no_gram && !knob_no_gram_global) {
            const size_t sl = 256 * (((size_t)P1 + 255) / 256), np_h = (size_t)W * 64;
            // (+ the build's tables / wave 3's counters, + the owners' column buffers)
            const size_t need = (4 * sl + np_h + sl / 2 + sl / 4 + sl / 8) * 8 + 8192 + 16384, stride = (((size_t)P1 + 15) / 16 * 16) * sl;
            if (fa + need <= lds_max && (size_t)n_fits * stride * 4 <= ((size_t)32 << 30)) {
                // (a device too full for the Gram matrices keeps the array form: slower, the same optimum)
                if (hipMalloc(&b.ggq, (size_t)n_fits * stride * 4) == hipSuccess) { gg_sl = (int)sl; gg_stride = stride; gg_lds = need; }
                else { (void)hipGetLastError(); b.ggq = nullptr; }
            }
        }
        const bool gram = !knob_no_gram && gg_sl == 0;
        const size_t pq = (size_t)(p + 1) < 192 ? (size_t)(p + 1) : 192;   // Gram columns the kernel can use
        const size_t need_q = gram ? pq * (pq + 1) / 2 * 8 : 0;             // its packed triangle
        size_t left = lds_max - gg_lds;
        f_lds = fa <= left ? 1 : 0; left -= f_lds ? fa : 0;
        // sample arrays: all five when they fit beside the whole Gram block; else only the two hot ones (tau, D) if
        // THAT makes room for the whole Gram block (thousands of samples, up to ~170 distinct patterns: the Gram form
        // with its accelerator converges where the array form runs into its sweep limit, r01: 2048 x 170, objective sum
        // of the grid 8.051e6 against 8.153e6, 0.73 s against 1.06 s); else all five again with a partial Gram block in
        // what is left (the previous behaviour); the hot ones alone when five do not fit at all
        const size_t hot_b = sbytes / 5 * 2;
        if (sbytes + need_q <= left) s_lds = 3;
        else if (gram && hot_b + need_q <= left) s_lds = 1;
        else if (sbytes <= left) s_lds = 3;
        else s_lds = hot_b <= left ? 1 : 0;
        const size_t s_in_lds = s_lds == 3 ? sbytes : (s_lds == 1 ? hot_b : 0);
        left -= s_in_lds;
        // the Gram block: up to the packed triangle of 192 features (a 64 x 64 or 128 x 128 square is preferred by the
        // kernel when it fits), before the column words
        size_t q_doubles = 0;
        if (gram) {
            const size_t square = pq <= 64 ? 64 * 64 : (pq <= 128 ? 128 * 128 : 0);
            q_doubles = need_q / 8;
            if (square * 8 <= left && square > q_doubles) q_doubles = square;
            if (q_doubles * 8 > left) q_doubles = left / 8;
            if (q_doubles < 36) q_doubles = 0;
        }
        left -= q_doubles * 8;
        const int c_lds = cbytes <= left ? 1 : 0; left -= c_lds ? cbytes : 0;
        if (gram && left >= 8) {   // leftover goes to the Gram block too (a square layout may now fit)
            const size_t most = (size_t)192 * 193 / 2;
            size_t more = q_doubles + left / 8;
            if (more > most) more = most;
            q_doubles = more;
        }
        if (gg_sl) q_doubles = gg_lds / 8;
        const size_t qbytes = q_doubles * 8;
        const int q_lds = q_doubles > 0;
        const size_t lds_b = s_in_lds + qbytes + (f_lds ? fa : 0) + (c_lds ? cbytes : 0);
        std::vector<uint64_t> bits((size_t)(p + 1) * W, 0);
        for (int i = 0; i < n; i++) {
            for (int j = 0; j < p; j++)
                if (X[(size_t)i * p + j] != 0.0f) bits[(size_t)j * W + (i >> 6)] |= 1ull << (i & 63);
            bits[(size_t)p * W + (i >> 6)] |= 1ull << (i & 63);  // constant-1 intercept column
        }
        SV_ALLOC(b.bits, bits.size() * 8);
        PSK_HIP(ctx, hipMemcpyAsync(b.bits, bits.data(), bits.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        // the same columns transposed for the register form of the descent: bit t of word l = sample 64 t + l
        std::vector<uint64_t> bitsT(!knob_no_cd_regs ? (size_t)(p + 1) * 64 : 0, 0);
        if (!bitsT.empty()) {
            for (int j = 0; j <= p; j++)
                for (int t = 0; t < W; t++) {
                    uint64_t x = bits[(size_t)j * W + t];
                    while (x) {
                        const int l = __builtin_ctzll(x);
                        x &= x - 1;
                        bitsT[(size_t)j * 64 + l] |= 1ull << t;
                    }
                }
            SV_ALLOC(b.bitsT, bitsT.size() * 8);
            PSK_HIP(ctx, hipMemcpyAsync(b.bitsT, bitsT.data(), bitsT.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        }
        const bool all_lds = f_lds && s_lds == 3 && c_lds && q_lds;
        const int wmreg = bitsT.empty() ? 0 : set_wmreg ? knob_wmreg : (W <= 16 ? 16 : W <= 32 ? 32 : 64);
        psk_l1_bits_launch L;
        L.bits = (const uint64_t *)b.bits; L.bitsT = (const uint64_t *)b.bitsT; L.ypm = (const int8_t *)b.y;
        L.fold = (const int32_t *)b.fold; L.fit_fold = (const int32_t *)b.ffold; L.fit_param = (const double *)b.param;
        L.n = n; L.p = p; L.W = W; L.n_fits = n_fits; L.max_iter = max_iter; L.tol = tol;
        L.coef = (double *)b.coef; L.icpt = (double *)b.icpt; L.work = (double *)b.work; L.iters = (int32_t *)b.iters; L.iwork = (int32_t *)b.iwork;
        L.f_lds = f_lds; L.s_lds = s_lds; L.c_lds = c_lds; L.q_doubles = (int)q_doubles;
        // CG steps per polish (the Gram-global form uses a quarter of them, at least 4, while steps are cut short early)
        L.cg_max = knob_cg_max;
        // polishes in a row while signs change; negative (the Gram-global form's default): as many as the descent has needed
        // sweeps when the accelerator is called (32, 64, 128), at most that many -- the 2048 x 169 grid 0.105 -> 0.031 s, the
        // 2048 x 907 grid 0.236 -> 0.245 s against a fixed 64
        L.polish_reps = set_reps ? knob_polish_reps : (gg_sl ? -128 : 64);
        L.gg_sl = gg_sl; L.gg_q = (float *)b.ggq; L.gg_stride = gg_stride;
        L.gg_polish_from = knob_polish_from;   // first polish of a descent after this many sweeps
        L.wmreg = wmreg; L.all_lds = all_lds ? 1 : 0; L.lds_bytes = lds_b; L.stream = ctx->stream;
        // two kernels, compiled apart (solver_l1_bits.h): the Gram matrix in global memory, or the LDS Gram block / array forms
        PSK_HIP(ctx, gg_sl ? psk_l1_bits_launch_gg(L) : psk_l1_bits_launch_gram(L));
        PSK_HIP(ctx, hipGetLastError());
        PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));  // `bits` (host) must outlive the copy
    } else {
        PSK_HIP(ctx, hipMemcpyAsync(b.xt, XT.data(), XT.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        if (lds > 64 * 1024)
            PSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(logreg_newglmnet_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        logreg_newglmnet_kernel<<<n_fits, SV_THREADS, lds, ctx->stream>>>(
            (const float *)b.xt, (const int8_t *)b.y, (const int32_t *)b.fold, n, p, (const double *)b.param,
            (const int32_t *)b.ffold, tol, max_iter, (double *)b.coef, (double *)b.icpt, (int32_t *)b.iters,
            (double *)b.work, (int32_t *)b.iwork, f_lds, s_lds);
    }
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
    // presence/absence design (every entry 0 or 1): the covariance form (solver_lasso.hip) up to 1,024 columns, else the
    // four-wave kernel on the bit-packed samples (per-column state in LDS); PSK_NO_LASSO_COV / PSK_NO_LASSO_BITS for A/B runs
    bool binary = n <= 4096 && !env_flag("PSK_NO_LASSO_BITS");
    for (size_t q = 0; binary && q < (size_t)n * p; q++) binary = (X[q] == 0.0f || X[q] == 1.0f);
    if (binary && p <= 1024 && !env_flag("PSK_NO_LASSO_COV")) {
        const int W = (n + 63) / 64, PP = 64 * ((p + 63) / 64);
        // the distinct held-out folds of the call: fits of one fold share its counts
        std::vector<int32_t> fold_ids, fidx(n_fits);
        for (int j = 0; j < n_fits; j++) {
            size_t q = 0;
            while (q < fold_ids.size() && fold_ids[q] != fit_fold[j]) q++;
            if (q == fold_ids.size()) fold_ids.push_back(fit_fold[j]);
            fidx[j] = (int32_t)q;
        }
        const int F = (int)fold_ids.size();
        if ((size_t)F * PP * PP * 2 <= ((size_t)2 << 30)) {
            std::vector<uint64_t> bits((size_t)PP * W, 0), tmask((size_t)F * W, 0);
            for (int i = 0; i < n; i++)
                for (int j = 0; j < p; j++)
                    if (X[(size_t)i * p + j] != 0.0f) bits[(size_t)j * W + (i >> 6)] |= 1ull << (i & 63);
            std::vector<double> yc((size_t)F * n, 0.0), fstat((size_t)F * 4, 0.0);
            for (int f = 0; f < F; f++) {
                double sy = 0.0, cnt = 0.0, yy = 0.0;
                for (int i = 0; i < n; i++)
                    if (fold[i] != fold_ids[f]) { sy += y[i]; cnt += 1.0; tmask[(size_t)f * W + (i >> 6)] |= 1ull << (i & 63); }
                if (cnt < 1.0) return psk_fail(ctx, PSK_EINVAL, "fold %d leaves no training sample", fold_ids[f]);
                const double ym = sy / cnt;
                for (int i = 0; i < n; i++)
                    if (fold[i] != fold_ids[f]) { const double d = y[i] - ym; yc[(size_t)f * n + i] = d; yy += d * d; }
                fstat[4 * f] = ym; fstat[4 * f + 1] = yy; fstat[4 * f + 2] = cnt;
            }
            // Workgroups go round the eight XCDs (workgroup b runs on XCD b % 8), each with its own 4-MB L2: the fits are
            // ordered by fold and XCD x takes a contiguous stretch of that order, so that the fits sharing an L2 read the
            // same one or two count matrices (1.6 MB each at 907 columns)
            std::vector<int32_t> order(n_fits);
            for (int j = 0; j < n_fits; j++) order[j] = j;
            std::stable_sort(order.begin(), order.end(), [&](int32_t u, int32_t v) { return fidx[u] < fidx[v]; });
            const int G = (n_fits + 7) / 8, n_blocks = 8 * G;
            std::vector<int32_t> block_fit(n_blocks, -1);
            for (int bq = 0; bq < n_blocks; bq++) {
                const int sidx = (bq % 8) * G + bq / 8;
                if (sidx < n_fits) block_fit[bq] = order[sidx];
            }
            void *d_bits = nullptr, *d_tmask = nullptr, *d_yc = nullptr, *d_fstat = nullptr, *d_C = nullptr, *d_Dg = nullptr, *d_q0 = nullptr,
                 *d_bf = nullptr, *d_fidx = nullptr;
            struct Free { std::vector<void **> v; ~Free() { for (void **q : v) if (*q) (void)hipFree(*q); } } fr;
            fr.v = {&d_bits, &d_tmask, &d_yc, &d_fstat, &d_C, &d_Dg, &d_q0, &d_bf, &d_fidx};
            SV_ALLOC(d_bits, bits.size() * 8);
            SV_ALLOC(d_tmask, tmask.size() * 8);
            SV_ALLOC(d_yc, yc.size() * 8);
            SV_ALLOC(d_fstat, fstat.size() * 8);
            SV_ALLOC(d_C, (size_t)F * PP * PP * 2);
            SV_ALLOC(d_Dg, (size_t)F * (PP / 64) * 4096 * 8);
            SV_ALLOC(d_q0, (size_t)F * PP * 8);
            SV_ALLOC(d_bf, (size_t)n_blocks * 4);
            SV_ALLOC(d_fidx, (size_t)n_fits * 4);
            PSK_HIP(ctx, hipMemcpyAsync(d_bits, bits.data(), bits.size() * 8, hipMemcpyHostToDevice, ctx->stream));
            PSK_HIP(ctx, hipMemcpyAsync(d_tmask, tmask.data(), tmask.size() * 8, hipMemcpyHostToDevice, ctx->stream));
            PSK_HIP(ctx, hipMemcpyAsync(d_yc, yc.data(), yc.size() * 8, hipMemcpyHostToDevice, ctx->stream));
            PSK_HIP(ctx, hipMemcpyAsync(d_fstat, fstat.data(), fstat.size() * 8, hipMemcpyHostToDevice, ctx->stream));
            PSK_HIP(ctx, hipMemcpyAsync(d_bf, block_fit.data(), (size_t)n_blocks * 4, hipMemcpyHostToDevice, ctx->stream));
            PSK_HIP(ctx, hipMemcpyAsync(d_fidx, fidx.data(), (size_t)n_fits * 4, hipMemcpyHostToDevice, ctx->stream));
            PSK_HIP(ctx, hipMemcpyAsync(b.param, fit_param, (size_t)n_fits * 8, hipMemcpyHostToDevice, ctx->stream));
            psk_lasso_cov_args A;
            A.bits = (const uint64_t *)d_bits; A.tmask = (const uint64_t *)d_tmask; A.yc = (const double *)d_yc;
            A.fstat = (const double *)d_fstat; A.C = (uint16_t *)d_C; A.Dg = (double *)d_Dg; A.q0 = (double *)d_q0;
            A.block_fit = (const int32_t *)d_bf; A.fit_param = (const double *)b.param; A.fit_fidx = (const int32_t *)d_fidx;
            A.n = n; A.p = p; A.PP = PP; A.W = W; A.n_folds = F; A.n_blocks = n_blocks; A.max_iter = max_iter; A.tol = tol;
            A.coef = (double *)b.coef; A.icpt = (double *)b.icpt; A.gaps = nullptr; A.iters = (int32_t *)b.iters; A.stream = ctx->stream;
            PSK_HIP(ctx, psk_lasso_cov_launch(A));
            PSK_HIP(ctx, hipMemcpyAsync(coef_out, b.coef, (size_t)n_fits * p * 8, hipMemcpyDeviceToHost, ctx->stream));
            PSK_HIP(ctx, hipMemcpyAsync(icpt_out, b.icpt, (size_t)n_fits * 8, hipMemcpyDeviceToHost, ctx->stream));
            std::vector<int32_t> itc(n_fits);
            PSK_HIP(ctx, hipMemcpyAsync(itc.data(), b.iters, (size_t)n_fits * 4, hipMemcpyDeviceToHost, ctx->stream));
            PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the host vectors must outlive their copies
            if (iters_out) memcpy(iters_out, itc.data(), (size_t)n_fits * 4);
            return PSK_OK;
        }
    }
    const size_t lds_bits = (size_t)p * (4 * 8 + SV_COOP_WAVES * 4);
    if (binary && lds_bits <= 150 * 1024) {
        const int W = (n + 63) / 64;
        std::vector<uint64_t> bitsT((size_t)p * 64, 0);   // word l of column j: bit t = sample 64 t + l
        for (int i = 0; i < n; i++)
            for (int j = 0; j < p; j++)
                if (X[(size_t)i * p + j] != 0.0f) bitsT[(size_t)j * 64 + (i & 63)] |= 1ull << (i >> 6);
        SV_ALLOC(b.bitsT, bitsT.size() * 8);
        PSK_HIP(ctx, hipMemcpyAsync(b.bitsT, bitsT.data(), bitsT.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        PSK_HIP(ctx, hipMemcpyAsync(b.y, y, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
        PSK_HIP(ctx, hipMemcpyAsync(b.fold, fold, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
        PSK_HIP(ctx, hipMemcpyAsync(b.param, fit_param, (size_t)n_fits * 8, hipMemcpyHostToDevice, ctx->stream));
        PSK_HIP(ctx, hipMemcpyAsync(b.ffold, fit_fold, (size_t)n_fits * 4, hipMemcpyHostToDevice, ctx->stream));
        auto kern = W <= 16 ? lasso_bits_kernel<16> : W <= 32 ? lasso_bits_kernel<32> : lasso_bits_kernel<64>;
        if (lds_bits > 64 * 1024)
            PSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bits));
        kern<<<n_fits, SV_COOP_THREADS, lds_bits, ctx->stream>>>((const uint64_t *)b.bitsT, (const double *)b.y, (const int32_t *)b.fold, n, p, W,
                                                                 (const double *)b.param, (const int32_t *)b.ffold, tol, max_iter,
                                                                 (double *)b.coef, (double *)b.icpt, (int32_t *)b.iters);
        PSK_HIP(ctx, hipGetLastError());
        PSK_HIP(ctx, hipMemcpyAsync(coef_out, b.coef, (size_t)n_fits * p * 8, hipMemcpyDeviceToHost, ctx->stream));
        PSK_HIP(ctx, hipMemcpyAsync(icpt_out, b.icpt, (size_t)n_fits * 8, hipMemcpyDeviceToHost, ctx->stream));
        std::vector<int32_t> itb(n_fits);
        PSK_HIP(ctx, hipMemcpyAsync(itb.data(), b.iters, (size_t)n_fits * 4, hipMemcpyDeviceToHost, ctx->stream));
        PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));   // bitsT (host) must outlive its copy
        if (iters_out) memcpy(iters_out, itb.data(), (size_t)n_fits * 4);
        return PSK_OK;
    }
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
