// The bit-packed L1-logistic kernel (0/1 designs: the default), ONE source for two kernels that are compiled apart so that
// neither carries the other's registers (as one kernel every four-wave instance needed 256 VGPR + 228-256
// AGPR and the 64-word instances spilled to scratch):
//   FORM 0 (solver_l1_gram.hip)  inner QP in covariance form on the Gram block in LDS (<= 192 active coordinates), else the
//                                array descent -- four waves with the samples in registers, or one wave from LDS;
//   FORM 1 (solver_l1_gg.hip)    the Gram matrix of a fit in global memory (gg_run: bf16-MFMA build, a visit divided over
//                                four waves, CG accelerator), the four-wave gradient pass, the array descent for < 64 active.
// The host picks the form per launch (psk_logreg_l1_fit); the algorithm around the QP -- liblinear's newGLMNET outer loop,
// shrinking, line search, stopping rule -- is the same text for both.
#pragma once
#include "solver_common.h"

namespace {

// Presence/absence designs (the default: 0/1 columns) take a bit-packed form of the same algorithm:
// column j is W = ceil(n/64) u64 words, sample i = bit (i & 63) of word (i >> 6), i.e. lane l owns exactly
// the bit-l samples.  A coordinate step loads its column with ONE coalesced load (lane t holds word t,
// AND-ed with the fit's training mask), then walks the words with scalar lane reads: no global load
// inside the loops, only the LDS-resident per-sample arrays.  (The float form pays an L2 round trip per
// 64 samples, ~7 us per coordinate at n = 2048 on a lone wave; this form ~0.3 us.)
#define FLD(ptr) (f_lds ? *(ptr) : lane0_load((ptr), lane))
// repeats of the Gram-global form's accelerator at a call after `sweeps` sweeps of the descent: polish_reps > 0: that many;
// polish_reps < 0: as many as the descent has needed sweeps so far, at most -polish_reps
__device__ __forceinline__ int gg_polish_repeats(int polish_reps, int sweeps)
{
    if (polish_reps >= 0) return polish_reps;
    return sweeps < -polish_reps ? sweeps : -polish_reps;
}
// ALL_LDS: every array of the fit lives in LDS (the usual case: a few hundred samples, <= ~1500 distinct
// columns).  The placement is then a compile-time fact, the pointers are LDS pointers and the loops use
// ds_read / ds_write; with run-time placement flags they are generic pointers and every access is a flat load.
// WMREG: 0, or the number of sample words per lane the register form of the descent holds (16, 32 or 64, a quarter of them in each of its four waves; see cd_coop) --
// a template parameter so that its 2 x WMREG doubles per lane do not weigh on the register allocation of the other forms.
// FORM: 0 = the LDS Gram block and the array forms, 1 = the Gram matrix in global memory (gg_run; four waves).
template <bool ALL_LDS, int WMREG, int FORM>
__global__ __launch_bounds__(WMREG > 0 ? SV_COOP_THREADS : SV_THREADS) void logreg_newglmnet_bits_kernel(
    const uint64_t *__restrict__ colbits, const int8_t *__restrict__ ypm, const int32_t *__restrict__ fold, int n, int p,
    int W, const double *__restrict__ fit_param, const int32_t *__restrict__ fit_fold, double tol, int max_newton,
    double *__restrict__ coef, double *__restrict__ icpt, int32_t *__restrict__ iters, double *__restrict__ work,
    int32_t *__restrict__ iwork, const int f_lds_rt, const int s_lds_rt, const int c_lds_rt, const int q_doubles_i, const int cg_max,
    const int polish_reps, const uint64_t *__restrict__ colT, const int gg_sl, float *__restrict__ gg_q, const size_t gg_stride,
    const int gg_polish_from)
{
    static_assert(FORM == 0 || WMREG > 0, "the Gram-global form runs on four waves");
    const size_t q_doubles = (size_t)q_doubles_i;  // LDS doubles reserved for the Gram block (or for the arrays of gg_run)
    // s_lds_rt: bit 0 = the two sample arrays every per-feature gradient pass reads (tau, D) are in LDS, bit 1 = the
    // other three (exp(w.x), its trial value, x.d: a few passes per Newton step) are
    const int f_lds = ALL_LDS ? 1 : f_lds_rt, s_mode = ALL_LDS ? 3 : s_lds_rt, c_lds = ALL_LDS ? 1 : c_lds_rt,
              q_lds = ALL_LDS ? 1 : (q_doubles_i > 0);
    extern __shared__ double sm_all[];
    double *Qm = sm_all;  // Gram block of the covariance-form QP: square when (64 NS)^2 fits, else packed triangle
    double *sm = sm_all + q_doubles;
    const int fit = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;   // waves 1..3 exist in the register form only (cd_coop)
    const double C = fit_param[fit];
    const int tf = fit_fold[fit];
#ifdef PSK_SV_STATS
    long long stat_sweeps = 0, stat_visits = 0, stat_t_cd = 0, stat_t0 = 0, stat_t_gram = 0, stat_t_build = 0, stat_builds = 0, stat_t_polish = 0, stat_gram_sweeps = 0;
    const long long stat_start = clock64();
#endif
    const int P1 = p + 1, NP = W * 64;  // sample arrays are padded to whole words
    double *gw = work + (size_t)fit * (5 * (size_t)P1 + 5 * (size_t)NP);
    // LDS layout: [feature arrays 5*P1 | active list P1 ints (padded)] if f_lds, [tau, D: 2*NP] if s_mode & 1,
    // [exp(w.x), its trial value, x.d: 3*NP] if s_mode & 2, [column bit words P1*W] if c_lds
    const size_t f_words = f_lds ? (5 * (size_t)P1 + (((size_t)P1 + 1) >> 1)) : 0;
    double *F = f_lds ? sm : gw;
    const bool h_lds = (s_mode & 1) != 0, o_lds = (s_mode & 2) != 0;
    double *Sg = gw + 5 * (size_t)P1;                                   // the global copies: ewx | ewxn | tau | D | xTd
    double *hot = sm + f_words, *oth = sm + f_words + (h_lds ? 2 * (size_t)NP : 0);
    const uint64_t *cb = colbits;
    if (c_lds) {
        uint64_t *lc = reinterpret_cast<uint64_t *>(sm + f_words + (h_lds ? 2 * (size_t)NP : 0) + (o_lds ? 3 * (size_t)NP : 0));
        if (wave == 0)
            for (size_t q = lane; q < (size_t)P1 * W; q += SV_THREADS) lc[q] = colbits[q];
        cb = lc;
    }
    double *w = F, *wpd = F + P1, *Hd = F + 2 * P1, *Gr = F + 3 * P1, *xjneg = F + 4 * P1;
    double *tau = h_lds ? hot : Sg + 2 * (size_t)NP, *D = h_lds ? hot + NP : Sg + 3 * (size_t)NP;
    double *ewx = o_lds ? oth : Sg, *ewxn = o_lds ? oth + NP : Sg + NP, *xTd = o_lds ? oth + 2 * (size_t)NP : Sg + 4 * (size_t)NP;
    int32_t *act = f_lds ? reinterpret_cast<int32_t *>(sm + 5 * (size_t)P1) : iwork + (size_t)fit * P1;
    const double nu = 1e-12, sigma = 0.01;

    uint32_t rng = ((uint32_t)fit + 1u) * 2654435761u | 1u;  // per-fit xorshift state of the sweep permutations (lane 0's copy counts)
    // the lane's training samples among its wave's words of the four-wave forms, transposed like colT (bit t = sample 64 t +
    // lane, t in [wave WQ, (wave + 1) WQ)): computed once -- every phase below took it from `fold` again, and the compiler
    // kept each phase's 16 loads alive across the whole Newton loop
    uint64_t tmask_w = 0;
    if (WMREG > 0) {
        constexpr int WQ0 = (WMREG > 0 ? WMREG : 32) / SV_COOP_WAVES;
        for (int q = 0; q < WQ0; q++) {
            const int t = wave * WQ0 + q, i = t * 64 + lane;
            if (t < W && i < n && fold[i] != tf) tmask_w |= 1ull << t;
        }
    }
    // what wave 0 hands to the other waves of the register form at the start of a descent, and their partial sums
    struct CdShared { double QP_Gmax_old, inner_eps, Gnorm1_init, l; int QP_active, active, cmd, fm; };
    __shared__ CdShared s_cd;
    __shared__ double s_part[2][SV_COOP_WAVES];
    __shared__ double s_qpart[SV_COOP_WAVES][192];   // build_coop: a wave's part of Q[k][m] for every slot k of the Gram block
    __shared__ int32_t s_fj[192];                    // ... and the feature of every slot (the QP uses `act` as scratch)
    // Column m of the Gram block, Q[k][m] = sum over the training samples that have both k-mers of D, on the fit's four
    // waves: wave v sums its quarter of the sample words for every slot k (column words transposed, the pair's AND as
    // EXEC masks, D of its samples in registers -- the operation of cd_coop's gradient pass) and the partial sums meet
    // in s_qpart.  One lane per slot walking the set bits of the AND through LDS (r01) took ~340,000 cycles per column:
    // 65-85 % of the covariance form's time at 2048 samples x 169 columns (per-fit statistics, r02).
    auto build_coop = [&](auto wm_tag) {
        constexpr int WM = decltype(wm_tag)::value, WQ = WM / SV_COOP_WAVES;
        const int t0 = wave * WQ, fm = s_cd.fm, na = s_cd.active;
        double Dq[WQ];
        const uint64_t tmask = tmask_w;
#pragma unroll
        for (int q = 0; q < WQ; q++) {
            const int t = t0 + q, i = t * 64 + lane;
            Dq[q] = t < W ? D[i] : 0.0;
        }
        const uint64_t mm = (colT[(size_t)fm * 64 + lane] & tmask) >> t0;
        uint64_t x_next = na > 0 ? mm & (colT[(size_t)s_fj[0] * 64 + lane] >> t0) : 0ull;
        for (int u = 0; u < na; u++) {
            const uint64_t x = x_next;
            if (u + 1 < na) x_next = mm & (colT[(size_t)s_fj[u + 1] * 64 + lane] >> t0);
            uint64_t M[WQ];
#pragma unroll
            for (int q = 0; q < WQ; q++) M[q] = __ballot((x >> q) & 1ull);
            double g = 0.0;
            if (WQ == 16) { masked_sum8(g, M, Dq); masked_sum8(g, M + (WQ == 16 ? 8 : 0), Dq + (WQ == 16 ? 8 : 0)); }
            else if (WQ == 8) masked_sum8(g, M, Dq);
            else if (WQ == 4) masked_sum4(g, M, Dq);
            else masked_sum2(g, M, Dq);
            g = psk_wave_sum_f64_dpp(g);
            if (lane == 0) s_qpart[wave][u] = g;
        }
        __syncthreads();   // the partial sums of every slot are in place
    };
    // Register form of the array descent, FOUR waves per fit (WMREG > 0).  A coordinate visit is ~450 instructions
    // when one wave does it, ~230 of them per-word work (32 words: masks, products, masked additions, the update);
    // a lone wave issues one instruction per 4 cycles, so the visit took ~2,800 cycles whatever else the CU had
    // free.  Here wave v owns words [v WM/4, (v + 1) WM/4) of every column -- D and x.d of those samples in its
    // registers -- sums its part of the gradient, and the four partial sums meet in LDS behind ONE barrier per
    // visit; every wave then takes the same decisions from the same bits (shrinking, step, stop), only wave 0
    // writes the shared state (coefficients, visiting order).  Waves 1..3 wait in helper_loop between descents.
    auto cd_coop = [&](auto wm_tag) -> int {
        constexpr int WM = decltype(wm_tag)::value, WQ = WM / SV_COOP_WAVES;
        const int t0 = wave * WQ;
        double QP_Gmax_old_c = s_cd.QP_Gmax_old;
        const double inner_eps_c = s_cd.inner_eps, Gnorm1_init_c = s_cd.Gnorm1_init, l_c = s_cd.l;
        int QP_active_c = s_cd.QP_active, iter_c = 0;
        const int active_c = s_cd.active;
        double Dr[WQ], Xr[WQ];
        const uint64_t tmask = tmask_w;
#pragma unroll
        for (int q = 0; q < WQ; q++) {
            const int t = t0 + q, i = t * 64 + lane;
            const bool in = t < W;
            Dr[q] = in ? D[i] : 0.0;
            Xr[q] = in ? xTd[i] : 0.0;
        }
        auto act_at = [&](int sx) { return f_lds ? act[sx] : __builtin_amdgcn_readfirstlane(lane == 0 ? act[sx] : 0); };
        auto col_t = [&](int j) { return (colT[(size_t)j * 64 + lane] & tmask) >> t0; };   // bit q = word t0 + q
        int visit = 0;
        while (iter_c < 1000) {
            __syncthreads();   // every wave has left the previous sweep: the order may change
            if (wave == 0 && lane == 0) {   // a fresh random visiting order every sweep, as liblinear's solve_l1r_lr
                for (int jj = 0; jj + 1 < QP_active_c; jj++) {
                    rng ^= rng << 13; rng ^= rng >> 17; rng ^= rng << 5;
                    const int ii = jj + (int)(rng % (uint32_t)(QP_active_c - jj));
                    const int32_t tt = act[ii]; act[ii] = act[jj]; act[jj] = tt;
                }
            }
            __syncthreads();
            double QP_Gmax_new = 0.0, QP_Gnorm1_new = 0.0;
            int j_next = QP_active_c > 0 ? act_at(0) : 0;
            uint64_t m_next = QP_active_c > 0 ? col_t(j_next) : 0ull;
            for (int sidx = 0; sidx < QP_active_c; sidx++) {
                const int j = j_next;
                const uint64_t m = m_next;
                if (sidx + 1 < QP_active_c) { j_next = act_at(sidx + 1); m_next = col_t(j_next); }
                const double H = FLD(&Hd[j]);
                const double wp = FLD(&wpd[j]);
                uint64_t M[WQ];   // word t0 + q of the column as a wave mask: the lanes whose sample has the k-mer
                double P[WQ];
#pragma unroll
                for (int q = 0; q < WQ; q++) {
                    M[q] = __ballot((m >> q) & 1ull);
                    P[q] = Dr[q] * Xr[q];
                }
                double Gw = 0.0;
                if (WQ == 16) { masked_sum8(Gw, M, P); masked_sum8(Gw, M + (WQ == 16 ? 8 : 0), P + (WQ == 16 ? 8 : 0)); }
                else if (WQ == 8) masked_sum8(Gw, M, P);
                else if (WQ == 4) masked_sum4(Gw, M, P);
                else masked_sum2(Gw, M, P);
                Gw = psk_wave_sum_f64_dpp(Gw);
                const int slot = visit & 1;   // two sets of partial sums: a wave may enter the next visit while another still reads
                visit++;
                if (lane == 0) s_part[slot][wave] = Gw;
                __syncthreads();
                double Gs = s_part[slot][0];
#pragma unroll
                for (int v = 1; v < SV_COOP_WAVES; v++) Gs += s_part[slot][v];
                const double G = Gs + FLD(&Gr[j]) + (wp - FLD(&w[j])) * nu;
                const double Gp = G + 1.0, Gn = G - 1.0;
                double viol = 0.0;
                if (wp == 0.0) {
                    if (Gp < 0) viol = -Gp;
                    else if (Gn > 0) viol = Gn;
                    else if (Gp > QP_Gmax_old_c / l_c && Gn < -QP_Gmax_old_c / l_c) {
                        QP_active_c--;
                        if (wave == 0 && lane == 0) { const int32_t tt = act[sidx]; act[sidx] = act[QP_active_c]; act[QP_active_c] = tt; }
                        __syncthreads();   // (every wave is here: the same G)
                        if (sidx < QP_active_c) { j_next = act_at(sidx); m_next = col_t(j_next); }   // swapped in: visited next
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
                if (fabs(z) < 1e-12 && !(z == -wp && wp != 0.0)) continue;  // see the LDS form below
                z = fmin(fmax(z, -10.0), 10.0);
                if (wave == 0 && lane == 0) wpd[j] = wp + z;
                if (WQ == 16) { masked_add8(Xr, M, z); masked_add8(Xr + (WQ == 16 ? 8 : 0), M + (WQ == 16 ? 8 : 0), z); }
                else if (WQ == 8) masked_add8(Xr, M, z);
                else if (WQ == 4) masked_add4(Xr, M, z);
                else masked_add2(Xr, M, z);
            }
            iter_c++;
            if (QP_Gnorm1_new <= inner_eps_c * Gnorm1_init_c) {
                if (QP_active_c == active_c) break;
                QP_active_c = active_c;
                QP_Gmax_old_c = 1e300;
                continue;
            }
            QP_Gmax_old_c = QP_Gmax_new;
        }
#pragma unroll
        for (int q = 0; q < WQ; q++)
            if (t0 + q < W) xTd[(t0 + q) * 64 + lane] = Xr[q];
        __syncthreads();   // x.d is whole again: wave 0 goes on to the line search, the others back to helper_loop
        return iter_c;
    };
    // ---- covariance form beyond the LDS Gram block ("gg": Gram matrix in global memory) ---------------------------------
    // More than 192 active coordinates: the array form pays a reduction over the samples per coordinate visit (~1,870
    // cycles on four waves at 2,048 samples; a 2,048 x 907 grid spent 3.3 s in its slowest fit, 98 % of it there).  Here
    // Q = X_A' D X_A is built ONCE per Newton step -- a real GEMM over 0/1 columns, on the bf16 matrix cores (32 samples per
    // product: a byte of a column's bit word becomes eight 1.0 / 0 through a table, D is the sum of two bf16 parts under the
    // other column's byte as a mask; f32 sums; upper triangle of 16 x 64 blocks, mirrored on the way out; ~5 M cycles at
    // 907 coordinates) -- and kept in global memory as f32 (slot-major: column m = SL floats, 3.7 MB for 907 coordinates;
    // the gradient vector g = Gr + Q d it updates stays f64).  A visit is then one column of Q, the step z, and
    // g += z Q[:, m].  What a visit costs is INSTRUCTIONS: a lone wave issues one every 5.5 cycles at best, 8 when it
    // depends on the one before, and pays 30 to 60 cycles for every branch it takes or v_cmp it branches on
    // (tools/_variants/ubench.hip, r03).  So the visits are divided over the fit's four waves, in lockstep -- one barrier per
    // TWO visits, nothing is polled -- and written without branches on the common path:
    //   wave 0    the steps: G of the slot, the soft-threshold step, the new coefficient (~55 instructions per pair);
    //   waves 1,2 own g -- slots 256 r + 4 lane + c, r in {0, 1} and {2, 3}, eight doubles per lane -- AND fetch the columns:
    //             the visiting order of a sweep is known in advance, so each keeps DEPTH tickets' columns (its rows of them)
    //             in registers and loads the next DEPTH in one burst per round; a column never passes through LDS on its way
    //             to g.  In an interval they apply the steps of the pair before and leave, for the NEXT pair's slots, g as
    //             it then is (s_set_gpr_idx + v_readlane) with the entries of Q that link those slots to the pair being
    //             stepped (from small LDS copies of their rows of the columns concerned); wave 0 completes G itself: steps
    //             and the updates they cause overlap;
    //   wave 3    liblinear's books, one interval behind: violations, the shrinking test, the stopping rule after a sweep's
    //             last visit; and the next sweep's random order, drawn whole when a sweep begins.
    // Same rule as liblinear for shrinking and stopping; a shrunk coordinate keeps its place in the order with 1 / H = 0 in
    // its place (its step is then exactly 0), so the order of a sweep never changes under the owners' feet.  The descent
    // runs in SEGMENTS: after sweeps 32, 64, 128, ... wave 3 ends one the way it ends the descent, g goes to LDS, the 256
    // threads take conjugate-gradient steps on the free set (gg_polish) and the next segment starts from that point.
    // 2,048 x 907 grid: 3.3 s -> 1.03 s -> 0.23 s with the accelerator (tests/golden/fit2048_907.npz); -DPSK_GG_CHECK
    // prints G against its definition every few hundred visits (1e-12).
    struct GgShared { int stop_at, last_A, par, iters, polish, nF, sweeps, nshrunk; uint32_t r32; double Gmax_old; };
    __shared__ GgShared s_gg;
#ifdef PSK_SV_STATS
    __shared__ long long s_stat_wait[4], s_stat_dead;   // (s_stat_dead: visits of the gg descent to slots that were shrunk out)
    if (threadIdx.x < 4) s_stat_wait[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_stat_dead = 0;
#endif
    if (FORM == 1 && threadIdx.x == 0) { s_gg.last_A = 0; s_gg.par = 0; s_gg.stop_at = -1; s_gg.polish = 0; s_gg.nF = 0; }   // (read behind the barrier that releases the first gg_run)
    const int SL = gg_sl;   // slots of the gg arrays (256 x ceil(P1 / 256)); 0 = form not available in this launch
    double *ggP = Qm;       // per slot: 1 / H (0 while the slot is shrunk out of the sweeps), w + d
    double *ggH = Qm + 2 * (size_t)SL;
    double *ggG = Qm + 3 * (size_t)SL;   // g of every slot between the segments of a descent (the owners keep it in registers within one)
    uint16_t *ggDs = reinterpret_cast<uint16_t *>(Qm + 4 * (size_t)SL);   // D of the training samples as two bf16 parts [2][NP]
    uint16_t *ggOrd = reinterpret_cast<uint16_t *>(Qm + 4 * (size_t)SL + NP);
    uint16_t *ggFeat = ggOrd + 2 * (size_t)SL;
    double *ggPub = reinterpret_cast<double *>(ggFeat + SL);   // [2][6]: for the pair of that interval's parity: g[c] | Q[c][a], Q[c][b] ; g[d] | Q[d][a], Q[d][b] ; Q[d][c]
    double *ggZG = ggPub + 12;                                 // [2][10]: wave 0's record of an interval (ggRec below)
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u4 *ggT1 = reinterpret_cast<u4 *>(reinterpret_cast<uint8_t *>(ggFeat + SL) + SL), *ggT2 = ggT1 + 256;   // byte -> eight bf16 ones / masks
    float *ggCol = reinterpret_cast<float *>(ggT2 + 256);   // [2 owner waves][2 parities][2 of a pair][64 lanes x 8]: a ticket's column, the owner's rows
    float *Qg = gg_q + (size_t)fit * gg_stride;
    // The accelerator of the LDS Gram form (polish, below) for this form, on the fit's 256 threads: with the signs of the
    // non-zero coordinates held fixed the model restricted to them is a plain quadratic; a few conjugate-gradient steps on
    // Q_FF delta = -(g_F + sign_F) give a descent direction, the step stops where the first coordinate would change sign
    // (it lands on exactly 0).  Thread t keeps slots 4 t ... 4 t + 3 of every vector; the direction is broadcast through LDS
    // and a product Q_{:,F} p reads the columns of F from global memory (rows of all slots: g of the others moves too).
    // Returns true when the step was cut short (the caller repeats).  Any point is a valid iterate of the descent.
    auto gg_polish = [&](int A, int cg_n, double &t_out) -> bool {
        const int tid = threadIdx.x;
        const bool on = 4 * tid < SL;   // thread t keeps slots 4 t ... 4 t + 3 of every vector (one 16-byte load per column of Q)
        typedef float f4 __attribute__((ext_vector_type(4)));
        double *pdL = reinterpret_cast<double *>(ggCol);          // the direction, by slot (the owners' column buffers rest)
        uint16_t *flist = reinterpret_cast<uint16_t *>(ggT1);     // the slots of F (wave 3's counters rest)
        int slot = 0;
        auto block_sum = [&](double x) {
            x = psk_wave_sum_f64_dpp(x);
            if (lane == 0) s_part[slot][wave] = x;
            __syncthreads();
            const double r = (s_part[slot][0] + s_part[slot][1]) + (s_part[slot][2] + s_part[slot][3]);
            slot ^= 1;
            return r;
        };
        bool inF[4];
        double sg[4], wp[4], r[4], pd[4], dl[4], Qd[4], Qp[4];
        double rs = 0.0;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int u = 4 * tid + e;
            const bool valid = u < A;
            wp[e] = valid ? ggP[2 * (size_t)u + 1] : 0.0;
            inF[e] = valid && wp[e] != 0.0;
            sg[e] = wp[e] > 0.0 ? 1.0 : -1.0;
            r[e] = inF[e] ? -(ggG[valid ? u : 0] + sg[e]) : 0.0;
            pd[e] = r[e];
            dl[e] = 0.0;
            Qd[e] = 0.0;
            rs += r[e] * r[e];
            if (on) pdL[u] = pd[e];
        }
        if (wave == 0) {   // F in slot order
            int cnt = 0;
            for (int base = 0; base < A; base += 64) {
                const int u = base + lane;
                const bool f = u < A && ggP[2 * (size_t)(u < A ? u : 0) + 1] != 0.0;
                const uint64_t m = __ballot(f);
                if (f) flist[cnt + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)u;
                cnt += __popcll(m);
            }
            if (lane == 0) s_gg.nF = cnt;
        }
        rs = block_sum(rs);   // (its barrier also publishes the direction and F)
        const double b2 = rs;
        t_out = 1.0;
        if (!(b2 > 0.0)) return false;
        const int nF = s_gg.nF;
        const float *Qt = Qg + (on ? 4 * tid : 0);
        for (int it = 0; it < cg_n; it++) {
#pragma unroll
            for (int e = 0; e < 4; e++) Qp[e] = 0.0;
            int i = 0;
            for (; i + 8 <= nF; i += 8) {   // eight columns requested together
                int v[8];
                double pv[8];
                f4 q[8];
#pragma unroll
                for (int c = 0; c < 8; c++) { v[c] = flist[i + c]; pv[c] = pdL[v[c]]; }
#pragma unroll
                for (int c = 0; c < 8; c++) q[c] = *reinterpret_cast<const f4 *>(Qt + (size_t)v[c] * SL);
#pragma unroll
                for (int c = 0; c < 8; c++)
#pragma unroll
                    for (int e = 0; e < 4; e++) Qp[e] = fma(pv[c], (double)q[c][e], Qp[e]);
            }
            for (; i < nF; i++) {
                const int v = flist[i];
                const double pv = pdL[v];
                const f4 q = *reinterpret_cast<const f4 *>(Qt + (size_t)v * SL);
#pragma unroll
                for (int e = 0; e < 4; e++) Qp[e] = fma(pv, (double)q[e], Qp[e]);
            }
            double pq = 0.0;
#pragma unroll
            for (int e = 0; e < 4; e++) pq += inF[e] ? pd[e] * Qp[e] : 0.0;
            pq = block_sum(pq);
            if (!(pq > 0.0)) break;
            const double a = rs / pq;
            double rn = 0.0;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                dl[e] += a * pd[e];
                Qd[e] += a * Qp[e];
                r[e] = inF[e] ? r[e] - a * Qp[e] : 0.0;
                rn += r[e] * r[e];
            }
            rn = block_sum(rn);   // (every thread has read the direction of this step by now)
            if (!(rn > 1e-18 * b2)) break;
            const double beta = rn / rs;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                pd[e] = r[e] + beta * pd[e];
                if (on) pdL[4 * tid + e] = pd[e];
            }
            rs = rn;
            __syncthreads();
        }
        // a direction that is not finite (breakdown of CG on a numerically singular block) is dropped
        double bad = 0.0;
#pragma unroll
        for (int e = 0; e < 4; e++) bad += (!(fabs(dl[e]) < 1e300) || !(fabs(Qd[e]) < 1e300)) ? 1.0 : 0.0;
        if (block_sum(bad) != 0.0) return false;
        // longest step in (0, 1] that keeps every sign
        double tmax = 1.0;
#pragma unroll
        for (int e = 0; e < 4; e++)
            if (inF[e] && (wp[e] + dl[e]) * sg[e] <= 0.0) tmax = fmin(tmax, -wp[e] / dl[e]);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) tmax = fmin(tmax, psk_shfl_xor_f64(tmax, d));
        if (lane == 0) s_part[slot][wave] = tmax;
        __syncthreads();
        tmax = fmin(fmin(s_part[slot][0], s_part[slot][1]), fmin(s_part[slot][2], s_part[slot][3]));
        slot ^= 1;
        if (!(tmax > 0.0)) return false;
        t_out = tmax;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int u = 4 * tid + e;
            if (u < A) {
                if (inF[e]) {
                    const bool hits = (wp[e] + dl[e]) * sg[e] <= 0.0 && -wp[e] / dl[e] <= tmax;
                    ggP[2 * (size_t)u + 1] = hits ? 0.0 : wp[e] + tmax * dl[e];
                }
                ggG[u] += tmax * Qd[e];
            }
        }
        __syncthreads();
        return tmax < 1.0;
    };
    // W0: the copy wave 0 calls from the Newton loop (the steps), or the one waves 1..3 call from helper_loop (owners, books):
    // two copies so that neither role's registers are live in the other's code
    auto gg_run = [&](auto w0_tag) __attribute__((always_inline)) -> int {
        constexpr bool W0 = decltype(w0_tag)::value;
#ifndef PSK_GG_DEPTH
#define PSK_GG_DEPTH 16
#endif
        constexpr int DEPTH = PSK_GG_DEPTH;   // (a power of two)
        typedef float f4 __attribute__((ext_vector_type(4)));
        typedef float f8 __attribute__((ext_vector_type(8)));
        typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
        typedef double d2 __attribute__((ext_vector_type(2)));
        typedef double d8 __attribute__((ext_vector_type(8)));
        const int A = s_cd.active, tid = threadIdx.x;
        const double inner_eps_c = s_cd.inner_eps, Gnorm1_init_c = s_cd.Gnorm1_init, l_c = s_cd.l;
#ifdef PSK_SV_STATS
        const long long stat_gg0 = clock64();
#endif
        {   // slot arrays, the first sweep's order, D of the training samples, the build's tables
            const bool keep = (A == s_gg.last_A);
            const int par = s_gg.par;
            for (int u = tid; u < SL; u += SV_COOP_THREADS) {
                const bool valid = u < A;
                const int f = valid ? act[u] : 0;
                const double hh = valid ? Hd[f] : 1.0;
                ggFeat[u] = (uint16_t)f;
                *reinterpret_cast<d2 *>(ggP + 2 * (size_t)u) = d2{1.0 / hh, valid ? w[f] : 0.0};
                ggH[u] = hh;
                ggG[u] = valid ? Gr[f] : 0.0;
                if (!keep) ggOrd[u] = (uint16_t)u;          // (the first Newton step, or the active set changed)
                else if (par) ggOrd[u] = ggOrd[SL + u];     // the last complete order of the previous Newton step
            }
            // D of the training samples as the sum of two bf16 (truncated: the weights the matrix cores see never exceed D, so
            // Q' + (H - diag Q') stays positive semidefinite), and the two byte -> eight-bf16 tables of the build
            for (int i = tid; i < NP; i += SV_COOP_THREADS) {
                const bool tr = i < n && fold[i] != tf;
                const float d32 = tr ? (float)D[i] : 0.f;
                const uint32_t b1 = __float_as_uint(d32) & 0xFFFF0000u;
                const float r32 = tr ? (float)(D[i] - (double)__uint_as_float(b1)) : 0.f;
                ggDs[i] = (uint16_t)(b1 >> 16);
                ggDs[NP + i] = (uint16_t)(__float_as_uint(r32 > 0.f ? r32 : 0.f) >> 16);
            }
            {
                const uint32_t b = tid & 255;
                uint32_t one[4], msk[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t lo = (b >> (2 * i)) & 1u, hi = (b >> (2 * i + 1)) & 1u;
                    one[i] = lo * 0x3F80u + hi * 0x3F800000u;
                    msk[i] = lo * 0xFFFFu + hi * 0xFFFF0000u;
                }
                if (tid < 256) {
                    ggT1[b] = u4{one[0], one[1], one[2], one[3]};
                    ggT2[b] = u4{msk[0], msk[1], msk[2], msk[3]};
                }
            }
            if (tid < 32) ggPub[tid] = 0.0;
            for (int i = tid; i < 4096; i += SV_COOP_THREADS) ggCol[i] = 0.f;
        }
        __syncthreads();
        {   // Q: tile (kb, mb) = 16 x 16 slots; a work item = tile row kb x four tile columns, items dealt round the waves.
            // v_mfma_f32_16x16x32_bf16 sums over 32 samples at a time: lane l holds, of its row (column) l & 15, the eight
            // samples 8 (l >> 4) ... + 7 of the chunk -- one BYTE of the column's bit word, expanded through the tables: 1.0 / 0
            // for the A side, D's two bf16 parts under the byte's mask for the B side (two products per tile and chunk).
            // (r03's first version fed v_mfma_f64_16x16x4 bit by bit: 27 M cycles per Newton step at 907 coordinates, a
            // seventh of a fit.)
            const int nt = (A + 15) >> 4, ng = (nt + 3) >> 2, kq = lane >> 4, li = lane & 15;
            int item = 0;
            for (int kb = 0; kb < nt; kb++)
                for (int gq = kb >> 2; gq < ng; gq++, item++) {
                    if ((item & (SV_COOP_WAVES - 1)) != wave) continue;
                    const int ka = kb * 16 + li;
                    const bool va = ka < A;
                    const uint64_t *pa = cb + (size_t)(va ? ggFeat[ka] : 0) * W;
                    const uint64_t *pb[4];
                    bool vb[4];
                    f4 acc[4];
                    uint64_t wa_n = va ? pa[0] : 0ull, wb_n[4];
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const int mc = (gq * 4 + c) * 16 + li;
                        vb[c] = mc < A;
                        pb[c] = cb + (size_t)(vb[c] ? ggFeat[mc] : 0) * W;
                        wb_n[c] = vb[c] ? pb[c][0] : 0ull;
                        acc[c] = f4{0.f, 0.f, 0.f, 0.f};
                    }
                    for (int t = 0; t < W; t++) {
                        const uint64_t wa = wa_n;
                        uint64_t wb[4];
#pragma unroll
                        for (int c = 0; c < 4; c++) wb[c] = wb_n[c];
                        if (t + 1 < W) {
                            wa_n = va ? pa[t + 1] : 0ull;
#pragma unroll
                            for (int c = 0; c < 4; c++) wb_n[c] = vb[c] ? pb[c][t + 1] : 0ull;
                        }
#pragma unroll
                        for (int c2 = 0; c2 < 2; c2++) {
                            const uint32_t ha = c2 ? (uint32_t)(wa >> 32) : (uint32_t)wa;
                            const u4 av = ggT1[(ha >> (8 * kq)) & 0xFFu];
                            const u4 d1 = *reinterpret_cast<const u4 *>(ggDs + t * 64 + c2 * 32 + kq * 8);
                            const u4 d2 = *reinterpret_cast<const u4 *>(ggDs + NP + t * 64 + c2 * 32 + kq * 8);
#pragma unroll
                            for (int c = 0; c < 4; c++) {
                                const uint32_t hb = c2 ? (uint32_t)(wb[c] >> 32) : (uint32_t)wb[c];
                                const u4 mk = ggT2[(hb >> (8 * kq)) & 0xFFu];
                                acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, av), __builtin_bit_cast(bf8, d1 & mk), acc[c], 0, 0, 0);
                                acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, av), __builtin_bit_cast(bf8, d2 & mk), acc[c], 0, 0, 0);
                            }
                        }
                    }
                    // C/D: column = lane & 15, rows 4 (lane >> 4) ... + 3
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const int mc = (gq * 4 + c) * 16 + li, kr = kb * 16 + 4 * kq;
                        if (mc >= nt * 16) continue;
                        *reinterpret_cast<f4 *>(Qg + (size_t)mc * SL + kr) = acc[c];
#pragma unroll
                        for (int r = 0; r < 4; r++) Qg[(size_t)(kr + r) * SL + mc] = acc[c][r];
                    }
                }
        }
        __syncthreads();
        for (int u = tid; u < A; u += SV_COOP_THREADS) Qg[(size_t)u * SL + u] = (float)ggH[u];   // (nu is on the diagonal of H)
        if (tid == 0) s_gg.stop_at = -1;
        __syncthreads();   // Q of this Newton step is whole and visible to the workgroup
#ifdef PSK_SV_STATS
        if (wave == 0) stat_t_build += clock64() - stat_gg0;
#endif
        int iter_c = 0;
#ifdef PSK_SV_STATS
        long long stat_bw = 0;   // this wave's cycles at the barrier of the descent (-> s_stat_wait, printed with the fit's statistics)
#define GG_BARRIER() do { const long long stat_b0 = clock64(); __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); \
                          __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); \
                          stat_bw += clock64() - stat_b0; } while (0)
#else
#define GG_BARRIER() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_s_barrier(); \
                          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); } while (0)
#endif
        // ---- TWO visits per barrier interval.  Interval I steps the slots of tickets a = 2 I and b = 2 I + 1 (ticket T = the
        // T-th visit of this descent = sweep T / A, position T % A).  What one interval costs beside its instructions -- the
        // barrier, an LDS round trip per hand-over, the drain of the writes before the barrier -- is paid once for two steps.
        // During interval I: wave 0 steps a and b from what the owners left in interval I - 1; the owners apply the steps of
        // pair I - 1 and leave, for pair I + 1 (tickets c, d), g[c], g[d] as they then are -- two steps behind c, three
        // behind d -- with the entries of Q that link them to a, b (and d to c); wave 3 books pair I - 1.
        double *ggRec = ggZG;   // [2][10]: z_a z_b | G_a G_b | (1/H, w)_a | (1/H, w)_b | slots a, b -- by the interval's parity
        // (wave 3's books that outlive a segment of the descent -- a segment ends where the accelerator is called, gg_polish
        // above -- rest in s_gg between segments)
        if (tid == 0) { s_gg.sweeps = 0; s_gg.nshrunk = 0; s_gg.Gmax_old = 1e300; s_gg.r32 = rng; }
        __syncthreads();
        for (;;) {   // segments of the descent
        if (!W0 && (wave == 1 || wave == 2)) {
            // ---- owners of g and of the columns.  Two register sets of DEPTH tickets: `a` is complete and used two tickets per
            // interval, `b` is loaded in ONE burst at the top of a round of DEPTH / 2 intervals and becomes `a` at its end,
            // so a column is DEPTH to 2 DEPTH - 1 visits old when it is due; the only place that waits for memory is the
            // copy, where the loads are a whole round old.  (A rotating single set, one load issued per visit, is what one
            // would write; the compiler's count of loads in flight does not survive the loop it makes of it -- rotated, exits
            // merged -- and it drained the queue, vmcnt(0), every visit.  For the same reason both groups are always loaded
            // -- clamped to the last group of the slot arrays, a duplicate at worst -- and the descent ends at a ticket that
            // is a multiple of DEPTH (a few more steps on the same model), so that this loop's only exit is at the top of a
            // round.)
            const int h = __builtin_amdgcn_readfirstlane(wave) - 1, rmax = (SL >> 8) - 1;
            const int r0 = min(2 * h, rmax), r1 = min(2 * h + 1, rmax);
            int dzo;
            asm volatile("v_mov_b32 %0, 0" : "=v"(dzo));   // (keeps the LDS reads of a pick per-lane loads: no scalar detour)
            d8 go;   // g of the slots 256 (2 h + (e >> 2)) + 4 lane + (e & 3)
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int u = 256 * (2 * h + (e >> 2)) + 4 * lane + (e & 3);
                go[e] = u < A ? ggG[u] : 0.0;
            }
            // a round's DEPTH tickets: their slots through ONE read of the order (lane u = ticket u of the round, across the
            // end of a sweep into the next order), then per ticket a lane read, the column's base and two loads
            const char *Qb = reinterpret_cast<const char *>(Qg);
            const uint32_t colb = (uint32_t)SL * 4u, vo0 = (uint32_t)(r0 * 64 + lane) * 16u, vo1 = (uint32_t)(r1 * 64 + lane) * 16u;
            int lk = 0, lpos = 0;
            auto round_order = [&]() __attribute__((always_inline)) {
                int pp = lpos + (lane & (DEPTH - 1)), kk = lk;
                if (pp >= A) { pp -= A; kk++; }
                const int ov = ggOrd[(kk & 1) * SL + pp];
                lpos += DEPTH;
                if (lpos >= A) { lpos -= A; lk++; }
                return ov;
            };
            auto issue = [&](f8 &x, int &mm, int ov, int u) __attribute__((always_inline)) {
                const int m = __builtin_amdgcn_readlane(ov, u);
                const char *col = Qb + (size_t)((uint32_t)m * colb);
                const f4 x0 = *reinterpret_cast<const f4 *>(col + vo0), x1 = *reinterpret_cast<const f4 *>(col + vo1);
                x = f8{x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                mm = m;
            };
            // this wave's rows of a ticket's column, in LDS for the picks: [parity of the interval that wrote it][first / second
            // of its pair]: interval I writes the columns of pair I + 1, those of pair I (written by I - 1) are still there
            auto colbuf = [&](int par, int which) __attribute__((always_inline)) { return ggCol + ((h * 2 + par) * 2 + which) * 512; };
            f8 sa_[DEPTH], sb_[DEPTH], l0 = f8{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, l1 = l0;   // l0, l1: the last pair of the round before
            int ma_[DEPTH], mb_[DEPTH];
            {
                const int ov = round_order();
#pragma unroll
                for (int u = 0; u < DEPTH; u++) issue(sa_[u], ma_[u], ov, u);
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the first set is whole
            *reinterpret_cast<f8 *>(colbuf(1, 0) + 8 * lane) = sa_[0];   // pair 0's columns, "written by interval -1"
            *reinterpret_cast<f8 *>(colbuf(1, 1) + 8 * lane) = sa_[1];
            // one round: the set `a` is used, the set `b` loaded; the two sets swap roles from round to round (no copy)
            auto round = [&](f8 (&a)[DEPTH], int (&ma)[DEPTH], f8 (&b)[DEPTH], int (&mb)[DEPTH], int T) __attribute__((always_inline)) -> bool {
                {
                    const int ov = round_order();
#pragma unroll
                    for (int u = 0; u < DEPTH; u++) issue(b[u], mb[u], ov, u);   // tickets T + DEPTH ... T + 2 DEPTH - 1
                }
#pragma unroll
                for (int j = 0; j < DEPTH / 2; j++) {   // interval (T + 2 j) / 2: tickets T + 2 j, T + 2 j + 1 are being stepped
                    GG_BARRIER();
                    if (j == 0 && T == __builtin_amdgcn_readfirstlane(s_gg.stop_at)) return true;
                    const int par = j & 1;   // (DEPTH / 2 is even: the parity of the interval is that of j)
                    const d2 zz = *reinterpret_cast<const d2 *>(ggRec + 10 * (par ^ 1));   // the steps of the pair before: asked for now, applied below
                    // the columns of the next pair go to LDS
                    const f8 cn0 = 2 * j + 2 < DEPTH ? a[2 * j + 2 < DEPTH ? 2 * j + 2 : 0] : b[0];
                    const f8 cn1 = 2 * j + 3 < DEPTH ? a[2 * j + 3 < DEPTH ? 2 * j + 3 : 0] : b[1];
                    const int mc = 2 * j + 2 < DEPTH ? ma[2 * j + 2 < DEPTH ? 2 * j + 2 : 0] : mb[0];
                    const int md = 2 * j + 3 < DEPTH ? ma[2 * j + 3 < DEPTH ? 2 * j + 3 : 0] : mb[1];
                    *reinterpret_cast<f8 *>(colbuf(par, 0) + 8 * lane) = cn0;
                    *reinterpret_cast<f8 *>(colbuf(par, 1) + 8 * lane) = cn1;
                    // the entries of Q for the next pair: rows c and d of the columns of a, b (and row d of c's), asked for now
                    const bool own_c = (mc >> 9) == h, own_d = (md >> 9) == h;
                    const int ec = ((mc >> 6) & 4) | (mc & 3), lc = (mc >> 2) & 63, ed = ((md >> 6) & 4) | (md & 3), ld = (md >> 2) & 63;
                    float q_ca = 0.f, q_cb = 0.f, q_da = 0.f, q_db = 0.f, q_dc = 0.f;
                    if (own_c) {
                        q_ca = colbuf(par ^ 1, 0)[8 * lc + ec + dzo];
                        q_cb = colbuf(par ^ 1, 1)[8 * lc + ec + dzo];
                    }
                    if (own_d) {
                        q_da = colbuf(par ^ 1, 0)[8 * ld + ed + dzo];
                        q_db = colbuf(par ^ 1, 1)[8 * ld + ed + dzo];
                        q_dc = colbuf(par, 0)[8 * ld + ed + dzo];
                    }
                    // the steps of the pair before, on this wave's rows of THEIR columns (a zero step takes the same
                    // instructions: a branch costs more than eight multiply-adds; explicit fma: -ffp-contract=off)
                    const f8 p0 = j > 0 ? a[j > 0 ? 2 * j - 2 : 0] : l0, p1 = j > 0 ? a[j > 0 ? 2 * j - 1 : 0] : l1;
#pragma unroll
                    for (int e = 0; e < 8; e++) go[e] = fma(zz[0], (double)p0[e], go[e]);
#pragma unroll
                    for (int e = 0; e < 8; e++) go[e] = fma(zz[1], (double)p1[e], go[e]);
                    // g of the next pair's slots as it is now: two steps behind c, three behind d
                    if (own_c) {
                        const double gv = psk_readlane_f64(go[ec], lc);
                        if (lane == 0) *reinterpret_cast<u4 *>(ggPub + 6 * par) = u4{(uint32_t)__double2loint(gv), (uint32_t)__double2hiint(gv), __float_as_uint(q_ca), __float_as_uint(q_cb)};
                    }
                    if (own_d) {
                        const double gv = psk_readlane_f64(go[ed], ld);
                        if (lane == 0) {
                            *reinterpret_cast<u4 *>(ggPub + 6 * par + 2) = u4{(uint32_t)__double2loint(gv), (uint32_t)__double2hiint(gv), __float_as_uint(q_da), __float_as_uint(q_db)};
                            reinterpret_cast<float *>(ggPub + 6 * par + 4)[0] = q_dc;
                        }
                    }
                }
                l0 = a[DEPTH - 2]; l1 = a[DEPTH - 1];
                return false;
            };
            for (int T = 0;; T += 2 * DEPTH) {
                if (round(sa_, ma_, sb_, mb_, T)) break;
                if (round(sb_, mb_, sa_, ma_, T + DEPTH)) break;
            }
            // the last burst is still in flight: nothing may leave this block with loads pending on registers that the code
            // after it reuses
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
            {   // the steps of the segment's last pair (wave 0's record of the interval before this round's top), then g goes
                // to LDS: the accelerator and the next segment start from it
                const d2 zz = *reinterpret_cast<const d2 *>(ggRec + 10 * ((DEPTH / 2 - 1) & 1));
#pragma unroll
                for (int e = 0; e < 8; e++) go[e] = fma(zz[0], (double)l0[e], go[e]);
#pragma unroll
                for (int e = 0; e < 8; e++) go[e] = fma(zz[1], (double)l1[e], go[e]);
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int u = 256 * (2 * h + (e >> 2)) + 4 * lane + (e & 3);
                    if (u < A) ggG[u] = go[e];
                }
            }
        } else if (!W0 && wave == 3) {
            // ---- liblinear's books, one interval behind the steps, and the next sweep's order.  The violations of the two
            // visits of pair I - 1, their shrinking tests (the marker 1 / H = 0 goes into the slot's place: the slot is visited
            // again a sweep later at the earliest) and, after a sweep's last visit, the stopping rule -- all from the record
            // wave 0 left: G of the visits and the slots' parameters as they were BEFORE the steps.  A verdict therefore takes
            // effect a few visits into the next sweep: the end of the descent at the next ticket that is a multiple of DEPTH
            // (a few more coordinate steps on the same model), the return of the shrunk slots and the new shrinking threshold
            // likewise.  The values pass through an address with an opaque zero added (see wave 0).
            int dz;
            asm volatile("v_mov_b32 %0, 0" : "=v"(dz));
            auto vmax = [](double x, double y) __attribute__((always_inline)) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
            uint32_t r32 = s_gg.r32;
            int sweeps = s_gg.sweeps, nshrunk = s_gg.nshrunk;
            int kq = 0, pq = -2;   // sweep and position of the FIRST visit of the pair being booked (pair I - 1; before the descent: none)
            double Gmax = 0.0, Gnorm1 = 0.0, Gmax_old = s_gg.Gmax_old, omt = 1.0 - Gmax_old / l_c;
            int stop_at = -1;
            bool gen = true;   // a sweep has begun whose successor's order is still to be drawn (the first sweep's, at once)
            for (int I = 0;; I++) {
                GG_BARRIER();
                if (2 * I == stop_at) break;
                const double *rec = ggRec + 10 * ((I + 1) & 1) + dz;
                const d2 GG = *reinterpret_cast<const d2 *>(rec + 2), Pa = *reinterpret_cast<const d2 *>(rec + 4), Pb = *reinterpret_cast<const d2 *>(rec + 6);
                const int sa = reinterpret_cast<const int *>(rec + 8)[0], sb = reinterpret_cast<const int *>(rec + 8)[1];
                bool sweep_ended = false;
#pragma unroll
                for (int v = 0; v < 2; v++) {   // the two visits of pair I - 1 (at I = 0 the record is zeros: 1 / H = 0, nothing)
                    const double G = GG[v], Hi = v ? Pb[0] : Pa[0], wp = v ? Pb[1] : Pa[1], aG = fabs(G);
                    const bool live = Hi != 0.0, zero = wp == 0.0;
#ifdef PSK_SV_STATS
                    if (lane == 0 && !live && I > 0) s_stat_dead++;
#endif
                    const bool shrink = live && zero && aG < omt;   // out of the sweeps until the whole set is taken up again
                    const double vz = vmax(aG - 1.0, 0.0), vn = fabs(G + copysign(1.0, wp));
                    const double viol = live ? (zero ? vz : vn) : 0.0;   // (0 for the visit that shrinks: |G| < 1)
                    Gmax = vmax(Gmax, viol);
                    Gnorm1 += viol;
                    nshrunk += shrink ? 1 : 0;
                    if (lane == 0 && shrink) ggP[2 * (size_t)(v ? sb : sa)] = 0.0;
                    if (__builtin_expect(I > 0 && pq + v == A - 1, 0)) {   // that was the last visit of a sweep: liblinear's rule
                        sweep_ended = true;
                        sweeps++;
                        bool stop = sweeps >= 1000;
                        const double gmax = psk_readlane_f64(Gmax, 0);
                        if (__builtin_amdgcn_readfirstlane((int)(Gnorm1 <= inner_eps_c * Gnorm1_init_c))) {
                            if (__builtin_amdgcn_readfirstlane(nshrunk) == 0) stop = true;
                            else {
                                nshrunk = 0;
                                Gmax_old = 1e300;
                                for (int u2 = lane; u2 < A; u2 += 64) ggP[2 * (size_t)u2] = 1.0 / ggH[u2];   // (wave 0 may have asked for one of these a moment ago: that visit is then skipped once more)
                            }
                        } else Gmax_old = gmax;
                        omt = 1.0 - Gmax_old / l_c;
                        Gmax = 0.0;
                        Gnorm1 = 0.0;
                        // the accelerator after sweeps gg_polish_from, 2 gg_polish_from, ... (powers of two): the segment ends like the descent
                        const bool pol = !stop && cg_max > 0 && sweeps >= gg_polish_from && (sweeps & (sweeps - 1)) == 0;
                        if ((stop || pol) && stop_at < 0) {
                            stop_at = __builtin_amdgcn_readfirstlane((2 * I + 2 + DEPTH) & ~(DEPTH - 1));   // > 2 I + 2: every wave reads it behind a later barrier
                            if (lane == 0) { s_gg.stop_at = stop_at; s_gg.iters = sweeps; s_gg.polish = pol ? 1 : 0; }
                        }
                    }
                }
                if (I > 0) {
                    pq += 2;
                    if (pq >= A) { pq -= A; kq++; }
                } else pq = 0;
                if (__builtin_expect(sweep_ended, 0)) gen = true;
                if (__builtin_expect(gen, 0)) {
                    // the order of the sweep after the one now under way, whole (a few thousand cycles, once per sweep): the slots
                    // sorted by a random 11-bit key -- histogram, offsets, scatter through LDS counters in the place of the
                    // build's tables; equal keys keep the counters' order.  (A Fisher-Yates step per visit, r03's first version,
                    // is two dependent LDS round trips: ~130 cycles of a wave that has the visits' bookkeeping to do.)  The sweep
                    // under way: the one the pair now being stepped (I) belongs to with its second visit.
                    gen = false;
                    int kc = kq, pc = pq + 1;   // pair I's second visit (pq is pair I's first by now)
                    if (pc >= A) { pc -= A; kc++; }
                    uint16_t *on = ggOrd + ((kc + 1) & 1) * SL;
                    const uint16_t *oc = ggOrd + (kc & 1) * SL;
                    uint32_t *bins = reinterpret_cast<uint32_t *>(ggT1);   // 2048 counters
                    r32 ^= r32 << 13; r32 ^= r32 >> 17; r32 ^= r32 << 5;
                    const uint32_t seed = (uint32_t)__builtin_amdgcn_readfirstlane((int)r32);
#pragma unroll
                    for (int i = 0; i < 32; i++) bins[lane + 64 * i] = 0u;
                    uint32_t keyv[16];
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        const int u = lane + 64 * e;
                        uint32_t x = ((uint32_t)u + 1u) * 0x9E3779B1u ^ seed;
                        x ^= x >> 15; x *= 0x85EBCA77u; x ^= x >> 13;
                        keyv[e] = x >> 21;
                        if (u < A) atomicAdd(&bins[keyv[e]], 1u);
                    }
                    uint32_t run = 0;   // this lane's 32 counters -> exclusive offsets
#pragma unroll
                    for (int i = 0; i < 32; i++) { const uint32_t c = bins[lane * 32 + i]; bins[lane * 32 + i] = run; run += c; }
                    const uint32_t base = psk_wave_incl_scan_u32(run, lane) - run;
#pragma unroll
                    for (int i = 0; i < 32; i++) bins[lane * 32 + i] += base;
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        const int u = lane + 64 * e;
                        if (u < A) on[atomicAdd(&bins[keyv[e]], 1u)] = (uint16_t)u;
                    }
                    // a sweep's first two slots are none of the previous sweep's last two (a pair's parameters are asked for
                    // while the pair before it is written)
                    if (lane == 0) {
                        const uint16_t t0 = oc[A - 1], t1 = oc[A - 2];
                        int cand = 2;
                        for (int q = 0; q < 2; q++)
                            while (on[q] == t0 || on[q] == t1) { const uint16_t x = on[q]; on[q] = on[cand]; on[cand] = x; cand++; }
                    }
                }
            }
            // the order of the sweep under way at the end is complete: the next segment / Newton step starts from it
            if (lane == 0) { s_gg.last_A = A; s_gg.par = kq & 1; s_gg.sweeps = sweeps; s_gg.nshrunk = nshrunk; s_gg.Gmax_old = Gmax_old; s_gg.r32 = r32; }
        } else if (W0) {
            // ---- the steps.  G of a slot = what its owner left (g some steps ago and the entries of Q that link it to the
            // slots stepped since) plus those steps; the slot's 1 / H and w, requested an interval ahead; the soft-threshold
            // form of liblinear's step: with u = w - G / H the minimiser of the one-variable model is
            // u - clamp(u, -1 / H, 1 / H) -- the same point as its three-way rule, exactly 0 when |u| <= 1 / H, from 5
            // instructions instead of 12.  (Its skip of steps below 1e-12 saves the array form a pass over the samples; here a
            // step costs the same whatever its size, so every step is taken.)  A shrunk slot has 1 / H = 0 in its place and
            // w = 0, and its step comes out as exactly 0 from the same arithmetic (u = 0, clamp(0, -0, 0) = 0); so does the
            // step of the visit that shrinks it (|G| < 1 - thr <= 1 means |u| <= 1 / H): this wave needs no flag at all --
            // violations, shrinking and the stopping rule are wave 3's.  The order arrives 64 entries at a time in a register
            // (lane i = the entry i places on).  Every lane computes the same steps; the values pass through an address with
            // an opaque zero added, so that the compiler takes them for lane-varying and builds selects: as wave-uniform
            // values it made ~20 scalar branches per visit of the rule, and a v_cmp -> s_cbranch pair costs a lone wave 30 to
            // 60 cycles (tools/_variants/ubench.hip, r03), a select 5.
            int dz;
            asm volatile("v_mov_b32 %0, 0" : "=v"(dz));
            // (v_max / v_min as they are: fmax() and fmin() first quieten their operands, an instruction each)
            auto vmax = [](double x, double y) __attribute__((always_inline)) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
            auto vmin = [](double x, double y) __attribute__((always_inline)) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
            auto step = [&](double G, double Hi, double wp) __attribute__((always_inline)) {
                const double u = fma(-G, Hi, wp);
                const double wnew = u - vmin(vmax(u, -Hi), Hi);
                return vmin(vmax(wnew - wp, -10.0), 10.0);
            };
            double zpa = 0.0, zpb = 0.0;   // the steps of the pair before
#ifdef PSK_GG_CHECK
            int gg_chk = 0;
#endif
            int wk = 0, wp0 = 0, wl = 2, wend = min(64, A);   // the window: sweep, position of lane 0's entry, next lane, lanes in use
            int ordv = ggOrd[min(lane, A - 1)];
            auto next_slot = [&]() __attribute__((always_inline)) {
                const int mnext = __builtin_amdgcn_readlane(ordv, wl);
                if (__builtin_expect(++wl == wend, 0)) {   // the window is used up: the next 64 entries of this sweep, or the head of the next sweep's order
                    wp0 += 64;
                    if (wp0 >= A) { wk++; wp0 = 0; }
                    wl = 0;
                    wend = min(64, A - wp0);
                    ordv = ggOrd[(wk & 1) * SL + min(wp0 + lane, A - 1)];
                }
                return mnext;
            };
            int m_a = __builtin_amdgcn_readlane(ordv, 0), m_b = __builtin_amdgcn_readlane(ordv, 1);
            d2 P_a = *reinterpret_cast<const d2 *>(ggP + 2 * (size_t)m_a + dz), P_b = *reinterpret_cast<const d2 *>(ggP + 2 * (size_t)m_b + dz);
            if (lane == 0) {   // pair 0: g[a], g[b] themselves, and the one entry of Q that links b to a
                *reinterpret_cast<d2 *>(ggPub + 6) = d2{ggG[m_a], 0.0};
                *reinterpret_cast<d2 *>(ggPub + 6 + 2) = d2{ggG[m_b], 0.0};
                reinterpret_cast<float *>(ggPub + 6 + 4)[0] = Qg[(size_t)m_a * SL + m_b];
            }
            // (a round of DEPTH / 2 intervals per pass of the loop, like the owners: the end of the descent is looked for once
            // per round, and the parity of an interval is a constant)
            auto interval = [&](int par) __attribute__((always_inline)) {
                // what the owners left for this pair in the interval before: parity par ^ 1
                const double *pub = ggPub + 6 * (par ^ 1) + dz;
                const d2 ea = *reinterpret_cast<const d2 *>(pub), eb = *reinterpret_cast<const d2 *>(pub + 2);
                const float q_ba = reinterpret_cast<const float *>(pub + 4)[0];
                const int m_c = next_slot(), m_d = next_slot();
                const d2 P_c = *reinterpret_cast<const d2 *>(ggP + 2 * (size_t)m_c + dz), P_d = *reinterpret_cast<const d2 *>(ggP + 2 * (size_t)m_d + dz);
                const double G_a = fma(zpb, (double)__int_as_float(__double2hiint(ea[1])), fma(zpa, (double)__int_as_float(__double2loint(ea[1])), ea[0]));
#ifdef PSK_GG_CHECK
                if (fit == 0 && (gg_chk++ % 509) == 0 && gg_chk < 30000) {   // G against its definition Gr[m] + sum_k Q[k][m] d_k
                    double acc = 0.0;
                    for (int kk = lane; kk < A; kk += 64) acc += (double)Qg[(size_t)m_a * SL + kk] * (ggP[2 * (size_t)kk + 1] - w[ggFeat[kk]]);
                    acc = psk_wave_sum_f64_dpp(acc) + Gr[ggFeat[m_a]];
                    if (lane == 0) printf("gg check a: interval %d slot %d G %.12e true %.12e diff %.3e\n", gg_chk, m_a, G_a, acc, G_a - acc);
                }
#endif
                const double z_a = step(G_a, P_a[0], P_a[1]);
                const double G_b = fma(z_a, (double)q_ba, fma(zpb, (double)__int_as_float(__double2hiint(eb[1])), fma(zpa, (double)__int_as_float(__double2loint(eb[1])), eb[0])));
#ifdef PSK_GG_CHECK
                if (fit == 0 && (gg_chk % 509) == 1 && gg_chk < 30000) {   // the same for b: w of slot a is not written yet, its step is added by hand
                    double acc = 0.0;
                    for (int kk = lane; kk < A; kk += 64) acc += (double)Qg[(size_t)m_b * SL + kk] * (ggP[2 * (size_t)kk + 1] + (kk == m_a ? z_a : 0.0) - w[ggFeat[kk]]);
                    acc = psk_wave_sum_f64_dpp(acc) + Gr[ggFeat[m_b]];
                    if (lane == 0) printf("gg check b: interval %d slot %d G %.12e true %.12e diff %.3e\n", gg_chk, m_b, G_b, acc, G_b - acc);
                }
#endif
                const double z_b = step(G_b, P_b[0], P_b[1]);
                if (lane == 0) {
                    double *rec = ggRec + 10 * par;
                    *reinterpret_cast<d2 *>(rec) = d2{z_a, z_b};
                    *reinterpret_cast<d2 *>(rec + 2) = d2{G_a, G_b};
                    *reinterpret_cast<d2 *>(rec + 4) = P_a;
                    *reinterpret_cast<d2 *>(rec + 6) = P_b;
                    reinterpret_cast<int *>(rec + 8)[0] = m_a;
                    reinterpret_cast<int *>(rec + 8)[1] = m_b;
                    ggP[2 * (size_t)m_a + 1] = P_a[1] + z_a;
                    ggP[2 * (size_t)m_b + 1] = P_b[1] + z_b;
                }
                zpa = z_a; zpb = z_b;
                P_a = P_c; P_b = P_d;
                m_a = m_c; m_b = m_d;
            };
            for (int T = 0;; T += DEPTH) {
#pragma unroll
                for (int j = 0; j < DEPTH / 2; j++) {
                    GG_BARRIER();
                    if (j == 0 && T == __builtin_amdgcn_readfirstlane(s_gg.stop_at)) goto steps_done;
                    interval(j & 1);
                }
            }
        steps_done:;
        }
        __syncthreads();   // the segment is over for every wave: g, w + d and the order under way are in LDS
        if (!s_gg.polish) break;
        {
#ifdef PSK_SV_STATS
            const long long stat_p0 = clock64();
#endif
            // long CG runs pay when the step they buy is taken whole; while steps are cut short early (many small coefficients
            // on their way to zero: each repeat lands one of them) short runs with many repeats do (2048 x 907 grid: 4 steps
            // x 64 repeats 0.23 s, 6 x 64 0.32 s; 1500 x 400: 16 steps 0.37 s, 4 steps 1.25 s; near-separable 256 x 70: 16
            // steps 65 Newton steps at most, 4 steps 350)
#ifndef PSK_GG_TDEEP
#define PSK_GG_TDEEP 0.5   // a step of at least this share of the way: the next CG run is a long one
#endif
            double t_last = 1.0;
            const int cg_short = cg_max < 4 ? cg_max : (cg_max / 4 > 4 ? cg_max / 4 : 4);
            const int reps_now = gg_polish_repeats(polish_reps, s_gg.iters);
            for (int rep = 0; rep < reps_now && gg_polish(A, t_last >= PSK_GG_TDEEP ? cg_max : cg_short, t_last); rep++) {}
            // the next segment: a new descent from this point, in the order of the sweep that was under way
            const int par = s_gg.par;
            for (int u = tid; u < SL; u += SV_COOP_THREADS)
                if (par) ggOrd[u] = ggOrd[SL + u];
            if (tid < 32) ggPub[tid] = 0.0;
            for (int i = tid; i < 4096; i += SV_COOP_THREADS) ggCol[i] = 0.f;
            if (tid == 0) { s_gg.stop_at = -1; s_gg.polish = 0; }
            __syncthreads();
#ifdef PSK_SV_STATS
            if (wave == 0) stat_t_polish += clock64() - stat_p0;
#endif
        }
        }   // segments
        rng = s_gg.r32;
        // back to the feature arrays
        if (wave == 0)
            for (int u = lane; u < A; u += 64) wpd[ggFeat[u]] = ggP[2 * (size_t)u + 1];
#undef GG_BARRIER
#ifdef PSK_SV_STATS
        if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long *>(&s_stat_wait[wave]), (unsigned long long)stat_bw);
#endif
        __syncthreads();
        iter_c = s_gg.iters;
        {   // x.d = X_A d for the line search: every wave its words of the samples (transposed columns, eight loads in
            // flight), the additions of a sample in the order of the active list as the one-wave loop had them
            constexpr int WMc = WMREG > 0 ? WMREG : 32, WQ = WMc / SV_COOP_WAVES;
            const int t0 = wave * WQ;
            const uint64_t tmask = tmask_w;
            double Xr[WQ];
#pragma unroll
            for (int q = 0; q < WQ; q++) Xr[q] = 0.0;
            for (int u0 = 0; u0 < A; u0 += 8) {
                uint64_t x[8];
                double dd[8];
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    const int u = u0 + c < A ? u0 + c : A - 1;
                    const int f = ggFeat[u];
                    dd[c] = u0 + c < A ? ggP[2 * (size_t)u + 1] - w[f] : 0.0;
                    x[c] = (colT[(size_t)f * 64 + lane] & tmask) >> t0;
                }
#pragma unroll
                for (int c = 0; c < 8; c++)
#pragma unroll
                    for (int q = 0; q < WQ; q++) Xr[q] += ((x[c] >> q) & 1ull) ? dd[c] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < WQ; q++)
                if (t0 + q < W) xTd[(t0 + q) * 64 + lane] = Xr[q];
            __syncthreads();
        }
        return iter_c;
    };
    // The gradient pass at the top of a Newton step on the fit's four waves (gg form): wave v sums D and tau over its
    // words of every column (build_coop's masks), SL / 4 columns between barriers, and the parts meet in the slot arrays
    // of the gg area (free between descents: the visiting order kept for the next descent lies behind them).
    // Wave 0 then walks the active list with the gradients in place -- liblinear's order, shrinking and sums.
    auto grad_coop = [&](auto wm_tag) {
        constexpr int WM = decltype(wm_tag)::value, WQ = WM / SV_COOP_WAVES;
        const int t0 = wave * WQ, tid = threadIdx.x;
        double Dq[WQ], Tq[WQ];
        const uint64_t tmask = tmask_w;
#pragma unroll
        for (int q = 0; q < WQ; q++) {
            const int t = t0 + q, i = t * 64 + lane;
            const bool in = t < W;
            Dq[q] = in ? D[i] : 0.0;
            Tq[q] = in ? tau[i] : 0.0;
        }
        const int CH = SL / 4;   // (64 .. 256 columns: two buffers of 8 CH doubles = the 4 SL doubles of the slot arrays)
        for (int c0 = 0, cpar = 0; c0 < P1; c0 += CH, cpar ^= 1) {
            double *buf = Qm + (size_t)cpar * 8 * CH;   // [CH columns][4 waves]{sum of D, sum of tau}
            const int cn = P1 - c0 < CH ? P1 - c0 : CH;
            for (int u0 = 0; u0 < cn; u0 += 8) {
                uint64_t x[8];
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    const int jj = c0 + u0 + c < P1 ? c0 + u0 + c : P1 - 1;
                    x[c] = (colT[(size_t)jj * 64 + lane] & tmask) >> t0;
                }
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    uint64_t M[WQ];
#pragma unroll
                    for (int q = 0; q < WQ; q++) M[q] = __ballot((x[c] >> q) & 1ull);
                    double hd = 0.0, tm = 0.0;
                    if (WQ == 16) {
                        masked_sum8(hd, M, Dq); masked_sum8(hd, M + (WQ == 16 ? 8 : 0), Dq + (WQ == 16 ? 8 : 0));
                        masked_sum8(tm, M, Tq); masked_sum8(tm, M + (WQ == 16 ? 8 : 0), Tq + (WQ == 16 ? 8 : 0));
                    } else if (WQ == 8) { masked_sum8(hd, M, Dq); masked_sum8(tm, M, Tq); }
                    else if (WQ == 4) { masked_sum4(hd, M, Dq); masked_sum4(tm, M, Tq); }
                    else { masked_sum2(hd, M, Dq); masked_sum2(tm, M, Tq); }
                    hd = psk_wave_sum_f64_dpp(hd);
                    tm = psk_wave_sum_f64_dpp(tm);
                    if (lane == 0 && u0 + c < cn) {
                        buf[((size_t)(u0 + c) * 4 + wave) * 2] = hd;
                        buf[((size_t)(u0 + c) * 4 + wave) * 2 + 1] = tm;
                    }
                }
            }
            __syncthreads();
            if (tid < cn) {
                const double *b = buf + (size_t)tid * 8;
                const int j = c0 + tid;
                Hd[j] = ((b[0] + b[2]) + (b[4] + b[6])) + nu;
                Gr[j] = -((b[1] + b[3]) + (b[5] + b[7])) + xjneg[j];
            }
        }
        __syncthreads();
    };
    if (WMREG > 0 && wave != 0) {   // helper_loop: waves 1..3 join every descent / column build of wave 0 and leave with it
        for (;;) {
            __syncthreads();
            const int cmd = s_cd.cmd;
            if (cmd == 2) break;
            if (FORM == 0 && cmd == 3) { if constexpr (FORM == 0) build_coop(std::integral_constant<int, (WMREG > 0 ? WMREG : 32)>{}); }
            else if (FORM == 1 && cmd == 4) { if constexpr (FORM == 1) gg_run(std::false_type{}); }
            else if (FORM == 1 && cmd == 5) { if constexpr (FORM == 1) grad_coop(std::integral_constant<int, (WMREG > 0 ? WMREG : 32)>{}); }
            else cd_coop(std::integral_constant<int, (WMREG > 0 ? WMREG : 32)>{});
        }
        return;
    }

    // training mask and (training & y = -1) mask of this fit: lane t keeps word t (W <= 64, i.e. n <= 4096)
    uint64_t trainw = 0, negw = 0;
    double npos = 0, nneg = 0;
    for (int t = 0; t < W; t++) {
        const int i = t * 64 + lane;
        const bool tr = (i < n) && fold[i] != tf;
        const bool ng = tr && ypm[i] < 0;
        const uint64_t bt = __ballot(tr), bn = __ballot(ng);
        if (lane == t) { trainw = bt; negw = bn; }
        npos += (tr && !ng) ? 1.0 : 0.0;
        nneg += ng ? 1.0 : 0.0;
        ewx[i] = 1.0; tau[i] = C * 0.5; D[i] = C * 0.25; xTd[i] = 0.0; ewxn[i] = 1.0;
    }
    npos = psk_wave_sum_f64_dpp(npos);
    nneg = psk_wave_sum_f64_dpp(nneg);
    const double l = npos + nneg;
    double mn = npos < nneg ? npos : nneg;
    if (mn < 1.0) mn = 1.0;
    const double eps = tol * mn / (l > 0 ? l : 1.0);

    // column j restricted to the training rows: word `lane` in lane `lane`
    auto load_col = [&](int j) -> uint64_t { return (lane < W) ? (cb[(size_t)j * W + lane] & trainw) : 0ull; };

    for (int j = 0; j < P1; j++) {
        double sneg = (lane < W) ? C * (double)__popcll(cb[(size_t)j * W + lane] & negw) : 0.0;
        sneg = psk_wave_sum_f64_dpp(sneg);
        if (lane == 0) { w[j] = 0.0; wpd[j] = 0.0; xjneg[j] = sneg; act[j] = j; }
    }
    double w_norm = 0.0, Gmax_old = 1e300, Gnorm1_init = -1.0, inner_eps = 1.0;
    int newton = 0, floor_steps = 0;   // floor_steps: Newton steps in a row taken inside the line search's rounding noise
    for (newton = 0; newton < max_newton; newton++) {
        double Gmax_new = 0.0, Gnorm1_new = 0.0;
        int active = P1;
        const bool grad4 = FORM == 1 && WMREG > 0 && gg_sl > 0 && f_lds;   // the four-wave pass (grad_coop)
        if constexpr (FORM == 1) {
            if (grad4) {
                if (lane == 0) s_cd.cmd = 5;
                __syncthreads();   // releases waves 1..3 (helper_loop)
                grad_coop(std::integral_constant<int, (WMREG > 0 ? WMREG : 32)>{});
            }
        }
        for (int sidx = 0; sidx < active; sidx++) {
            const int j = f_lds ? act[sidx] : __builtin_amdgcn_readfirstlane(lane == 0 ? act[sidx] : 0);
            double hd = 0.0, tmp = 0.0;
            if (grad4) {
                const double grad = Gr[j], wj = w[j];
                const double Gp = grad + 1.0, Gn = grad - 1.0;
                double viol = 0.0;
                if (wj == 0.0) {
                    if (Gp < 0) viol = -Gp;
                    else if (Gn > 0) viol = Gn;
                    else if (Gp > Gmax_old / l && Gn < -Gmax_old / l) {
                        active--;
                        if (lane == 0) { const int32_t tt = act[sidx]; act[sidx] = act[active]; act[active] = tt; }
                        sidx--;
                        continue;
                    }
                } else if (wj > 0) viol = fabs(Gp);
                else viol = fabs(Gn);
                if (viol > Gmax_new) Gmax_new = viol;
                Gnorm1_new += viol;
                continue;
            }
            {
                const uint64_t cw = load_col(j);
                for (int t = 0; t < W; t++) {
                    const uint64_t xw = psk_readlane_u64(cw, t);
                    const double b = ((xw >> lane) & 1) ? 1.0 : 0.0;
                    const int i = t * 64 + lane;
                    hd += b * D[i];
                    tmp += b * tau[i];
                }
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
                else if (Gp > Gmax_old / l && Gn < -Gmax_old / l) {
                    active--;
                    if (lane == 0) { const int32_t tt = act[sidx]; act[sidx] = act[active]; act[active] = tt; }
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

        for (int t = 0; t < W; t++) xTd[t * 64 + lane] = 0.0;
        double QP_Gmax_old = 1e300;
        int QP_active = active, iter = 0;
        // Covariance form of the same QP when the active set fits the Gram block: slot u = lane + 64 q owns
        // feature act[u]; g = Gr + Q d with Q = X_A' D X_A (+ nu on the diagonal) lives in NS registers per
        // lane, so a coordinate step is a few readlanes, the scalar update and one FMA per slot instead of a
        // reduction over the samples.  Q is stored packed (lower triangle, qcap (qcap + 1) / 2 doubles of
        // LDS) and a column is built the first time its feature moves.  Same visiting order, shrinking and
        // stopping rule as the array form below.
        auto gram_qp = [&](auto ns_tag, auto packed_tag) {
            constexpr int NS = decltype(ns_tag)::value;
            constexpr bool PACKED = decltype(packed_tag)::value;  // square (64 NS)^2 layout when it fits, else packed
            int fjs[NS], perm[NS], kq[NS], tri[NS];
            double g[NS], h[NS], hinv[NS], wr[NS], wpr[NS];
            uint64_t have[NS];
#pragma unroll
            for (int q = 0; q < NS; q++) {
                const int u = lane + 64 * q;
                const bool mine = u < active;
                fjs[q] = mine ? act[u] : 0;
                g[q] = mine ? FLD(&Gr[fjs[q]]) : 0.0;
                h[q] = mine ? FLD(&Hd[fjs[q]]) : 1.0;
                hinv[q] = 1.0 / h[q];  // one division per feature and Newton step instead of one per coordinate step
                wr[q] = mine ? FLD(&w[fjs[q]]) : 0.0;
                wpr[q] = wr[q];
                perm[q] = u;
                have[q] = 0;
                kq[q] = mine ? u : 0;  // idle slots read (and ignore) a valid entry
                tri[q] = kq[q] * (kq[q] + 1) / 2;
                if (WMREG > 0 && mine) s_fj[u] = fjs[q];   // for build_coop (read behind the barrier that releases it)
            }
            // a[u >> 6] of lane u & 63 for a wave-uniform slot id u: uniform branches, one lane read each
            auto pick_i = [&](const int *a, int u) {
                const int ln = u & 63;
                if (NS == 1 || u < 64) return __builtin_amdgcn_readlane(a[0], ln);
                if (NS == 2 || u < 128) return __builtin_amdgcn_readlane(a[NS > 1 ? 1 : 0], ln);
                return __builtin_amdgcn_readlane(a[NS > 2 ? 2 : 0], ln);
            };
            auto pick_d = [&](const double *a, int u) {
                const int ln = u & 63;
                if (NS == 1 || u < 64) return psk_readlane_f64(a[0], ln);
                if (NS == 2 || u < 128) return psk_readlane_f64(a[NS > 1 ? 1 : 0], ln);
                return psk_readlane_f64(a[NS > 2 ? 2 : 0], ln);
            };
            auto qidx = [&](int k, int m) {
                if (!PACKED) return m * (64 * NS) + k;
                const int hi = k > m ? k : m, lo = k > m ? m : k;
                return hi * (hi + 1) / 2 + lo;
            };
            // builds column m of Q if it is not there yet; returns true when it had to
            auto ensure_col = [&](int m) {
                bool built = false;
#pragma unroll
                for (int q = 0; q < NS; q++) if ((m >> 6) == q) { built = (have[q] >> (m & 63)) & 1; have[q] |= 1ull << (m & 63); }
                if (built) return false;
                const int fm = pick_i(fjs, m);
                if (WMREG > 0) {
                    if (lane == 0) { s_cd.cmd = 3; s_cd.fm = fm; s_cd.active = active; }
                    __syncthreads();   // releases waves 1..3 into build_coop
                    build_coop(std::integral_constant<int, (WMREG > 0 ? WMREG : 32)>{});
#pragma unroll
                    for (int q = 0; q < NS; q++) {
                        const int k = lane + 64 * q;
                        if (k >= active) continue;
                        double acc = s_qpart[0][k];
#pragma unroll
                        for (int v = 1; v < SV_COOP_WAVES; v++) acc += s_qpart[v][k];
                        if (k == m) acc = h[q];
                        Qm[qidx(k, m)] = acc;
                    }
                    return true;
                }
#pragma unroll
                for (int q = 0; q < NS; q++) {
                    const int k = lane + 64 * q;
                    if (k >= active) continue;
                    double acc = 0.0;
                    for (int t = 0; t < W; t++) {
                        const uint64_t tw = psk_readlane_u64(trainw, t);
                        uint64_t x = cb[(size_t)fm * W + t] & cb[(size_t)fjs[q] * W + t] & tw;
                        while (x) {
                            acc += D[t * 64 + __builtin_ctzll(x)];
                            x &= x - 1;
                        }
                    }
                    if (k == m) acc = h[q];
                    Qm[qidx(k, m)] = acc;
                }
                return true;
            };
            // Accelerator for ill-conditioned models (near-duplicate columns make cyclic CD crawl: 1000 sweeps
            // per Newton step were the norm): with the signs of the non-zero coordinates held fixed, the model
            // restricted to them is a plain quadratic.  A few conjugate-gradient steps on Q_FF delta = -(g_F + sign_F)
            // give a descent direction; the step stops where the first coordinate would change sign (it lands on
            // exactly 0).  Any point is a valid CD iterate, so the sweeps -- and liblinear's stopping rule -- go on
            // unchanged from there.  Each polish can drop one coordinate to zero, so polishes repeat while the step
            // is cut short; short CG runs with many repeats beat long ones (r01 sweep on two 256-sample problems:
            // 16 CG steps x up to 64 repeats after sweeps 4, 8, 16, ... gave 0.18 -> 0.012 s and 1.9 -> 0.12 s).
            auto polish = [&]() -> bool {  // true when the step was cut short by a sign change
                bool inF[NS];
                double sg[NS], dl[NS], r[NS], pd[NS], Qp[NS], Qd[NS];
                double rs = 0.0;
#pragma unroll
                for (int q = 0; q < NS; q++) {
                    inF[q] = (lane + 64 * q < active) && wpr[q] != 0.0;
                    sg[q] = wpr[q] > 0.0 ? 1.0 : -1.0;
                    r[q] = inF[q] ? -(g[q] + sg[q]) : 0.0;
                    pd[q] = r[q];
                    dl[q] = 0.0;
                    Qd[q] = 0.0;
                    rs += r[q] * r[q];
                }
                rs = psk_wave_sum_f64_dpp(rs);
                const double b2 = rs;
                if (!(b2 > 0.0)) return false;
                for (int u = 0; u < active; u++)
                    if (pick_d(pd, u) != 0.0) (void)ensure_col(u);
                for (int it = 0; it < cg_max; it++) {
#pragma unroll
                    for (int q = 0; q < NS; q++) Qp[q] = 0.0;
                    for (int u = 0; u < active; u++) {
                        const double pj = pick_d(pd, u);
                        if (pj == 0.0) continue;
                        const int tri_u = u * (u + 1) / 2, sq_u = u * (64 * NS);
#pragma unroll
                        for (int q = 0; q < NS; q++)
                            Qp[q] += pj * Qm[PACKED ? (kq[q] > u ? tri[q] + u : tri_u + kq[q]) : sq_u + kq[q]];
                    }
                    double pq = 0.0;
#pragma unroll
                    for (int q = 0; q < NS; q++) pq += inF[q] ? pd[q] * Qp[q] : 0.0;
                    pq = psk_wave_sum_f64_dpp(pq);
                    if (!(pq > 0.0)) break;
                    const double a = rs / pq;
                    double rn = 0.0;
#pragma unroll
                    for (int q = 0; q < NS; q++) {
                        dl[q] += a * pd[q];
                        Qd[q] += a * Qp[q];
                        r[q] = inF[q] ? r[q] - a * Qp[q] : 0.0;
                        rn += r[q] * r[q];
                    }
                    rn = psk_wave_sum_f64_dpp(rn);
                    if (!(rn > 1e-18 * b2)) break;
                    const double beta = rn / rs;
#pragma unroll
                    for (int q = 0; q < NS; q++) pd[q] = r[q] + beta * pd[q];
                    rs = rn;
                }
                // a direction that is not finite (breakdown of CG on a numerically singular block) is dropped
                bool bad = false;
#pragma unroll
                for (int q = 0; q < NS; q++) bad |= !(fabs(dl[q]) < 1e300) || !(fabs(Qd[q]) < 1e300);
                if (__ballot(bad)) return false;
                // longest step in (0, 1] that keeps every sign
                double tmax = 1.0;
#pragma unroll
                for (int q = 0; q < NS; q++)
                    if (inF[q] && (wpr[q] + dl[q]) * sg[q] <= 0.0) tmax = fmin(tmax, -wpr[q] / dl[q]);
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) tmax = fmin(tmax, psk_shfl_xor_f64(tmax, d));
                if (!(tmax > 0.0)) return false;
#pragma unroll
                for (int q = 0; q < NS; q++) {
                    if (inF[q]) {
                        const bool hits = (wpr[q] + dl[q]) * sg[q] <= 0.0 && -wpr[q] / dl[q] <= tmax;
                        wpr[q] = hits ? 0.0 : wpr[q] + tmax * dl[q];
                    }
                    g[q] += tmax * Qd[q];
                }
                return tmax < 1.0;
            };
            while (iter < 1000) {
                if (f_lds) {  // fresh random visiting order (see the array form); `act` in LDS serves as the scratch
#pragma unroll
                    for (int q = 0; q < NS; q++) if (lane + 64 * q < QP_active) act[lane + 64 * q] = perm[q];
                    if (lane == 0) {
                        for (int jj = 0; jj + 1 < QP_active; jj++) {
                            rng ^= rng << 13; rng ^= rng >> 17; rng ^= rng << 5;
                            const int ii = jj + (int)(rng % (uint32_t)(QP_active - jj));
                            const int32_t tt = act[ii]; act[ii] = act[jj]; act[jj] = tt;
                        }
                    }
#pragma unroll
                    for (int q = 0; q < NS; q++) if (lane + 64 * q < QP_active) perm[q] = act[lane + 64 * q];
                }
                double QP_Gmax_new = 0.0, QP_Gnorm1_new = 0.0;
                for (int sidx = 0; sidx < QP_active; sidx++) {
                    const int m = pick_i(perm, sidx);
                    double qcol[NS];
                    const int tri_m = m * (m + 1) / 2, sq_m = m * (64 * NS);
#pragma unroll
                    for (int q = 0; q < NS; q++)
                        qcol[q] = Qm[PACKED ? (kq[q] > m ? tri[q] + m : tri_m + kq[q]) : sq_m + kq[q]];
                    const double G = pick_d(g, m), H = pick_d(h, m), Hi = pick_d(hinv, m), wp = pick_d(wpr, m);
                    const double Gp = G + 1.0, Gn = G - 1.0;
                    double viol = 0.0;
                    if (wp == 0.0) {
                        if (Gp < 0) viol = -Gp;
                        else if (Gn > 0) viol = Gn;
                        else if (Gp > QP_Gmax_old / l && Gn < -QP_Gmax_old / l) {
                            QP_active--;
                            const int last = pick_i(perm, QP_active);
#pragma unroll
                            for (int q = 0; q < NS; q++) {
                                if (lane + 64 * q == sidx) perm[q] = last;
                                if (lane + 64 * q == QP_active) perm[q] = m;
                            }
                            sidx--;
                            continue;
                        }
                    } else if (wp > 0) viol = fabs(Gp);
                    else viol = fabs(Gn);
                    if (viol > QP_Gmax_new) QP_Gmax_new = viol;
                    QP_Gnorm1_new += viol;
                    double z;
                    if (Gp < H * wp) z = -Gp * Hi;
                    else if (Gn > H * wp) z = -Gn * Hi;
                    else z = -wp;
                    if (fabs(z) < 1e-12 && !(z == -wp && wp != 0.0)) continue;  // see the array form below
                    z = fmin(fmax(z, -10.0), 10.0);
#ifdef PSK_SV_STATS
                    const long long stat_b0 = clock64();
#endif
                    if (ensure_col(m)) {
#pragma unroll
                        for (int q = 0; q < NS; q++) qcol[q] = Qm[qidx(kq[q], m)];
#ifdef PSK_SV_STATS
                        stat_t_build += clock64() - stat_b0;
                        stat_builds++;
#endif
                    }
#pragma unroll
                    for (int q = 0; q < NS; q++) {
                        if ((m >> 6) == q && lane == (m & 63)) wpr[q] += z;  // the slot test is wave-uniform
                        g[q] += z * qcol[q];
                    }
                }
                iter++;
                if (QP_Gnorm1_new <= inner_eps * Gnorm1_init) {
                    if (QP_active == active) break;
                    QP_active = active;
                    QP_Gmax_old = 1e300;
                    continue;
                }
                QP_Gmax_old = QP_Gmax_new;
                if (iter >= 4 && (iter & (iter - 1)) == 0) {  // after sweeps 4, 8, 16, ...
#ifdef PSK_SV_STATS
                    const long long stat_p0 = clock64();
#endif
                    for (int rep = 0; rep < polish_reps && polish(); rep++) {}
#ifdef PSK_SV_STATS
                    stat_t_polish += clock64() - stat_p0;
#endif
                }
            }
            // back to the array form: new visiting order, wpd, and xTd = X_A d
            int act_new[NS];
#pragma unroll
            for (int q = 0; q < NS; q++) {
                const int uu = perm[q];  // slot at position lane + 64 q
                int f = 0;
#pragma unroll
                for (int q2 = 0; q2 < NS; q2++) {
                    const int cand = __shfl(fjs[q2], uu & 63);
                    if ((uu >> 6) == q2) f = cand;
                }
                act_new[q] = f;
            }
            double dr[NS];
#pragma unroll
            for (int q = 0; q < NS; q++) {
                dr[q] = wpr[q] - wr[q];
                if (lane + 64 * q < active) { wpd[fjs[q]] = wpr[q]; act[lane + 64 * q] = act_new[q]; }
            }
            for (int u = 0; u < active; u++) {
                const double d = pick_d(dr, u);
                if (d == 0.0) continue;
                const uint64_t cw = load_col(pick_i(fjs, u));
                for (int t = 0; t < W; t++) {
                    const uint64_t xw = psk_readlane_u64(cw, t);
                    if ((xw >> lane) & 1) xTd[t * 64 + lane] += d;
                }
            }
        };
        using std::integral_constant;
#ifdef PSK_SV_STATS
        stat_t0 = clock64();
        const long long stat_g0 = stat_t0;
        const long long stat_cd_before = stat_t_cd;
#endif
        // which inner solver: 0 = the array descent, 1 = gg_run, 2 .. 6 = the LDS Gram block (slots per lane, square / packed)
        int qp_form = 0;
        if (FORM == 1) qp_form = (WMREG > 0 && SL > 0 && active >= 64) ? 1 : 0;
        else if (SL == 0 && q_lds && active <= 64 && q_doubles >= 64 * 64) qp_form = 2;
        else if (SL == 0 && q_lds && active <= 128 && q_doubles >= 128 * 128) qp_form = 3;
        else if (SL == 0 && q_lds && active <= 192 && (size_t)active * (active + 1) / 2 <= q_doubles) qp_form = active <= 64 ? 4 : active <= 128 ? 5 : 6;
        if (FORM == 1 && qp_form == 1) {
            if constexpr (FORM == 1) {
            // the Gram matrix in global memory (gg_run above); its LDS area takes the place of the Gram block
            if (lane == 0) {
                s_cd.inner_eps = inner_eps; s_cd.Gnorm1_init = Gnorm1_init; s_cd.l = l;
                s_cd.active = active; s_cd.cmd = 4;
            }
            __syncthreads();   // releases waves 1..3 (helper_loop)
            iter = gg_run(std::true_type{});   // (leaves x.d = X_A d in place)
#ifdef PSK_SV_STATS
            stat_visits += (long long)iter * active;
#endif
            }
        } else if (FORM == 0 && qp_form >= 2) {
            if constexpr (FORM == 0) {
            if (qp_form == 2) gram_qp(integral_constant<int, 1>{}, integral_constant<bool, false>{});
            else if (qp_form == 3) gram_qp(integral_constant<int, 2>{}, integral_constant<bool, false>{});
            else if (qp_form == 4) gram_qp(integral_constant<int, 1>{}, integral_constant<bool, true>{});
            else if (qp_form == 5) gram_qp(integral_constant<int, 2>{}, integral_constant<bool, true>{});
            else gram_qp(integral_constant<int, 3>{}, integral_constant<bool, true>{});
            }
        } else {
        // Array form with the samples in REGISTERS: lane l owns samples l, l + 64, ..., so D and
        // x.d of a fit are W doubles per lane each, and the column of a coordinate arrives TRANSPOSED (colT: bit t of
        // lane l's word = sample 64 t + l) -- one coalesced load, requested a step ahead, no lane reads.  A coordinate
        // step is then W predicated FMAs, one wave sum and W predicated adds, without an LDS access; with the arrays in
        // LDS every word step waited for two LDS reads (~100 clocks: the loop is not unrolled, W is a run-time value)
        // and a step took ~1 us (r02: 2048 samples x 907 columns, 16 M steps = the 16 s of that grid).  Same visiting
        // order as the LDS form below (PSK_NO_CD_REGS); four waves share the words of a fit: cd_coop above.
#ifdef PSK_SV_STATS
        stat_t0 = clock64();
#endif
        if (WMREG > 0) {
            if (lane == 0) {
                s_cd.QP_Gmax_old = QP_Gmax_old; s_cd.inner_eps = inner_eps; s_cd.Gnorm1_init = Gnorm1_init; s_cd.l = l;
                s_cd.QP_active = QP_active; s_cd.active = active; s_cd.cmd = 1;
            }
            __syncthreads();   // releases waves 1..3 (helper_loop); x.d = 0 and this Newton step's D are visible to them
            iter = cd_coop(std::integral_constant<int, (WMREG > 0 ? WMREG : 32)>{});
#ifdef PSK_SV_STATS
            stat_t_cd += clock64() - stat_t0;
            stat_sweeps += iter;
            stat_visits += (long long)iter * active;
#endif
        } else
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
                const int j = f_lds ? act[sidx] : __builtin_amdgcn_readfirstlane(lane == 0 ? act[sidx] : 0);
                const double H = FLD(&Hd[j]);
                const double wp = FLD(&wpd[j]);
                const uint64_t cw = load_col(j);
                double G = 0.0;
                for (int t = 0; t < W; t++) {
                    const uint64_t xw = psk_readlane_u64(cw, t);
                    const double b = ((xw >> lane) & 1) ? 1.0 : 0.0;
                    const int i = t * 64 + lane;
                    G += b * D[i] * xTd[i];
                }
                G = psk_wave_sum_f64_dpp(G) + FLD(&Gr[j]) + (wp - FLD(&w[j])) * nu;
                const double Gp = G + 1.0, Gn = G - 1.0;
                double viol = 0.0;
                if (wp == 0.0) {
                    if (Gp < 0) viol = -Gp;
                    else if (Gn > 0) viol = Gn;
                    else if (Gp > QP_Gmax_old / l && Gn < -QP_Gmax_old / l) {
                        QP_active--;
                        if (lane == 0) { const int32_t tt = act[sidx]; act[sidx] = act[QP_active]; act[QP_active] = tt; }
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
                for (int t = 0; t < W; t++) {
                    const uint64_t xw = psk_readlane_u64(cw, t);
                    if ((xw >> lane) & 1) xTd[t * 64 + lane] += z;
                }
            }
            iter++;
            if (QP_Gnorm1_new <= inner_eps * Gnorm1_init) {
                if (QP_active == active) break;
                QP_active = active;
                QP_Gmax_old = 1e300;
                continue;
            }
            QP_Gmax_old = QP_Gmax_new;
        }
        }

        double delta = 0.0, w_norm_new = 0.0;
        for (int j = 0; j < P1; j++) {
            const double wp = FLD(&wpd[j]);
            delta += FLD(&Gr[j]) * (wp - FLD(&w[j]));
            w_norm_new += fabs(wp);
        }
        delta += (w_norm_new - w_norm);
        double negsum = 0.0;
        for (int t = 0; t < W; t++) {
            const uint64_t nw = psk_readlane_u64(negw, t);
            if ((nw >> lane) & 1) negsum += C * xTd[t * 64 + lane];
        }
        negsum = psk_wave_sum_f64_dpp(negsum);
        bool accepted = false, floor_rebuild = false;
        for (int ls = 0; ls < 20; ls++) {
            double cs = 0.0;
            for (int t = 0; t < W; t++) {
                const uint64_t tw = psk_readlane_u64(trainw, t);
                if (!((tw >> lane) & 1)) continue;
                const int i = t * 64 + lane;
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
                for (int t = 0; t < W; t++) {
                    const uint64_t tw = psk_readlane_u64(trainw, t);
                    if (!((tw >> lane) & 1)) continue;
                    const int i = t * 64 + lane;
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
            for (int t = 0; t < W; t++) xTd[t * 64 + lane] *= 0.5;
        }
        if (!accepted || floor_rebuild) {
            // rejected 20 times: back to the current w, and exp(w.x) is rebuilt from w (see the float kernel)
            for (int j = 0; j < P1; j++) { if (lane == 0) wpd[j] = w[j]; }
            for (int t = 0; t < W; t++) xTd[t * 64 + lane] = 0.0;
            for (int j = 0; j < P1; j++) {
                const double wj = FLD(&w[j]);
                if (wj == 0.0) continue;
                const uint64_t cw = load_col(j);
                for (int t = 0; t < W; t++) {
                    const uint64_t xw = psk_readlane_u64(cw, t);
                    if ((xw >> lane) & 1) xTd[t * 64 + lane] += wj;
                }
            }
            for (int t = 0; t < W; t++) {
                const uint64_t tw = psk_readlane_u64(trainw, t);
                if (!((tw >> lane) & 1)) continue;
                const int i = t * 64 + lane;
                const double en = exp(xTd[i]);
                const double tt = 1.0 / (1.0 + en);
                ewx[i] = en; tau[i] = C * tt; D[i] = C * en * tt * tt;
            }
        }
#ifdef PSK_SV_STATS
        if (stat_t_cd == stat_cd_before) { stat_t_gram += clock64() - stat_g0; stat_gram_sweeps += iter; }
#endif
        if (iter == 1) inner_eps *= 0.25;
        if (floor_steps >= 3) { newton++; break; }   // at the floor of what doubles resolve (see the line search)
        Gmax_old = Gmax_new;
    }
    for (int j = 0; j < p; j++) { if (lane == 0) coef[(size_t)fit * p + j] = w[j]; }
    if (lane == 0) { icpt[fit] = w[p]; iters[fit] = newton; }
    if (WMREG > 0) {   // the other waves leave helper_loop
        if (lane == 0) s_cd.cmd = 2;
        __syncthreads();
    }
#ifdef PSK_SV_STATS
    if (lane == 0 && SL > 0)
        printf("fit %d waits at the descent's barrier: wave0 %lld wave1 %lld wave2 %lld wave3 %lld; visits to shrunk slots %lld\n", fit, s_stat_wait[0], s_stat_wait[1], s_stat_wait[2], s_stat_wait[3], s_stat_dead);
    if (lane == 0)
        printf("fit %d C %g newton %d sweeps %lld visits<= %lld cd_cycles %lld total_cycles %lld gram_cycles %lld gram_sweeps %lld builds %lld build_cycles %lld polish_cycles %lld\n",
               fit, C, newton, stat_sweeps, stat_visits, stat_t_cd, (long long)(clock64() - stat_start), stat_t_gram, stat_gram_sweeps, stat_builds,
               stat_t_build, stat_t_polish);
#endif
}
#undef FLD

template <int FORM>
hipError_t l1_bits_launch(const psk_l1_bits_launch &a)
{
    auto pick = [&](auto all) {
        constexpr bool A = decltype(all)::value;
        if constexpr (FORM == 0)
            return a.wmreg == 0 ? logreg_newglmnet_bits_kernel<A, 0, FORM> : a.wmreg == 16 ? logreg_newglmnet_bits_kernel<A, 16, FORM>
                 : a.wmreg == 32 ? logreg_newglmnet_bits_kernel<A, 32, FORM> : logreg_newglmnet_bits_kernel<A, 64, FORM>;
        else
            return a.wmreg == 16 ? logreg_newglmnet_bits_kernel<A, 16, FORM>
                 : a.wmreg == 32 ? logreg_newglmnet_bits_kernel<A, 32, FORM> : logreg_newglmnet_bits_kernel<A, 64, FORM>;
    };
    auto kern = a.all_lds ? pick(std::true_type{}) : pick(std::false_type{});
    if (a.lds_bytes > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)a.lds_bytes);
        if (e != hipSuccess) return e;
    }
    kern<<<a.n_fits, a.wmreg > 0 ? SV_COOP_THREADS : SV_THREADS, a.lds_bytes, a.stream>>>(
        a.bits, a.ypm, a.fold, a.n, a.p, a.W, a.fit_param, a.fit_fold, a.tol, a.max_iter, a.coef, a.icpt, a.iters, a.work, a.iwork,
        a.f_lds, a.s_lds, a.c_lds, a.q_doubles, a.cg_max, a.polish_reps, a.bitsT, a.gg_sl, a.gg_q, a.gg_stride, a.gg_polish_from);
    return hipGetLastError();
}

}  // namespace
