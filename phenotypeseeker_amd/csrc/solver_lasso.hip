// a10, continuous phenotypes: the Lasso behind GridSearchCV (modeling.py:999-1000 `Lasso(max_iter, tol)`, :1041 the alpha
// grid, :1078-1080 / :1208-1216 the search) on a 0/1 design, in COVARIANCE form.
//
// scikit-learn's solver (enet_coordinate_descent, the one `Lasso` calls with precompute=False) keeps the residual
// R = y - Xw and visits the features in cyclic order:  tmp = X_j'R + w_j |X_j|^2,  w_j <- sign(tmp) max(|tmp| - alpha n, 0)
// / |X_j|^2,  R updated at once; after a sweep whose largest step is below tol x the largest coefficient (or the last
// sweep allowed) it evaluates the duality gap and stops when gap < tol y'y.  Everything it needs of the samples is
// g = X'R (and y'y): with Q = X'X (X, y centred over the training rows),  g_k -= dd_j Q[k][j]  after a step dd_j -- the same
// iterates, no pass over the samples.  For a 0/1 design Q[k][j] = c_kj - c_k c_j / n with c_kj the number of training
// samples that carry both k-mers: exact integers, counted once per held-out fold with popcounts (every fit of a fold
// shares them) and kept as u16 (n <= 4,096), 1.6 MB per fold at 907 columns.
//
// One workgroup of four waves per fit.  The features are cut into blocks of 64; row k = 64 c + lane of g, w, c_k lives
// in ONE lane of wave c & 3 (register c >> 2).  Block c is stepped by its owner wave: 64 dependent coordinate steps
// on the block's own 64 x 64 piece of Q (lane = row, a step = one lane read of the step, one multiply-add per lane), the
// 64 steps dd_c go to LDS, and EVERY wave applies them to its other rows -- 64 multiply-adds per row -- while the next
// block's owner has applied them to that block's rows first and is already stepping: the order of the coordinates and
// what each step sees are exactly the cyclic descent's, one barrier per 64 visits instead of one per visit.
// r03's lasso_bits_kernel (four waves, masked sums over the samples, one barrier per visit) took ~1,300 cycles per
// visit; this form ~110.  The stop is scikit-learn's: the duality gap from g, w, X'y and y'y (R'R = R'y - w'g,
// R'y = y'y - w'X'y), evaluated under its conditions, max_iter sweeps at most, the sweeps counted as it counts them.
// Columns with zero centred norm are skipped as it skips them.  One deliberate difference: the step divides by
// multiplying with 1 / |X_j|^2 (at most one ulp per step from its division).
#include "solver_common.h"

namespace {

constexpr int LC_THREADS = 256;

// co-occurrence counts over the training samples of fold f: C[f][k][j] = #{i in train_f : x_ik = x_ij = 1} (k = j: the
// column's count).  A 64 x 64 tile per workgroup; thread (row i, wave jq) sums 16 columns over the words in LDS.
__global__ __launch_bounds__(LC_THREADS) void lasso_cooc_kernel(const uint64_t *__restrict__ bits, const uint64_t *__restrict__ tmask,
                                                                uint16_t *__restrict__ C, int W, int PP)
{
    __shared__ uint64_t A[16][64];
    __shared__ uint64_t B[64][16];
    const int bi = blockIdx.x, bj = blockIdx.y, f = blockIdx.z, t = threadIdx.x, i = t & 63, jq = t >> 6;
    uint32_t acc[16];
#pragma unroll
    for (int jj = 0; jj < 16; jj++) acc[jj] = 0;
    for (int w0 = 0; w0 < W; w0 += 16) {
        for (int q = t; q < 1024; q += LC_THREADS) {
            const int row = q >> 4, w = q & 15;
            const bool in = w0 + w < W;
            const uint64_t m = in ? tmask[(size_t)f * W + w0 + w] : 0ull;
            A[w][row] = in ? (bits[(size_t)(bi * 64 + row) * W + w0 + w] & m) : 0ull;
            B[row][w] = in ? bits[(size_t)(bj * 64 + row) * W + w0 + w] : 0ull;
        }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 16; w++) {
            const uint64_t a = A[w][i];
#pragma unroll
            for (int jj = 0; jj < 16; jj++) acc[jj] += (uint32_t)__popcll(a & B[jq * 16 + jj][w]);
        }
        __syncthreads();
    }
    // TILED: the 64 x 64 counts of (row block bi, column block bj) are 8 KB, stored [8 columns e][row i][8 counts]: the 16 bytes
    // a lane of the descent reads per load are next to its neighbours' (row-major, the 64 lanes of a load touched 64 lines)
    const int nb = PP >> 6;
    uint16_t *tile = C + (((size_t)f * nb + bi) * nb + bj) * 4096;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        uint16_t *out = tile + ((size_t)(2 * jq + h) * 64 + i) * 8;
#pragma unroll
        for (int jj = 0; jj < 8; jj++) out[jj] = (uint16_t)acc[8 * h + jj];
    }
}

// count of rows i, j of block pair (bi, bj) in the tiled layout
__device__ __forceinline__ size_t lc_tile_at(int nb, int bi, int bj, int i, int j)
{
    return ((size_t)bi * nb + bj) * 4096 + ((size_t)(j >> 3) * 64 + i) * 8 + (j & 7);
}

// X'y of the centred problem of fold f: q0[f][k] = sum over the training samples with the k-mer of (y_i - mean_f y)
// (yc[f][i], 0 for held-out samples: the centring of X drops out because the centred y sums to zero)
__global__ __launch_bounds__(64) void lasso_xty_kernel(const uint64_t *__restrict__ bits, const double *__restrict__ yc,
                                                       double *__restrict__ q0, int W, int PP, int n)
{
    const int k = blockIdx.x, f = blockIdx.y, lane = threadIdx.x;
    double s = 0.0;
    for (int t = 0; t < W; t++) {
        const int i = t * 64 + lane;
        const uint64_t x = bits[(size_t)k * W + t];
        if (i < n && ((x >> lane) & 1ull)) s += yc[(size_t)f * n + i];
    }
    s = psk_wave_sum_f64_dpp(s);
    if (lane == 0) q0[(size_t)f * PP + k] = s;
}

// the diagonal blocks of Q, centred, as doubles: Dg[f][c][j][i] = Q[64 c + i][64 c + j] (what the 64 steps of a block use)
__global__ __launch_bounds__(LC_THREADS) void lasso_diag_kernel(const uint16_t *__restrict__ C, const double *__restrict__ fstat,
                                                                double *__restrict__ Dg, int PP)
{
    const int c = blockIdx.x, f = blockIdx.y, nb = PP >> 6;
    const double ntr = fstat[4 * f + 2];
    const uint16_t *Cf = C + (size_t)f * PP * PP;
    for (int q = threadIdx.x; q < 4096; q += LC_THREADS) {
        const int j = q >> 6, i = q & 63;
        const double ci = (double)Cf[lc_tile_at(nb, c, c, i, i)], cj = (double)Cf[lc_tile_at(nb, c, c, j, j)];
        Dg[(((size_t)f * nb + c) * 64 + j) * 64 + i] = (double)Cf[lc_tile_at(nb, c, c, i, j)] - ci * cj / ntr;
    }
}

struct LcRec {            // what the owner of a block leaves for everybody (slot = period & 3)
    double dd[64];        // the 64 steps
    double s;             // sum_j dd_j c_j / n: the centring term of the update, g_k += c_k s
    uint64_t nz;          // which steps are non-zero (0: nothing to apply)
    int check, pad;       // the sweep ended and scikit-learn would evaluate the gap now
};

// R4: register rows per lane (blocks of 64 features per wave): p <= 256 R4
template <int R4>
__global__ __launch_bounds__(LC_THREADS) void lasso_cov_kernel(
    const uint16_t *__restrict__ C, const double *__restrict__ Dg, const double *__restrict__ q0, const double *__restrict__ fstat,
    const int32_t *__restrict__ block_fit, const double *__restrict__ fit_param, const int32_t *__restrict__ fit_fidx, int p, int PP,
    double tol, int max_iter, double *__restrict__ coef, double *__restrict__ icpt, int32_t *__restrict__ iters, double *__restrict__ gaps)
{
    extern __shared__ double lq_all[];   // [4 waves][64 + 8][64]: the diagonal block of Q a wave steps next, staged a period or two ahead
    __shared__ LcRec rec[4];
    __shared__ double s_red[4][4];
    __shared__ double s_dmax, s_wmax;
    const int fit = block_fit[blockIdx.x];
    if (fit < 0) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nb = PP >> 6;
    const int f = fit_fidx[fit];
    const double ym = fstat[4 * f], yy = fstat[4 * f + 1], ntr = fstat[4 * f + 2];
    const double an = fit_param[fit] * ntr, tol_s = tol * yy, inv_n = 1.0 / ntr;
    const uint16_t *Cf = C + (size_t)f * PP * PP;
    const double *Dgf = Dg + (size_t)f * nb * 4096;
    const double *q0f = q0 + (size_t)f * PP;
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    // this lane's rows (the centred norm of a row is recomputed from its count when its block is stepped)
    double g[R4], wv[R4], cnt[R4];
    int ap[R4];   // the first period whose steps row r has not received yet (wave-uniform)
#pragma unroll
    for (int r = 0; r < R4; r++) {
        const int cb = 4 * r + wave, k = 64 * cb + lane;
        const bool valid = cb < nb && k < p;
        cnt[r] = valid ? (double)Cf[lc_tile_at(nb, cb, cb, lane, lane)] : 0.0;
        g[r] = valid ? q0f[k] : 0.0;               // w = 0: R = y, g = X'y
        wv[r] = 0.0;
        ap[r] = 0;
    }
    if (threadIdx.x == 0) { s_dmax = 0.0; s_wmax = 0.0; }
    __syncthreads();
    // A row's counts against the 64 columns of a block are 128 bytes per lane, wanted as soon as the block's steps are
    // published.  xA holds them for block blkA -- every wave asks for the block being stepped NOW, a period before it can
    // apply that block's steps -- so the multiply-adds never wait for memory (the first version loaded inside the pass:
    // ~1.5 us of L2 latency per row, 9 us per period, three times the arithmetic).
    u4 xA[R4][8];
    int blkA = -1;
    auto chunk = [&](int cb, int b) __attribute__((always_inline)) {   // [e] = counts against columns 8 e .. 8 e + 7 of block b: stride 64
        return reinterpret_cast<const u4 *>(Cf + ((size_t)cb * nb + b) * 4096) + lane;
    };
    // does row block cb receive anything from period P: not its own block's steps (taken in place), not an all-zero record
    auto needs = [&](int cb, int P) __attribute__((always_inline)) { return cb < nb && cb != P % nb && rec[P & 3].nz != 0ull; };
    // the steps of period P on one row: g -= sum_j dd_j c_kj - c_k s
    auto consume = [&](double &gr, double cr, const u4 (&x)[8], int P) __attribute__((always_inline)) {
        typedef double d2 __attribute__((ext_vector_type(2)));
        const LcRec &R = rec[P & 3];
        const d2 *dd2 = reinterpret_cast<const d2 *>(R.dd);
        // eight chains, and a fence after every group of eight multiply-adds: left alone the compiler lines a chain's
        // sixteen multiply-adds up one behind the other (fewer live registers), each waiting for the one before it -- a lone
        // wave issues in order, so a row took ~3,700 cycles instead of ~1,300
        double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        // the steps come from LDS eight at a time, the next eight asked for before the current ones are used (read where they
        // are used, every pair of multiply-adds waited ~100 cycles for its ds_read: 13,000 of a period's 15,000 cycles)
        d2 cur[4], nxt[4];
#pragma unroll
        for (int h = 0; h < 4; h++) cur[h] = dd2[h];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            if (e + 1 < 8) {
#pragma unroll
                for (int h = 0; h < 4; h++) nxt[h] = dd2[4 * (e + 1) + h];
            }
#pragma unroll
            for (int h = 0; h < 4; h++) {
                uint32_t v = x[e][h];
                // (opaque: the conversions of a chunk are NOT hoisted out of the loop over the periods that calls this -- as
                // loop invariants they were 64 doubles, 128 registers, kept for a loop of one or two passes)
                asm("" : "+v"(v) : "s"(P));   // (not volatile: a volatile asm is a scheduling barrier, 64 of them serialise the pass)
                // u16 -> double through the exponent of 2^52 (and / shift, one addition: v_cvt_f64_u32 runs at a quarter of the rate)
                const double c0 = __hiloint2double(0x43300000, (int)(v & 0xFFFFu)) - 4503599627370496.0;
                const double c1 = __hiloint2double(0x43300000, (int)(v >> 16)) - 4503599627370496.0;
                acc[2 * h] = fma(cur[h][0], c0, acc[2 * h]);
                acc[2 * h + 1] = fma(cur[h][1], c1, acc[2 * h + 1]);
            }
#pragma unroll
            for (int h = 0; h < 4; h++) cur[h] = nxt[h];
            // (the fence: an empty asm every chain passes through -- the group's multiply-adds cannot sink below it)
            asm("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]));
        }
        gr = (gr - (((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7])))) + cr * R.s;
    };
    // The same for a period in which few coordinates moved (late sweeps: the support is a tenth of the columns and only it
    // moves): a multiply-add per NON-ZERO step -- the count picked out of the lane's 32 registers by a wave-uniform index,
    // the step by a lane read -- instead of 64: ~80 cycles per non-zero step against ~4,000 for the full row.
    auto consume_sparse = [&](double &gr, double cr, const u4 (&x)[8], int P) __attribute__((always_inline)) {
        typedef uint32_t v32 __attribute__((ext_vector_type(32)));
        const LcRec &R = rec[P & 3];
        const uint64_t nzv = R.nz;
        uint64_t m = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(nzv >> 32)) << 32) |
                     (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)nzv);
        const double ddv = R.dd[lane];   // lane l holds step l
        v32 xv;
#pragma unroll
        for (int e = 0; e < 8; e++)
#pragma unroll
            for (int h = 0; h < 4; h++) xv[4 * e + h] = x[e][h];
        double acc = 0.0;
        while (m) {
            const int j = __builtin_ctzll(m);
            m &= m - 1;
            const uint32_t v = xv[j >> 1];
            const uint32_t cc = (v >> ((j & 1) * 16)) & 0xFFFFu;
            acc = fma(psk_readlane_f64(ddv, j), (double)cc, acc);
        }
        gr = (gr - acc) + cr * R.s;
    };
    // ... and for ALL the rows of this wave at once, when their counts against the period's block are the ones in xA: one
    // walk over the non-zero steps, the lane read of the step and the register index shared by the four rows, whose
    // multiply-adds are independent of each other (a row at a time, a non-zero step was a chain of ~12 dependent
    // instructions: ~2,000 cycles per row and period where this takes ~1,500 for the four)
    auto consume_sparse_rows = [&](int P, unsigned rows) __attribute__((always_inline)) {
        typedef uint32_t v32 __attribute__((ext_vector_type(32)));
        const LcRec &R = rec[P & 3];
        const uint64_t nzv = R.nz;
        uint64_t m = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(nzv >> 32)) << 32) |
                     (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)nzv);
        const double ddv = R.dd[lane];
        v32 xv[R4];
        double acc[R4];
#pragma unroll
        for (int r = 0; r < R4; r++) {
            acc[r] = 0.0;
#pragma unroll
            for (int e = 0; e < 8; e++)
#pragma unroll
                for (int h = 0; h < 4; h++) xv[r][4 * e + h] = xA[r][e][h];
        }
        while (m) {
            const int j = __builtin_ctzll(m);
            m &= m - 1;
            const double ddj = psk_readlane_f64(ddv, j);
            const int idx = j >> 1, sh = (j & 1) * 16;
#pragma unroll
            for (int r = 0; r < R4; r++) acc[r] = fma(ddj, (double)((xv[r][idx] >> sh) & 0xFFFFu), acc[r]);
        }
#pragma unroll
        for (int r = 0; r < R4; r++)
            if ((rows >> r) & 1u) g[r] = (g[r] - acc[r]) + cnt[r] * R.s;
    };
    // brings row `only` (or every row: -1) up to date with the periods before `upto`, period by period.  Rows whose counts
    // against the period's block are in xA and a period with few non-zero steps: all rows at once (above).  Else row by row;
    // a block that is not in xA (the wave that stepped a block while the period before it was dense has that period's
    // steps left for its other rows) is loaded for row r + 1 while row r is worked on.  ONE call site and one instance of
    // the 64 multiply-adds per row: with three call sites and three chunk sources the loop body was ~100 KB of code for four
    // waves that each walk a different part of it -- more than the instruction cache.
#ifdef PSK_LC_STATS
    long long st_rows = 0, st_fresh = 0, st_wait = 0, st_joint = 0;
#endif
    auto bring = [&](int upto, int only) __attribute__((always_inline)) {
#ifdef PSK_LC_STATS
        {   // what of a call is waiting for the loads asked for a period ago
            const long long w0 = clock64();
            __builtin_amdgcn_s_waitcnt(0x0F70);
            st_wait += clock64() - w0;
        }
#endif
        int Pmin = upto;
#pragma unroll
        for (int r = 0; r < R4; r++)
            if ((only < 0 || r == only) && ap[r] < Pmin) Pmin = ap[r];
        for (int P = Pmin; P < upto; P++) {
            unsigned rows = 0;   // the rows that receive period P now (wave-uniform)
#pragma unroll
            for (int r = 0; r < R4; r++)
                if ((only < 0 || r == only) && ap[r] <= P && needs(4 * r + wave, P)) rows |= 1u << r;
            if (rows == 0) continue;
            const bool sparse = __popcll(rec[P & 3].nz) <= 16, inA = P % nb == blkA;
            if (sparse && inA) {
#ifdef PSK_LC_STATS
                st_joint++;
#endif
                consume_sparse_rows(P, rows);
                continue;
            }
            u4 xB[2][8];
            auto askB = [&](int r) __attribute__((always_inline)) {
                const u4 *src = chunk(4 * r + wave, P % nb);
#pragma unroll
                for (int e = 0; e < 8; e++) xB[r & 1][e] = src[64 * e];
            };
            if (!inA && (rows & 1u)) askB(0);
#pragma unroll
            for (int r = 0; r < R4; r++) {
                if (!inA && r + 1 < R4 && ((rows >> (r + 1)) & 1u)) askB(r + 1);
                if ((rows >> r) & 1u) {
                    u4 x[8];
                    if (inA) {
#pragma unroll
                        for (int e = 0; e < 8; e++) x[e] = xA[r][e];
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; e++) x[e] = xB[r & 1][e];
                    }
#ifdef PSK_LC_STATS
                    st_rows++;
                    if (!inA) st_fresh++;
#endif
                    if (sparse) consume_sparse(g[r], cnt[r], x, P);
                    else consume(g[r], cnt[r], x, P);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R4; r++)
            if (only < 0 || r == only) ap[r] = upto;
    };
    // The 64 steps of a block read one column of its diagonal block each: from LDS, where the wave has put the block
    // while it was not stepping (read from global memory as the steps went -- eight loads ahead -- a step took ~700 cycles:
    // the counts of 11 folds, 25 MB, do not stay in a 4-MB L2)
    double *lq = lq_all + (size_t)wave * (4096 + 512);   // (+ 512: the steps read eight columns ahead, past the block's end)
    int lds_blk = -1;
    auto stage = [&](int cn) __attribute__((always_inline)) {
        const u4 *src = reinterpret_cast<const u4 *>(Dgf + (size_t)cn * 4096);
        u4 *dst = reinterpret_cast<u4 *>(lq);
#pragma unroll 1
        for (int it = 0; it < 32; it += 8) {
            u4 t[8];
#pragma unroll
            for (int e = 0; e < 8; e++) t[e] = src[(it + e) * 64 + lane];
#pragma unroll
            for (int e = 0; e < 8; e++) dst[(it + e) * 64 + lane] = t[e];
        }
        lds_blk = cn;
    };
    if (wave < nb) stage(wave);
    for (int i = lane; i < 512; i += 64) lq[4096 + i] = 0.0;
    int sweeps = 0;
    double gap = tol_s + 1.0;
#ifdef PSK_LC_STATS   // make EXTRA=-DPSK_LC_STATS: where a wave's cycles go (printed for the fits that reach the sweep limit)
    long long st_bring_s = 0, st_seq = 0, st_pub = 0, st_bring_o = 0, st_pref = 0, st_bar = 0, st_quiet = 0;
    const long long st_start = clock64();
#define LC_T(var) do { const long long t_ = clock64(); var += t_ - st_t; st_t = t_; } while (0)
#else
#define LC_T(var) do { } while (0)
#endif
    for (int T = 0;; T++) {
        const int c = T % nb, sw = c & 3, rs = c >> 2;
#ifdef PSK_LC_STATS
        long long st_t = clock64();
#endif
        // the sweep that ended with the period before this one: scikit-learn's test for evaluating the gap was made by the
        // wave that stepped its last block
        const bool swept = T > 0 && c == 0;
        const bool chk = swept && rec[(T - 1) & 3].check != 0;
        // this period's steps need the stepped block's rows up to date; a gap needs every row
        // (the stepping wave: only the block's rows while the period before it was dense -- the rest waits a period, four
        // rows of 64 multiply-adds would sit in front of the steps --; every row when few steps were taken: the joint walk is
        // cheap, nothing is left over, and the wave asks for its counts against this block like everybody else)
        const bool lean = T > 0 && __popcll(rec[(T - 1) & 3].nz) <= 16;
        bring(T, (wave == sw && !chk && !lean) ? rs : -1);
        if (swept) sweeps++;
        if (chk) {
            // the gap as scikit-learn evaluates it, from g = X'R, w, X'y and y'y
            double dn = 0.0, wg = 0.0, wq = 0.0, l1 = 0.0;
#pragma unroll
            for (int r = 0; r < R4; r++) {
                const int k = 64 * (4 * r + wave) + lane;
                const double xty = (4 * r + wave < nb && k < p) ? q0f[k] : 0.0;
                dn = fmax(dn, fabs(g[r]));
                wg += wv[r] * g[r];
                wq += wv[r] * xty;
                l1 += fabs(wv[r]);
            }
            dn = psk_wave_max_f64_dpp(dn);
            wg = psk_wave_sum_f64_dpp(wg);
            wq = psk_wave_sum_f64_dpp(wq);
            l1 = psk_wave_sum_f64_dpp(l1);
            if (lane == 0) { s_red[wave][0] = dn; s_red[wave][1] = wg; s_red[wave][2] = wq; s_red[wave][3] = l1; }
            __syncthreads();
            const double dual = fmax(fmax(s_red[0][0], s_red[1][0]), fmax(s_red[2][0], s_red[3][0]));
            const double WG = (s_red[0][1] + s_red[1][1]) + (s_red[2][1] + s_red[3][1]);
            const double WQ = (s_red[0][2] + s_red[1][2]) + (s_red[2][2] + s_red[3][2]);
            const double L1 = (s_red[0][3] + s_red[1][3]) + (s_red[2][3] + s_red[3][3]);
            const double Ry = yy - WQ, RR = Ry - WG;     // R'y = y'y - w'X'y;  R'R = R'y - w'X'R
            double cst = 1.0;
            if (dual > an) {
                cst = an / dual;
                gap = 0.5 * (RR + RR * (cst * cst));
            } else gap = RR;
            gap += an * L1 - cst * Ry;
            __syncthreads();   // (s_red is free again)
            if (gap < tol_s) break;
        }
        if (sweeps >= max_iter) break;   // (the last sweep allowed always evaluates the gap)
        if (wave == sw) {
            LC_T(st_bring_s);
            // ---- this wave steps block c
            double gs = 0.0, ws = 0.0, cs = 0.0;
#pragma unroll
            for (int r = 0; r < R4; r++)
                if (r == rs) { gs = g[r]; ws = wv[r]; cs = cnt[r]; ap[r] = T + 1; }   // (its own steps are applied as they are taken)
            const double ns = cs - cs * cs / ntr;                  // sum (x - mean)^2 with x^2 = x: the expression of the diagonal of Dg
            const double is = ns > 0.0 ? 1.0 / ns : 0.0;           // (a column all 0 or all 1 on the training rows never moves)
            double ddc = 0.0;
            // a block whose coordinates are all at zero and inside the dead zone cannot move: every step would be exactly 0
            const bool quiet = (ws == 0.0 && fabs(gs) <= an) || is == 0.0;
            const uint64_t loud = __ballot(!quiet);
            if (loud != 0ull && lds_blk != c) stage(c);   // (only when this wave steps two blocks in a row: the sweep's wrap with nb = 4 m + 1)
            if (loud != 0ull && __popcll(loud) <= 24) {
                // Few coordinates of the block can move: only they are stepped.  A coordinate at zero inside the dead zone takes
                // a step of exactly 0 whenever it is visited, and what it sees only changes when ANOTHER coordinate moves -- so
                // from the current position the next coordinate that is NOT in that state is found by one ballot, stepped, and
                // the search goes on behind it with the updated g: the visits skipped are exactly the no-ops of the full walk.
                const double *dg = lq + lane;
                int pos = 0;
                while (pos < 64) {
                    const double t = fma(ws, ns, gs);
                    const bool act = is != 0.0 && (ws != 0.0 || fabs(t) > an);
                    const uint64_t mk = __ballot(act) & (~0ull << pos);
                    if (mk == 0ull) break;
                    const int j = __builtin_ctzll(mk);
                    const double qj = dg[j * 64];
                    const double mag = fabs(t) - an;
                    const double wn = copysign(mag > 0.0 ? mag : 0.0, t) * is;
                    const double ddl = wn - ws;
                    const double ddj = psk_readlane_f64(ddl, j);
                    const bool me = lane == j;
                    ws = me ? wn : ws;
                    ddc = me ? ddl : ddc;
                    gs = fma(-ddj, qj, gs);
                    pos = j + 1;
                }
            } else if (loud != 0ull) {
                const double *dg = lq + lane;
                double q[8];
#pragma unroll
                for (int u = 0; u < 8; u++) q[u] = dg[u * 64];
#pragma unroll 1
                for (int j0 = 0; j0 < 64; j0 += 8) {
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const int j = j0 + u;
                        const double qj = q[u];
                        q[u] = dg[(j + 8) * 64];   // (eight columns ahead, unconditionally: past the block's end lie 512 zeros)
                        // every lane evaluates the step of ITS coordinate from what it holds; lane j's is the one that counts
                        const double t = fma(ws, ns, gs);
                        const double mag = fabs(t) - an;
                        const double wn = copysign(mag > 0.0 ? mag : 0.0, t) * is;
                        const double ddl = wn - ws;
                        const double ddj = psk_readlane_f64(ddl, j);
                        const bool me = lane == j;
                        ws = me ? wn : ws;
                        ddc = me ? ddl : ddc;
                        gs = fma(-ddj, qj, gs);
                    }
                }
            }
#ifdef PSK_LC_STATS
            if (loud == 0ull) st_quiet++;
#endif
            LC_T(st_seq);
#pragma unroll
            for (int r = 0; r < R4; r++)
                if (r == rs) { g[r] = gs; wv[r] = ws; }
            // what the others need, and the sweep's largest step / coefficient (scikit-learn's d_w_max, w_max)
            LcRec &R = rec[T & 3];
            R.dd[lane] = ddc;
            const uint64_t nz = __ballot(ddc != 0.0);
            const double sv = psk_wave_sum_f64_dpp(ddc * cs) * inv_n;
            // (DPP butterflies: the shuffle form is twelve ds_bpermute round trips, ~1,500 cycles on the steps' critical path)
            const double dm = psk_wave_max_f64_dpp(fabs(ddc)), wm = psk_wave_max_f64_dpp(fabs(ws));
            const double dmax = fmax(c == 0 ? 0.0 : s_dmax, dm), wmax = fmax(c == 0 ? 0.0 : s_wmax, wm);
            const bool last = c == nb - 1;
            const int check = last && (wmax == 0.0 || dmax / wmax < tol || sweeps == max_iter - 1);
            if (lane == 0) { R.s = sv; R.nz = nz; R.check = check; s_dmax = dmax; s_wmax = wmax; }
            LC_T(st_pub);
        }
        bool all_up = true;   // every row of this wave has every period before this one (the stepping wave: if it took the lean way)
#pragma unroll
        for (int r = 0; r < R4; r++) all_up = all_up && ap[r] >= T;
        if (wave == sw && !all_up) {
            // (its other rows still need the counts in xA)
        } else {
            LC_T(st_bring_o);
            // ---- everybody else asks for its rows' counts against the block being stepped now, and puts the block it
            // steps next into LDS
#pragma unroll
            for (int r = 0; r < R4; r++)
                if (4 * r + wave < nb) {
                    const u4 *src = chunk(4 * r + wave, c);
#pragma unroll
                    for (int e = 0; e < 8; e++) xA[r][e] = src[64 * e];
                }
            blkA = c;
            if (wave != sw) {
                int cn = c + ((wave - c) & 3);
                if (cn >= nb) cn = wave;
                if (wave < nb && cn != lds_blk) stage(cn);
            }
            LC_T(st_pref);
        }
        __syncthreads();   // the steps of period T are published
        LC_T(st_bar);
    }
#ifdef PSK_LC_STATS
    if (lane == 0 && sweeps >= max_iter)
        printf("fit %d wave %d sweeps %d periods %d: total %lld  step-bring %lld seq %lld publish %lld | bring %lld prefetch %lld | barrier %lld  quiet blocks %lld\n",
               fit, wave, sweeps, sweeps * nb, (long long)(clock64() - st_start), st_bring_s, st_seq, st_pub, st_bring_o, st_pref, st_bar, st_quiet);
    if (lane == 0 && sweeps >= max_iter) printf("fit %d wave %d row passes %lld (fresh %lld) joint sparse passes %lld wait for prefetched loads %lld\n", fit, wave, st_rows, st_fresh, st_joint, st_wait);
#endif
#undef LC_T
    // coefficients and the intercept of the uncentred problem: mean_y - sum mean_k w_k
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < R4; r++) {
        const int k = 64 * (4 * r + wave) + lane;
        if (4 * r + wave < nb && k < p) coef[(size_t)fit * p + k] = wv[r];
        acc += cnt[r] * inv_n * wv[r];
    }
    acc = psk_wave_sum_f64_dpp(acc);
    if (lane == 0) s_red[wave][0] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        icpt[fit] = ym - ((s_red[0][0] + s_red[1][0]) + (s_red[2][0] + s_red[3][0]));
        iters[fit] = sweeps;
        if (gaps) gaps[fit] = gap;
    }
}

}  // namespace

hipError_t psk_lasso_cov_launch(const psk_lasso_cov_args &a)
{
    const int nb = a.PP / 64;
    lasso_cooc_kernel<<<dim3(nb, nb, a.n_folds), LC_THREADS, 0, a.stream>>>(a.bits, a.tmask, a.C, a.W, a.PP);
    lasso_xty_kernel<<<dim3(a.PP, a.n_folds), 64, 0, a.stream>>>(a.bits, a.yc, a.q0, a.W, a.PP, a.n);
    lasso_diag_kernel<<<dim3(nb, a.n_folds), LC_THREADS, 0, a.stream>>>(a.C, a.fstat, a.Dg, a.PP);
    const int r4 = (nb + 3) / 4;
    auto kern = r4 <= 1 ? lasso_cov_kernel<1> : r4 == 2 ? lasso_cov_kernel<2> : r4 == 3 ? lasso_cov_kernel<3> : lasso_cov_kernel<4>;
    const size_t lds = 4 * (4096 + 512) * sizeof(double);
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    kern<<<a.n_blocks, LC_THREADS, lds, a.stream>>>(a.C, a.Dg, a.q0, a.fstat, a.block_fit, a.fit_param, a.fit_fidx, a.p, a.PP, a.tol,
                                                  a.max_iter, a.coef, a.icpt, a.iters, a.gaps);
    return hipGetLastError();
}
