// Neighbour joining of the Mash distance matrix (was Bio.Phylo's DistanceTreeConstructor.nj in
// Samples.get_weights, modeling.py:447-458): N - 2 sequential joins, each O(m^2) work that parallelises --
// row sums, the minimum of d[i][j] - r[i] - r[j] over the lower triangle, the update of one row/column.
// One workgroup runs all joins (they are strictly sequential); the matrix stays in L2 (N = 1024: 8 MB), joined
// rows are dropped from an index list (in LDS, with the node distances) instead of being moved.  Every thread walks
// one column; the walk is latency-bound, so the loads of 8 steps are issued before their (ordered) adds / compares.
// The arithmetic is the reference library's, in its order -- left-to-right row sums over the current order, (d - r_i) - r_j, first minimum in (i ascending,
// j < i ascending) scan order, the scan's (1, 0) -> (0, 1) start-up quirk -- so the merge list (and the tree the
// host builds from it) is bit-identical to the scalar loops (tests: against weights.nj on random matrices with
// ties).  Compiled with -ffp-contract=off like the rest of the library.
#include <cstddef>

#include "dev_utils.h"
#include "psk_internal.h"

namespace {

constexpr int NJ_THREADS = 1024;
constexpr int NJ_MAX = 4 * NJ_THREADS;  // leaves: the index list and the node distances live in LDS (48 KB)
constexpr int NJ_UNROLL = 8;

__global__ __launch_bounds__(NJ_THREADS) void nj_kernel(double *__restrict__ D, int n, int32_t *__restrict__ mi_out,
                                                        int32_t *__restrict__ mj_out, double *__restrict__ d1_out,
                                                        double *__restrict__ d2_out, double *__restrict__ last_out)
{
    __shared__ double s_val[NJ_THREADS];
    __shared__ int s_i[NJ_THREADS], s_j[NJ_THREADS];
    __shared__ int s_mi, s_mj;
    __shared__ int32_t idx[NJ_MAX];   // position -> physical row/column
    __shared__ double nd[NJ_MAX];
    const int tid = threadIdx.x;
    for (int p = tid; p < n; p += NJ_THREADS) idx[p] = p;
    __syncthreads();
    int m = n;
    for (int it = 0; m > 2; it++, m--) {
        // node_dist of every position: row sum in position order, / (m - 2).  Column walk of the symmetric matrix
        // so that neighbouring threads read neighbouring addresses.
        for (int p = tid; p < m; p += NJ_THREADS) {
            const int pi = idx[p];
            double acc = D[(size_t)idx[0] * n + pi];
            int q = 1;
            for (; q + NJ_UNROLL <= m; q += NJ_UNROLL) {
                double v[NJ_UNROLL];
#pragma unroll
                for (int u = 0; u < NJ_UNROLL; u++) v[u] = D[(size_t)idx[q + u] * n + pi];
#pragma unroll
                for (int u = 0; u < NJ_UNROLL; u++) acc += v[u];
            }
            for (; q < m; q++) acc += D[(size_t)idx[q] * n + pi];
            nd[p] = acc / (double)(m - 2);
        }
        __syncthreads();
        // per-thread minimum over its rows i (j < i ascending, strict <), then the block minimum by (value, i)
        double best = INFINITY;
        int bi = 0x7fffffff, bj = 0;
        for (int p = tid; p < m; p += NJ_THREADS) {
            if (p == 0) continue;
            const int pi = idx[p];
            const double ri = nd[p];
            double rb = INFINITY;
            int rj = 0;
            int q = 0;
            for (; q + NJ_UNROLL <= p; q += NJ_UNROLL) {
                double v[NJ_UNROLL];
#pragma unroll
                for (int u = 0; u < NJ_UNROLL; u++) v[u] = D[(size_t)idx[q + u] * n + pi];
#pragma unroll
                for (int u = 0; u < NJ_UNROLL; u++) {
                    const double t = (v[u] - ri) - nd[q + u];
                    if (t < rb) { rb = t; rj = q + u; }
                }
            }
            for (; q < p; q++) {
                const double t = (D[(size_t)idx[q] * n + pi] - ri) - nd[q];
                if (t < rb) { rb = t; rj = q; }
            }
            if (rb < best || (rb == best && p < bi)) { best = rb; bi = p; bj = rj; }
        }
        s_val[tid] = best; s_i[tid] = bi; s_j[tid] = bj;
        __syncthreads();
        for (int off = NJ_THREADS / 2; off > 0; off >>= 1) {
            if (tid < off) {
                const double v = s_val[tid + off];
                const int vi = s_i[tid + off];
                if (v < s_val[tid] || (v == s_val[tid] && vi < s_i[tid])) { s_val[tid] = v; s_i[tid] = vi; s_j[tid] = s_j[tid + off]; }
            }
            __syncthreads();
        }
        if (tid == 0) {
            int mi = s_i[0], mj = s_j[0];
            if (mi == 1 && mj == 0) { mi = 0; mj = 1; }  // the library's scan starts from this pair the other way round
            s_mi = mi; s_mj = mj;
            const double dij = D[(size_t)idx[mi] * n + idx[mj]];
            const double d1 = (dij + nd[mi] - nd[mj]) / 2.0;
            mi_out[it] = mi; mj_out[it] = mj;
            d1_out[it] = d1;
            d2_out[it] = dij - d1;
        }
        __syncthreads();
        const int mi = s_mi, mj = s_mj;
        const int pmi = idx[mi], pmj = idx[mj];
        const double dij = D[(size_t)pmi * n + pmj];
        // new distances of the joined node (kept in row/column mj)
        for (int p = tid; p < m; p += NJ_THREADS) {
            if (p == mi || p == mj) continue;
            const int pk = idx[p];
            const double v = (D[(size_t)pmi * n + pk] + D[(size_t)pmj * n + pk] - dij) / 2.0;
            D[(size_t)pmj * n + pk] = v;
            D[(size_t)pk * n + pmj] = v;
        }
        __syncthreads();
        // drop position mi (order of the rest preserved)
        int moved[4];  // up to 4096 leaves
        int cnt = 0;
        for (int p = tid; p < m - 1; p += NJ_THREADS) moved[cnt++] = (p >= mi) ? idx[p + 1] : idx[p];
        __syncthreads();
        cnt = 0;
        for (int p = tid; p < m - 1; p += NJ_THREADS) idx[p] = moved[cnt++];
        __syncthreads();
    }
    if (tid == 0) last_out[0] = D[(size_t)idx[1] * n + idx[0]];
}

// ---- the same joins on many workgroups (r04) ------------------------------------------------------------------------------
// One workgroup reads the whole matrix twice per join through ONE compute unit's 64 B / clock: 8 MB x 2 at 1,024 leaves,
// ~100 us for the first joins, 89 ms for the tree.  Here workgroup w OWNS the columns 64 w .. 64 w + 63 of
// the matrix (every row of them; one column per lane, 512 contiguous bytes per row and wave) and never reads another
// column: by symmetry its column sums are its rows' sums, the scan of row i over j < i is a walk down column i, and the new
// distances of a joined node are computed column by column from rows mi and mj.  What has to cross workgroups is small and
// goes through agent-scope atomics (no cache flushes: a column slice is only ever touched by its own compute unit):
//   barrier 1   the node distances of every position (m doubles), read back into every workgroup's LDS;
//   barrier 2   each workgroup's best (value, i, j, d[i][j]); every workgroup reduces the <= 64 of them itself;
//   barrier 3   the joined node's new distances by position (m doubles): the owner of column mj writes that column.
// The arithmetic, its order and the tie-breaks are nj_kernel's -- the merge lists are bit-identical (tests/test_weights.py
// runs both against the host loop).  Every loop over the rows keeps NJG_UNROLL loads in flight: the chains are latency-bound.
constexpr int NJG_T = 64, NJG_UNROLL = 64, NJG_XB = 16;   // lanes / columns per workgroup; loads in flight down a column; per lane of an exchange

struct NjgCand { double val, dij; int i, j; };
struct NjgShared {
    unsigned bar, fail, pad[14];
    NjgCand cand[NJ_MAX / NJG_T];
    double nd[NJ_MAX], vnew[NJ_MAX];
};

// What crosses workgroups goes through agent-scope atomics: relaxed loads and stores that the hardware keeps coherent
// whatever compute unit or XCD a workgroup runs on (no cache flush anywhere: a column slice is private to its owner), ~2 us
// per hop on this part.  (Tried, r04: the same exchanges as returning atomics executed in the one L2 the participants
// share -- workgroups 0, 8, 16, ... of a launch run on XCD 0, checked through HW_REG_XCC_ID --: correct, and four times
// SLOWER, 100 us per join: sixteen pollers' read-modify-writes on the barrier word queue up in the L2's atomic unit.)
__device__ __forceinline__ void njg_put(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double njg_get(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// false: the barrier timed out (a participant is missing: never seen; the caller gives up and the host falls back)
__device__ __forceinline__ bool njg_barrier(NjgShared *sh, unsigned &target, unsigned nwg)
{
    __builtin_amdgcn_s_waitcnt(0);   // this lane's atomic stores have reached the coherence point
    __syncthreads();
    target += nwg;
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&sh->bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(&sh->bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > (1u << 22)) { ok = false; break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    ok = __syncthreads_and(ok);
    return ok;
}

__global__ __launch_bounds__(NJG_T) void nj_grid_kernel(double *__restrict__ D, int n, int nwg, NjgShared *__restrict__ sh,
                                                        int32_t *__restrict__ mi_out, int32_t *__restrict__ mj_out,
                                                        double *__restrict__ d1_out, double *__restrict__ d2_out,
                                                        double *__restrict__ last_out)
{
    // workgroup b runs on XCD b % 8: the participants are the workgroups of ONE XCD (one L2 behind all of them)
    if (blockIdx.x & 7) return;
    const int w = blockIdx.x >> 3, tid = threadIdx.x;
    __shared__ int32_t idx[NJ_MAX];   // position -> physical row / column (every workgroup keeps the same list)
    __shared__ double nd[NJ_MAX];
    for (int p = tid; p < n; p += NJG_T) idx[p] = p;
    __syncthreads();
    const int c = w * NJG_T + tid;            // this lane's column
    int mypos = c < n ? c : -1;               // ... and its position in the current order (-1: beyond n, or joined away)
    const double *col = D + (c < n ? c : 0);  // D[row * n + c]
    unsigned target = 0;
    int m = n;
    for (int it = 0; m > 2; it++, m--) {
        // ---- node distances: the column sum in position order, / (m - 2)
        if (mypos >= 0) {
            double acc = col[(size_t)idx[0] * n];
            int q = 1;
            for (; q + NJG_UNROLL <= m; q += NJG_UNROLL) {
                double v[NJG_UNROLL];
#pragma unroll
                for (int u = 0; u < NJG_UNROLL; u++) v[u] = col[(size_t)idx[q + u] * n];
#pragma unroll
                for (int u = 0; u < NJG_UNROLL; u++) acc += v[u];
            }
            for (; q < m; q++) acc += col[(size_t)idx[q] * n];
            njg_put(&sh->nd[mypos], acc / (double)(m - 2));
        }
        if (!njg_barrier(sh, target, nwg)) { if (tid == 0) __hip_atomic_store(&sh->fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
        // (NJG_XB reads in flight per lane, one wait)
        for (int base = 0; base < m; base += NJG_T * NJG_XB) {
            double v[NJG_XB];
#pragma unroll
            for (int u = 0; u < NJG_XB; u++) {
                const int p = base + u * NJG_T + tid;
                v[u] = njg_get(&sh->nd[p < m ? p : 0]);
            }
#pragma unroll
            for (int u = 0; u < NJG_XB; u++) {
                const int p = base + u * NJG_T + tid;
                if (p < m) nd[p] = v[u];
            }
        }
        __syncthreads();
        // ---- this lane's row (its column, by symmetry): the first minimum of (d - r_i) - r_j over j < i
        double best = INFINITY, bd = 0.0;
        int bi = 0x7fffffff, bj = 0;
        if (mypos >= 1) {
            const int p = mypos;
            const double ri = nd[p];
            int q = 0;
            for (; q + NJG_UNROLL <= p; q += NJG_UNROLL) {
                double v[NJG_UNROLL];
#pragma unroll
                for (int u = 0; u < NJG_UNROLL; u++) v[u] = col[(size_t)idx[q + u] * n];
#pragma unroll
                for (int u = 0; u < NJG_UNROLL; u++) {
                    const double t = (v[u] - ri) - nd[q + u];
                    if (t < best) { best = t; bj = q + u; bd = v[u]; }
                }
            }
            for (; q < p; q++) {
                const double dv = col[(size_t)idx[q] * n];
                const double t = (dv - ri) - nd[q];
                if (t < best) { best = t; bj = q; bd = dv; }
            }
            bi = p;
        }
        // the workgroup's best by (value, i): a wave of 64 lanes
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            const double ov = psk_shfl_xor_f64(best, d), od = psk_shfl_xor_f64(bd, d);
            const int oi = __shfl_xor(bi, d, 64), oj = __shfl_xor(bj, d, 64);
            if (ov < best || (ov == best && oi < bi)) { best = ov; bi = oi; bj = oj; bd = od; }
        }
        if (tid == 0) {
            njg_put(&sh->cand[w].val, best);
            njg_put(&sh->cand[w].dij, bd);
            __hip_atomic_store(&sh->cand[w].i, bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sh->cand[w].j, bj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!njg_barrier(sh, target, nwg)) { if (tid == 0) __hip_atomic_store(&sh->fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
        // every workgroup reduces the candidates itself (lane t takes workgroup t's)
        {
            double cv = INFINITY, cd = 0.0;
            int ci = 0x7fffffff, cj = 0;
            if (tid < nwg) {
                cv = njg_get(&sh->cand[tid].val);
                cd = njg_get(&sh->cand[tid].dij);
                ci = __hip_atomic_load(&sh->cand[tid].i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                cj = __hip_atomic_load(&sh->cand[tid].j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) {
                const double ov = psk_shfl_xor_f64(cv, d), od = psk_shfl_xor_f64(cd, d);
                const int oi = __shfl_xor(ci, d, 64), oj = __shfl_xor(cj, d, 64);
                if (ov < cv || (ov == cv && oi < ci)) { cv = ov; ci = oi; cj = oj; cd = od; }
            }
            best = cv; bi = ci; bj = cj; bd = cd;
        }
        int mi = bi, mj = bj;
        if (mi == 1 && mj == 0) { mi = 0; mj = 1; }  // the library's scan starts from this pair the other way round
        const double dij = bd;
        if (w == 0 && tid == 0) {
            const double d1 = (dij + nd[mi] - nd[mj]) / 2.0;
            mi_out[it] = mi; mj_out[it] = mj;
            d1_out[it] = d1;
            d2_out[it] = dij - d1;
        }
        const int pmi = idx[mi], pmj = idx[mj];
        // ---- the joined node (kept in row / column mj): this lane's entry of its row, from rows mi and mj of its column
        if (mypos >= 0 && mypos != mi && mypos != mj) {
            const double v = (col[(size_t)pmi * n] + col[(size_t)pmj * n] - dij) / 2.0;
            D[(size_t)pmj * n + c] = v;
            njg_put(&sh->vnew[mypos], v);
        }
        if (!njg_barrier(sh, target, nwg)) { if (tid == 0) __hip_atomic_store(&sh->fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
        if (pmj / NJG_T == w) {   // the owner of column mj writes that column
            for (int base = 0; base < m; base += NJG_T * NJG_XB) {
                double v[NJG_XB];
#pragma unroll
                for (int u = 0; u < NJG_XB; u++) {
                    const int p = base + u * NJG_T + tid;
                    v[u] = njg_get(&sh->vnew[p < m ? p : 0]);
                }
#pragma unroll
                for (int u = 0; u < NJG_XB; u++) {
                    const int p = base + u * NJG_T + tid;
                    if (p < m && p != mi && p != mj) D[(size_t)idx[p] * n + pmj] = v[u];
                }
            }
        }
        __syncthreads();
        // drop position mi (order of the rest preserved)
        if (mypos == mi) mypos = -1;
        else if (mypos > mi) mypos--;
        for (int base = mi; base < m - 1; base += NJG_T) {   // (a chunk of 64 positions moves down by one at a time)
            const int p = base + tid;
            const int v = p < m - 1 ? idx[p + 1] : 0;
            __syncthreads();
            if (p < m - 1) idx[p] = v;
            __syncthreads();
        }
    }
    // the last pair's distance: in column idx[0], row idx[1]
    if (mypos == 0) last_out[0] = col[(size_t)idx[1] * n];
}

}  // namespace

extern "C" int psk_nj_merges(psk_ctx *ctx, const double *dist, int n, int32_t *mi_out, int32_t *mj_out, double *d1_out,
                             double *d2_out, double *last_out)
{
    if (!ctx) return PSK_EINVAL;
    if (!dist || !mi_out || !mj_out || !d1_out || !d2_out || !last_out) return psk_fail(ctx, PSK_EINVAL, "null buffer");
    if (n < 3 || n > NJ_MAX) return psk_fail(ctx, PSK_EINVAL, "neighbour joining on the GPU takes 3..%d leaves, got %d", NJ_MAX, n);
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nn = (size_t)n * n;
    const size_t bytes = nn * 8 + (size_t)n * (4 + 4 + 8 + 8) + 64;
    PSK_TRY(dev_reserve(ctx, ctx->keysA, bytes));
    uint8_t *b = ctx->keysA.as<uint8_t>();
    double *D = reinterpret_cast<double *>(b);
    double *d1 = D + nn, *d2 = d1 + n, *last = d2 + n;
    int32_t *mi = reinterpret_cast<int32_t *>(last + 1), *mj = mi + n;
    PSK_HIP(ctx, hipMemcpyAsync(D, dist, nn * 8, hipMemcpyHostToDevice, ctx->stream));
    // (PSK_NJ_ONE_WG=1: the one-workgroup kernel, the A/B and the cross-check of the tests)
    const char *one = getenv("PSK_NJ_ONE_WG");
    bool grid_done = false;
    // (below 512 leaves the one workgroup is the faster one: 4 ms against 7 at 256 -- a join costs the grid ~25 us of
    // exchanges whatever its size; 1,024 leaves: 67 against 92 ms, 2,048: 240 against 645)
    if (!(one && *one && strcmp(one, "0") != 0) && (n >= 512 || getenv("PSK_NJ_GRID"))) {
        const int nwg = (n + NJG_T - 1) / NJG_T;
        PSK_TRY(dev_reserve(ctx, ctx->keysB, sizeof(NjgShared)));
        NjgShared *sh = ctx->keysB.as<NjgShared>();
        PSK_HIP(ctx, hipMemsetAsync(sh, 0, offsetof(NjgShared, cand), ctx->stream));
        nj_grid_kernel<<<8 * nwg, NJG_T, 0, ctx->stream>>>(D, n, nwg, sh, mi, mj, d1, d2, last);
        PSK_HIP(ctx, hipGetLastError());
        unsigned fail = 0;
        PSK_HIP(ctx, hipMemcpyAsync(&fail, &sh->fail, 4, hipMemcpyDeviceToHost, ctx->stream));
        PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        grid_done = fail == 0;
        if (getenv("PSK_TRACE")) fprintf(stderr, "psk_nj_merges: %d leaves on %d workgroups: %s\n", n, nwg, grid_done ? "done" : "gave up (a workgroup never arrived): one workgroup");
        // (one of its workgroups never arrived: the matrix may be half-joined -- again)
        if (!grid_done) PSK_HIP(ctx, hipMemcpyAsync(D, dist, nn * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    if (!grid_done) nj_kernel<<<1, NJ_THREADS, 0, ctx->stream>>>(D, n, mi, mj, d1, d2, last);
    PSK_HIP(ctx, hipGetLastError());
    const size_t joins = (size_t)n - 2;
    PSK_HIP(ctx, hipMemcpyAsync(mi_out, mi, joins * 4, hipMemcpyDeviceToHost, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(mj_out, mj, joins * 4, hipMemcpyDeviceToHost, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(d1_out, d1, joins * 8, hipMemcpyDeviceToHost, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(d2_out, d2, joins * 8, hipMemcpyDeviceToHost, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(last_out, last, 8, hipMemcpyDeviceToHost, ctx->stream));
    PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PSK_OK;
}
