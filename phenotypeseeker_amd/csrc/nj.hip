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


// ---- the same joins with the matrix in LDS (r04) ----------------------------------------------------------------------------
// nj_grid_kernel still walks its columns through L2 (64 loads in flight, ~1 us per batch) and pays three counter barriers +
// three read-backs per join: 68 ms at 1,024 leaves.  Here a workgroup owns C columns (C = 64 ... 8: the largest power of two
// whose n x C slice fits 128 KB) and keeps them IN LDS for the whole tree -- 64 workgroups at 1,024 leaves, 256 at 2,048 --
// and nothing is indexed through a position list: a joined-away row stays where it is as a row of -0.0 (the identity of the
// IEEE addition, so the column sum over ALL physical rows in ascending order is the library's left-to-right sum over the
// live ones, bit for bit), its node distance is -inf (its candidates come out +inf and never win the strict <), and the
// positions the merge list wants are popcounts of a live bitmap.  The scan is a flat walk of the slice by all 256 threads
// (a thread's elements t, t + 256, ... are one column, rows ascending), reduced by (value, i, j) = the library's first minimum.
// Exchanges carry their own arrival: a value travels as two 8-byte words (join number << 32 | half of the double), written
// and polled with relaxed agent-scope atomics; a reader takes a value when both halves carry the join it waits for -- one
// hop per exchange instead of store-drain + counter + poll + read-back.  A slot is rewritten one join later, which its
// readers have left by then: a workgroup publishes exchange k + 1 only after it has consumed exchange k, and nobody passes
// exchange k + 1 without everybody's contribution.
// A join's critical path is two hops and ONE chain of n dependent additions, not three hops and two chains: the joined
// node's distances v_k are published by the owners of the columns k, and EVERY workgroup's wave 1 gathers them and sums the
// new node's row itself (the same additions in the same order as the owner's column walk) while wave 0 sums the workgroup's
// own columns and waves 2-3 collect the others' node distances; the owner of the joined node's column writes the v_k into
// its slice on the way.  Same arithmetic, order and tie-breaks as nj_kernel: merge lists bit-identical
// (tests/test_weights.py).  1,024 leaves 13.4 ms (nj_grid_kernel 65, nj_kernel 89), 2,048 leaves 44 ms (240 / 645).
constexpr int NJL_T = 256;
constexpr size_t NJL_SLICE_BYTES = 128 * 1024;
constexpr int NJL_MAXE = 8;               // exchange entries a thread polls at a time
constexpr int NJL_MAXN = 2048;            // leaves: a slice of 8 columns = 128 KB
constexpr unsigned NJL_SPINS = 1u << 21;  // polls before a workgroup gives up (a participant never started): ~1 s

struct NjlX { unsigned long long w[2]; };
struct NjlCandX { unsigned long long w[6]; unsigned long long pad[2]; };   // value, d[i][j], i, j
struct NjlShared { unsigned fail, pad[15]; };

__device__ __forceinline__ void njl_put_word(unsigned long long *p, unsigned tag, unsigned half)
{
    __hip_atomic_store(p, ((unsigned long long)tag << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long njl_get_word(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void njl_put(NjlX *x, double v, unsigned tag)
{
    njl_put_word(&x->w[0], tag, (unsigned)__double2loint(v));
    njl_put_word(&x->w[1], tag, (unsigned)__double2hiint(v));
}

// Member `me` of a team of `team` threads takes the entries e = me + team * u < n of x[] that `want` admits, NJL_MAXE at a
// time, as they arrive -> store(e, value); false: timed out
template <class Want, class Store>
__device__ __forceinline__ bool njl_gather(const NjlX *__restrict__ x, int n, unsigned tag, int me, int team, Want want, Store store)
{
    for (int e0 = me; e0 < n; e0 += team * NJL_MAXE) {
        unsigned pending = 0;
#pragma unroll
        for (int u = 0; u < NJL_MAXE; u++) {
            const int e = e0 + team * u;
            if (e < n && want(e)) pending |= 1u << u;
        }
        unsigned spins = 0;
        while (pending) {
            unsigned long long a[NJL_MAXE], b[NJL_MAXE];
#pragma unroll
            for (int u = 0; u < NJL_MAXE; u++) {
                const int e = (pending >> u) & 1u ? e0 + team * u : me;   // (a harmless address for the entries not waited for)
                a[u] = njl_get_word(&x[e].w[0]);
                b[u] = njl_get_word(&x[e].w[1]);
            }
#pragma unroll
            for (int u = 0; u < NJL_MAXE; u++) {
                if (((pending >> u) & 1u) && (unsigned)(a[u] >> 32) == tag && (unsigned)(b[u] >> 32) == tag) {
                    store(e0 + team * u, __hiloint2double((int)(unsigned)b[u], (int)(unsigned)a[u]));
                    pending &= ~(1u << u);
                }
            }
            if (pending) {
                if (++spins > NJL_SPINS) return false;
                __builtin_amdgcn_s_sleep(1);
            }
        }
    }
    return true;
}

// -0.0 + p[0].x + p[0].y + p[stride].x + ... : `pairs` 16-byte pairs, a multiple of 16.  Two register sets of eight pairs:
// the loads of one are under way while the dependent additions of the other run.  No branch inside the loop and scheduling
// barriers between the groups, the stride a template constant: written as "load next, add these, move" the compiler rotated the loop back into load ->
// wait -> add, and with the loads under `if (more)` it merged the register sets through copies behind an s_waitcnt 0 --
// 21 cycles per addition either way, against the ~9 of the additions themselves.
template <int STRIDE>
__device__ __forceinline__ double njl_chain(const double2 *__restrict__ p, int pairs)
{
    // (a lone wave issues an instruction every ~5 cycles whatever it is: with a run-time stride every read had its own
    // address arithmetic, ~120 instructions per 32 additions = 21 cycles per addition; with the stride a constant the
    // sixteen reads of a round are immediates off one pointer -- and still the compiler loaded the next block early into
    // fresh registers and copied: 16 v_mov_b64 per round, 13 cycles per addition.  So the reads are written out: the
    // compiler sees eight registers go into an asm and come out of it, the s_waitcnt that makes them valid is part of the
    // asm that hands them to the additions.  LDS returns in order: when at most eight reads are outstanding, the eight
    // issued before them have landed.)
    typedef double njl_d2 __attribute__((ext_vector_type(2)));
    double acc = -0.0;
    njl_d2 A[8], B[8];
    uint32_t q = (uint32_t)(uintptr_t)p;   // LDS byte address
    const uint32_t q0 = q;
#define NJL_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define NJL_RD8(X, addr, base) { NJL_RD(X[0], addr, (base + 0) * STRIDE * 16); NJL_RD(X[1], addr, (base + 1) * STRIDE * 16); \
                                 NJL_RD(X[2], addr, (base + 2) * STRIDE * 16); NJL_RD(X[3], addr, (base + 3) * STRIDE * 16); \
                                 NJL_RD(X[4], addr, (base + 4) * STRIDE * 16); NJL_RD(X[5], addr, (base + 5) * STRIDE * 16); \
                                 NJL_RD(X[6], addr, (base + 6) * STRIDE * 16); NJL_RD(X[7], addr, (base + 7) * STRIDE * 16); }
#define NJL_WAIT8(X) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(X[0]), "+v"(X[1]), "+v"(X[2]), "+v"(X[3]), "+v"(X[4]), "+v"(X[5]), "+v"(X[6]), "+v"(X[7]))
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // nothing of the compiler's is in flight when the counting begins
    NJL_RD8(A, q, 0)
    for (int r = 0; r < pairs; r += 16) {
        const uint32_t qn = r + 16 < pairs ? q + 16 * STRIDE * 16 : q0;   // (the last round loads the first block again, for nobody)
        NJL_RD8(B, q, 8)
        NJL_WAIT8(A);
#pragma unroll
        for (int u = 0; u < 8; u++) { acc += A[u].x; acc += A[u].y; }
        NJL_RD8(A, qn, 0)
        NJL_WAIT8(B);
#pragma unroll
        for (int u = 0; u < 8; u++) { acc += B[u].x; acc += B[u].y; }
        q = qn;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(A[4]), "+v"(A[5]), "+v"(A[6]), "+v"(A[7]));   // the stray last block
#undef NJL_RD
#undef NJL_RD8
#undef NJL_WAIT8
    return acc;
}

// The node joined last, in wave 1 of every workgroup: lane L takes the distances v_e of rows e = K L ... K L + K - 1 from the
// columns' owners as they arrive (rows joined away count -0.0, the node itself its diagonal +0.0), hands them to the owner
// of the node's column on the way (store), and the wave adds them in row order WITHOUT a copy in LDS: in phase L every
// lane adds its K values to the sum so far, lane L's result is the true one and is broadcast.  n_pad <= 64 K.
template <int K, class Alive, class Store>
__device__ __forceinline__ double njl_newnode_sum(const NjlX *__restrict__ x, int n, int n_pad, unsigned tag, int keep, int lane,
                                                  Alive alive, Store store, bool *ok)
{
    double my[K];
    const int e_base = lane * K;
#pragma unroll
    for (int p = 0; p < K / NJL_MAXE; p++) {
        unsigned pending = 0;
#pragma unroll
        for (int u = 0; u < NJL_MAXE; u++) {
            const int e = e_base + NJL_MAXE * p + u;
            my[NJL_MAXE * p + u] = e == keep ? 0.0 : -0.0;
            if (e < n && e != keep && alive(e)) pending |= 1u << u;
        }
        unsigned spins = 0;
        while (pending) {
            unsigned long long a[NJL_MAXE], b[NJL_MAXE];
#pragma unroll
            for (int u = 0; u < NJL_MAXE; u++) {
                const int e = (pending >> u) & 1u ? e_base + NJL_MAXE * p + u : 0;   // (a harmless address for the entries not waited for)
                a[u] = njl_get_word(&x[e].w[0]);
                b[u] = njl_get_word(&x[e].w[1]);
            }
#pragma unroll
            for (int u = 0; u < NJL_MAXE; u++) {
                if (((pending >> u) & 1u) && (unsigned)(a[u] >> 32) == tag && (unsigned)(b[u] >> 32) == tag) {
                    const double v = __hiloint2double((int)(unsigned)b[u], (int)(unsigned)a[u]);
                    my[NJL_MAXE * p + u] = v;
                    store(e_base + NJL_MAXE * p + u, v);
                    pending &= ~(1u << u);
                }
            }
            if (pending) {
                if (++spins > NJL_SPINS) { *ok = false; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
    }
    double acc = -0.0;
    const int phases = (n_pad + K - 1) / K;
    for (int L = 0; L < phases; L++) {
        double t = acc;
#pragma unroll
        for (int k = 0; k < K; k++) t += my[k];
        acc = psk_readlane_f64(t, L);
    }
    return acc;
}

struct NjlBest { double val, dij; int i, j; };
__device__ __forceinline__ bool njl_before(const NjlBest &a, const NjlBest &b)   // the scan order's first minimum
{
    return a.val < b.val || (a.val == b.val && (a.i < b.i || (a.i == b.i && a.j < b.j)));
}
// minimum over the workgroup by (value, i, j), in every thread (s_red: 4 entries; two barriers).  In a wave: the minimum of
// the values by four DPP steps inside the rows of 16 lanes and three scalar minima of the row leaders, then -- among the
// lanes that hold it -- the minimum of i << 16 | j the same way, then the winner's d[i][j] by a lane read: ~50
// instructions without an LDS round trip (six ds_bpermute rounds of six registers each before: a third of a join's 3 us
// between the scan and the candidates' exchange).  i, j < 2^16 (2,048 leaves at most); i = 0x7fffffff: no candidate.
__device__ __forceinline__ uint32_t njl_dpp_u32(uint32_t v, const int tag)
{
    int x = (int)v;
    switch (tag) {  // constant-folded (the builtin wants an immediate control word)
    case 0: x = __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false); break;    // quad_perm [1,0,3,2]
    case 1: x = __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false); break;    // quad_perm [2,3,0,1]
    case 2: x = __builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, 0xF, false); break;   // row_half_mirror
    default: x = __builtin_amdgcn_update_dpp(x, x, 0x140, 0xF, 0xF, false); break;  // row_mirror
    }
    return (uint32_t)x;
}
__device__ __forceinline__ NjlBest njl_wg_min(NjlBest c, NjlBest *s_red)
{
    double v = c.val;
    v = fmin(v, psk_dpp_f64(v, 0));
    v = fmin(v, psk_dpp_f64(v, 1));
    v = fmin(v, psk_dpp_f64(v, 2));
    v = fmin(v, psk_dpp_f64(v, 3));
    const double m = fmin(fmin(psk_readlane_f64(v, 0), psk_readlane_f64(v, 16)), fmin(psk_readlane_f64(v, 32), psk_readlane_f64(v, 48)));
    const bool have = (unsigned)c.i < 0x10000u && (unsigned)c.j < 0x10000u;
    uint32_t key = have && c.val == m ? ((uint32_t)c.i << 16) | (uint32_t)c.j : 0xFFFFFFFFu;
    uint32_t k = key, o;
    o = njl_dpp_u32(k, 0); k = o < k ? o : k;
    o = njl_dpp_u32(k, 1); k = o < k ? o : k;
    o = njl_dpp_u32(k, 2); k = o < k ? o : k;
    o = njl_dpp_u32(k, 3); k = o < k ? o : k;
    const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)k, 0), k1 = (uint32_t)__builtin_amdgcn_readlane((int)k, 16);
    const uint32_t k2 = (uint32_t)__builtin_amdgcn_readlane((int)k, 32), k3 = (uint32_t)__builtin_amdgcn_readlane((int)k, 48);
    const uint32_t k01 = k0 < k1 ? k0 : k1, k23 = k2 < k3 ? k2 : k3, kmin = k01 < k23 ? k01 : k23;
    NjlBest w;
    w.val = m; w.dij = 0.0; w.i = 0x7fffffff; w.j = 0x7fffffff;
    if (kmin != 0xFFFFFFFFu) {   // (wave-uniform)
        const uint64_t who = __ballot(key == kmin);
        const int lw = __builtin_ctzll(who);
        w.dij = psk_readlane_f64(c.dij, lw);
        w.i = (int)(kmin >> 16); w.j = (int)(kmin & 0xFFFFu);
    } else w.val = INFINITY;
    __syncthreads();   // (the previous use of s_red is over)
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = w;
    __syncthreads();
    NjlBest r = s_red[0];
#pragma unroll
    for (int q = 1; q < NJL_T / 64; q++) {
        const NjlBest o2 = s_red[q];
        if (njl_before(o2, r)) r = o2;
    }
    return r;
}

template <int lgC>
__global__ __launch_bounds__(NJL_T) void nj_lds_kernel(const double *__restrict__ D, int n, int nwg, NjlShared *__restrict__ sh,
                                                       NjlX *__restrict__ nd_x, NjlX *__restrict__ vnew_x, NjlCandX *__restrict__ cand_x,
                                                       int32_t *__restrict__ mi_out, int32_t *__restrict__ mj_out,
                                                       double *__restrict__ d1_out, double *__restrict__ d2_out,
                                                       double *__restrict__ last_out)
{
    extern __shared__ __attribute__((aligned(16))) double njl_lds[];
    constexpr int C = 1 << lgC;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, w = blockIdx.x, c0 = w << lgC;
    const int n_pad = (n + 31) & ~31, n_words = (n + 63) >> 6;
    // slice: rows in pairs, [n_pad / 2][C][2]: d[r][c0 + cc] at ((r >> 1) * C + cc) * 2 + (r & 1) -- a lane's 16-byte read is two
    // rows of its column, the lanes' reads lie side by side
    double *slice = njl_lds;
    double *nd_l = slice + (size_t)n_pad * C;             // [n + C]: node distances by physical row (-inf: joined away / beyond n)
    unsigned long long *live = reinterpret_cast<unsigned long long *>(nd_l + n + C);   // [n_words]
    NjlBest *s_red = reinterpret_cast<NjlBest *>(live + n_words + 1);
    __shared__ int s_fail;
    const double NEG0 = -0.0, NINF = -INFINITY;
    auto at = [&](int r, int col) -> size_t { return ((((size_t)(r >> 1) << lgC) + col) << 1) + (r & 1); };
    for (int e = tid; e < n_pad * C; e += NJL_T) {
        const int r = ((e >> (lgC + 1)) << 1) | (e & 1), c = c0 + ((e >> 1) & (C - 1));
        slice[e] = (r < n && c < n) ? D[(size_t)r * n + c] : NEG0;
    }
    for (int e = tid; e < n + C; e += NJL_T) nd_l[e] = NINF;
    for (int e = tid; e < n_words; e += NJL_T) live[e] = e == n_words - 1 && (n & 63) ? (1ull << (n & 63)) - 1ull : ~0ull;
    if (tid == 0) s_fail = 0;
    __syncthreads();
    int f0 = 0, f1 = 1;        // the first two live rows (positions 0 and 1)
    int keep = -1;             // the row of the node joined last
    int m = n;
#ifdef PSK_NJ_STATS
    unsigned long long st_t[6] = {0, 0, 0, 0, 0, 0}, st_c = wall_clock64();
#define NJL_STAT(k) { const unsigned long long now = wall_clock64(); st_t[k] += now - st_c; st_c = now; }
#else
#define NJL_STAT(k)
#endif
    for (int it = 0; m > 2; it++, m--) {
        const unsigned tag = (unsigned)it + 1u;
        if (wave == 0) {
            // ---- node distances of this workgroup's columns: a column per lane, every physical row in order
            const int col = c0 + lane;
            if (lane < C && col < n) {
                const double acc = njl_chain<C>(reinterpret_cast<const double2 *>(slice) + lane, n_pad >> 1);
                const bool alive = (live[col >> 6] >> (col & 63)) & 1ull;
                if (col != keep) njl_put(&nd_x[col], alive ? acc / (double)(m - 2) : NINF, tag);
            }
            NJL_STAT(0)
        } else if (wave == 1) {
            // ---- the node joined last: its distances from the columns' owners; its node distance summed here, by everybody
            if (it > 0) {
                const int kc = (keep >> lgC) == w ? keep - c0 : -1;
                bool ok = true;
                auto alive = [&](int e) { return (bool)((live[e >> 6] >> (e & 63)) & 1ull); };
                auto store = [&](int e, double v) { if (kc >= 0) slice[at(e, kc)] = v; };
                double acc;
                if (n_pad <= 512) acc = njl_newnode_sum<8>(vnew_x, n, n_pad, tag - 1u, keep, lane, alive, store, &ok);
                else if (n_pad <= 1024) acc = njl_newnode_sum<16>(vnew_x, n, n_pad, tag - 1u, keep, lane, alive, store, &ok);
                else acc = njl_newnode_sum<32>(vnew_x, n, n_pad, tag - 1u, keep, lane, alive, store, &ok);
                if (!ok) s_fail = 1;
                if (lane == 0) nd_l[keep] = acc / (double)(m - 2);
            }
            NJL_STAT(0)
        } else {
            // ---- everybody else's node distances
            if (!njl_gather(nd_x, n, tag, tid - 128, 128, [&](int e) { return e != keep; }, [&](int e, double v) { nd_l[e] = v; })) s_fail = 1;
        }
        __syncthreads();
        NJL_STAT(1)
        if (s_fail) { if (tid == 0) __hip_atomic_store(&sh->fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
        // ---- first minimum of (d - r_i) - r_j over j < i, rows i = this workgroup's columns
        NjlBest b;
        b.val = INFINITY; b.dij = 0.0; b.i = 0; b.j = 0x7fffffff;
        {
            const int lim = (((c0 + C < n ? c0 + C : n) + 1) & ~1) << lgC;   // elements of the rows below the last column (whole pairs)
            // thread t takes the row PAIRS t / C, t / C + 256 / C, ... of column t % C: one 16-byte read each of the slice and
            // of the node distances per two candidates, four pairs in flight (one element at a time the loop paid two LDS
            // round trips per element, 5.7 us per join; eight single elements per batch, 5.0: a lone wave's issue rate)
            const int pcol = tid & (C - 1), pc = c0 + pcol;
            const double prc = nd_l[pc];
            const int npairs = lim >> (lgC + 1);
            const double2 *s2 = reinterpret_cast<const double2 *>(slice), *n2 = reinterpret_cast<const double2 *>(nd_l);
            b.i = pc;
            // per candidate two subtractions, a compare, a select of the row and a minimum (d[i][j] of the winner is read again
            // afterwards); the row pairs wholly below the workgroup's first column need no row test
            const int nfree = c0 >> 1;
            double bv = INFINITY;
            int bj = 0x7fffffff;
            for (int g0 = tid >> lgC; g0 < npairs; g0 += 4 * (NJL_T >> lgC)) {
                double2 d[4], rq[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int g = g0 + u * (NJL_T >> lgC) < npairs ? g0 + u * (NJL_T >> lgC) : 0;
                    d[u] = s2[((size_t)g << lgC) + pcol];
                    rq[u] = n2[g];
                }
                __builtin_amdgcn_sched_barrier(0);
                if (g0 + 3 * (NJL_T >> lgC) < nfree) {   // (wave-uniform: the lanes of a wave differ by < 64 / C pairs ... only when C = 64; tested per lane)
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int q = 2 * (g0 + u * (NJL_T >> lgC));
                        const double t0 = (d[u].x - prc) - rq[u].x, t1 = (d[u].y - prc) - rq[u].y;
                        bj = t0 < bv ? q : bj; bv = fmin(bv, t0);
                        bj = t1 < bv ? q + 1 : bj; bv = fmin(bv, t1);
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int g = g0 + u * (NJL_T >> lgC), q = 2 * g;
                        const bool in = g < npairs;
                        const double t0 = in && q < pc ? (d[u].x - prc) - rq[u].x : INFINITY;
                        const double t1 = in && q + 1 < pc ? (d[u].y - prc) - rq[u].y : INFINITY;
                        bj = t0 < bv ? q : bj; bv = fmin(bv, t0);
                        bj = t1 < bv ? q + 1 : bj; bv = fmin(bv, t1);
                    }
                }
            }
            b.val = bv; b.j = bj;
            if (b.j == 0x7fffffff) b.i = 0x7fffffff;
        }
        b = njl_wg_min(b, s_red);
        if (b.j != 0x7fffffff) b.dij = slice[at(b.j, b.i - c0)];   // (every thread the same entry)
        NJL_STAT(2)
        if (tid == 0) {
            NjlCandX *x = &cand_x[w];
            njl_put_word(&x->w[0], tag, (unsigned)__double2loint(b.val)); njl_put_word(&x->w[1], tag, (unsigned)__double2hiint(b.val));
            njl_put_word(&x->w[2], tag, (unsigned)__double2loint(b.dij)); njl_put_word(&x->w[3], tag, (unsigned)__double2hiint(b.dij));
            njl_put_word(&x->w[4], tag, (unsigned)b.i); njl_put_word(&x->w[5], tag, (unsigned)b.j);
        }
        // ---- every workgroup reduces the candidates itself (thread t takes workgroup t's)
        b.val = INFINITY; b.dij = 0.0; b.i = 0x7fffffff; b.j = 0x7fffffff;
        if (tid < nwg) {
            const NjlCandX *x = &cand_x[tid];
            unsigned spins = 0;
            for (;;) {
                unsigned long long v[6];
#pragma unroll
                for (int u = 0; u < 6; u++) v[u] = njl_get_word(&x->w[u]);
                bool ok = true;
#pragma unroll
                for (int u = 0; u < 6; u++) ok = ok && (unsigned)(v[u] >> 32) == tag;
                if (ok) {
                    b.val = __hiloint2double((int)(unsigned)v[1], (int)(unsigned)v[0]);
                    b.dij = __hiloint2double((int)(unsigned)v[3], (int)(unsigned)v[2]);
                    b.i = (int)(unsigned)v[4]; b.j = (int)(unsigned)v[5];
                    break;
                }
                if (++spins > NJL_SPINS) { s_fail = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        b = njl_wg_min(b, s_red);
        NJL_STAT(3)
        if (s_fail || b.i >= n || b.j >= n) { if (tid == 0) __hip_atomic_store(&sh->fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
        // the joined node stays in the row of the pair's lower member j, row i goes -- except for the library's start-up
        // quirk: its scan begins at the pair (position 1, position 0) the other way round
        int kill = b.i;
        keep = b.j;
        if (keep == f0 && kill == f1) { kill = f0; keep = f1; }
        const double dij = b.dij;
        if (wave == 0) {   // the merge record (workgroup 0), the joined node's row, the row that goes
            if (w == 0) {
                int pk = 0, pp = 0;   // positions = live rows below
                for (int l = lane; l < n_words; l += 64) {
                    const unsigned long long lv = live[l];
                    const int lo = l << 6;
                    pk += __popcll(kill >= lo + 64 ? lv : kill > lo ? lv & ((1ull << (kill - lo)) - 1ull) : 0ull);
                    pp += __popcll(keep >= lo + 64 ? lv : keep > lo ? lv & ((1ull << (keep - lo)) - 1ull) : 0ull);
                }
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) { pk += __shfl_xor(pk, d, 64); pp += __shfl_xor(pp, d, 64); }
                if (lane == 0) {
                    const double d1 = (dij + nd_l[kill] - nd_l[keep]) / 2.0;
                    mi_out[it] = pk; mj_out[it] = pp;
                    d1_out[it] = d1;
                    d2_out[it] = dij - d1;
                }
            }
            if (lane < C) {
                const int col = c0 + lane;
                const bool alive = col < n && ((live[col >> 6] >> (col & 63)) & 1ull);
                if (alive && col != kill && col != keep) {
                    const double v = (slice[at(kill, lane)] + slice[at(keep, lane)] - dij) / 2.0;
                    slice[at(keep, lane)] = v;
                    njl_put(&vnew_x[col], v, tag);
                }
                slice[at(kill, lane)] = NEG0;
            }
            if (lane == 0) live[kill >> 6] &= ~(1ull << (kill & 63));
        }
        __syncthreads();
        NJL_STAT(4)
        if (kill == f0 || kill == f1) {   // the first two live rows again (every thread, the same walk)
            int found = 0;
            f0 = f1 = n;
            for (int l = 0; l < n_words && found < 2; l++) {
                unsigned long long lv = live[l];
                while (lv && found < 2) {
                    const int r = (l << 6) + __builtin_ctzll(lv);
                    if (found == 0) f0 = r; else f1 = r;
                    found++;
                    lv &= lv - 1;
                }
            }
        }
    }
    // the last pair's distance: the joined node's row `keep` in the other's column (the column of `keep` itself has not
    // been filled in: nobody needs it any more)
    const int other = keep == f0 ? f1 : f0;
    if ((other >> lgC) == w && tid == 0) last_out[0] = slice[at(keep, other - c0)];
#ifdef PSK_NJ_STATS
    if ((w == 0 || w == nwg - 1) && lane == 0)   // 100 MHz ticks: own chain / gather | wait at the barrier | scan + reduce | candidates | update
        printf("nj_lds wg %d wave %d: role %llu us, barrier %llu, scan %llu, cand %llu, update %llu (n %d)\n", w, wave, st_t[0] / 100, st_t[1] / 100,
               st_t[2] / 100, st_t[3] / 100, st_t[4] / 100, n);
#endif
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
    // (r04) the matrix in LDS, a workgroup per C columns: from PSK_NJ_LDS_MIN leaves on (default 288: below, the one workgroup is as fast -- 256 leaves 4.1 ms either way) while a workgroup per
    // slice fits the compute units; PSK_NJ_LDS=0 switches it off
    const bool one_wg = one && *one && strcmp(one, "0") != 0;
    const char *lds_knob = getenv("PSK_NJ_LDS"), *lds_min = getenv("PSK_NJ_LDS_MIN");
    int lds_from = 288;
    if (lds_min && *lds_min) {   // (a knob of the tests: a whole number of leaves, 3 ... NJ_MAX)
        char *end = nullptr;
        const long v = strtol(lds_min, &end, 10);
        if (*end || v < 3 || v > NJ_MAX) return psk_fail(ctx, PSK_EINVAL, "PSK_NJ_LDS_MIN=%s: expected a number of leaves, 3 ... %d", lds_min, NJ_MAX);
        lds_from = (int)v;
    }
    const int n_pad = (n + 31) & ~31;
    int lgC = 6;
    while (lgC > 3 && (size_t)n_pad * 8 * (1u << lgC) > NJL_SLICE_BYTES) lgC--;
    const int nwg_l = (n + (1 << lgC) - 1) >> lgC;
    if (!one_wg && !(lds_knob && strcmp(lds_knob, "0") == 0) && !getenv("PSK_NJ_GRID") && n >= lds_from &&
        n <= NJL_MAXN && nwg_l <= NJL_T && nwg_l <= (ctx->n_cu > 0 ? ctx->n_cu : 256)) {
        const size_t xbytes = sizeof(NjlShared) + 2 * (size_t)n * sizeof(NjlX) + (size_t)nwg_l * sizeof(NjlCandX);
        PSK_TRY(dev_reserve(ctx, ctx->keysB, xbytes));
        uint8_t *xb = ctx->keysB.as<uint8_t>();
        NjlShared *sh = reinterpret_cast<NjlShared *>(xb);
        NjlX *nd_x = reinterpret_cast<NjlX *>(xb + sizeof(NjlShared)), *vnew_x = nd_x + n;
        NjlCandX *cand_x = reinterpret_cast<NjlCandX *>(vnew_x + n);
        PSK_HIP(ctx, hipMemsetAsync(xb, 0, xbytes, ctx->stream));
        const size_t lds = ((size_t)n_pad * (1u << lgC) + n + (1u << lgC)) * 8 + ((size_t)((n + 63) >> 6) + 1) * 8 + 4 * sizeof(NjlBest) + 64;
        auto launch = [&](auto kern) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            kern<<<nwg_l, NJL_T, lds, ctx->stream>>>(D, n, nwg_l, sh, nd_x, vnew_x, cand_x, mi, mj, d1, d2, last);
        };
        switch (lgC) {
        case 6: launch(nj_lds_kernel<6>); break;
        case 5: launch(nj_lds_kernel<5>); break;
        case 4: launch(nj_lds_kernel<4>); break;
        default: launch(nj_lds_kernel<3>); break;
        }
        PSK_HIP(ctx, hipGetLastError());
        unsigned fail = 0;
        PSK_HIP(ctx, hipMemcpyAsync(&fail, &sh->fail, 4, hipMemcpyDeviceToHost, ctx->stream));
        PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        grid_done = fail == 0;   // (the matrix itself is untouched: the slices live in LDS)
        if (getenv("PSK_TRACE")) fprintf(stderr, "psk_nj_merges: %d leaves in LDS, %d workgroups of %d columns: %s\n", n, nwg_l, 1 << lgC, grid_done ? "done" : "gave up (a workgroup never arrived)");
    }
    if (!grid_done && !one_wg && (n >= 512 || getenv("PSK_NJ_GRID"))) {
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
