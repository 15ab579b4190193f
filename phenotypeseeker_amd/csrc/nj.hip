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
    nj_kernel<<<1, NJ_THREADS, 0, ctx->stream>>>(D, n, mi, mj, d1, d2, last);
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
