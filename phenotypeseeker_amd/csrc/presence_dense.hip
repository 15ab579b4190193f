// a2+a3 for small word spaces: union and presence matrix straight from the samples' presence bitmaps (the dense
// list form of dense_count.hip).  Replaces the glistcompare -u tree (modeling.py:350-380) and the glistquery -l /
// split text mapping (modeling.py:317-348) without a sort and without a list:
//   pd_or_kernel         occ[x] = OR over the samples of word x of their bitmaps (which words of the space occur at
//                        all = the union); rows per chunk of 16 bitmap words
//   (exclusive scan of the chunk row counts -> first matrix row of every chunk, total = M)
//   pd_transpose_kernel  one workgroup per chunk: 256 samples x 16 bitmap words are staged in LDS, every wave turns
//                        64 samples x 64 word values into 64 row words with a 6-stage butterfly (the bit matrix
//                        transposed across the lanes), and the rows that exist are written compacted -- 32 B per
//                        lane, consecutive lanes on consecutive rows
// Traffic: every bitmap is read twice (2 x N x 2^2k / 8 bytes: 4.3 GB at 256 samples, k = 13, against 19 GB of list
// reads in the tiled build) and the matrix is written once.  HBM-bound bit shuffling: no MFMA.
#include "dev_utils.h"
#include "psk_internal.h"

namespace {

constexpr int PD_CHUNK = 16;                 // bitmap words per workgroup = 1024 candidate rows
constexpr int PD_THREADS = 256;
constexpr int PD_ROWLEN = PD_CHUNK + 1;      // padded LDS row (u64): lane l reads row l without a 64-way conflict

__global__ __launch_bounds__(256) void pd_or_kernel(const uint64_t *const *__restrict__ bitmaps, int n, uint64_t n_words,
                                                     uint64_t *__restrict__ occ, uint32_t *__restrict__ chunk_rows)
{
    const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;   // n_words is a multiple of 256
    uint64_t v = 0;
    int s = 0;
    for (; s + 8 <= n; s += 8) {
        uint64_t a[8];
#pragma unroll
        for (int e = 0; e < 8; e++) a[e] = bitmaps[s + e][x];
#pragma unroll
        for (int e = 0; e < 8; e++) v |= a[e];
    }
    for (; s < n; s++) v |= bitmaps[s][x];
    occ[x] = v;
    uint32_t c = (uint32_t)__popcll(v);
#pragma unroll
    for (int d = 1; d < PD_CHUNK; d <<= 1) c += __shfl_xor(c, d, 64);
    if ((threadIdx.x & (PD_CHUNK - 1)) == 0) chunk_rows[x / PD_CHUNK] = c;
    (void)n_words;
}

// 64 x 64 bit matrix, row l in lane l -> its transpose, row j in lane j: the recursive swap of the off-diagonal blocks, six
// stages (block sizes 32 ... 1), WITHOUT the LDS (r04).  As xor-shuffles the partner's word cost two ds_bpermute per stage --
// twelve per block, and the kernel was bound by them (2.4 TB/s).  On gfx950 the two stages that cross rows of 16 lanes are
// lane-swap instructions: v_permlane32_swap(lo, hi) leaves {lo: own or partner's low dword, hi: ...} exactly as stage 32 wants
// them (ONE instruction), v_permlane16_swap(d & 0xFFFF, d >> 16) hands every row the two halves it keeps, per dword; the four
// stages inside a row fetch the partner's dword by DPP (quad permutations for xor 1 / 2; a row shift left in the banks whose
// lanes have the bit clear and a row shift right in the others for xor 4 / 8) and merge with a rotate and a bit-field
// insert whose per-lane mask says which half is kept.
template <int J>
__device__ __forceinline__ uint32_t pd_partner_dpp(uint32_t v)
{
    const int x = (int)v;
    if (J == 1) return (uint32_t)__builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
    if (J == 2) return (uint32_t)__builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    if (J == 4) {
        const int p = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xF, 0x5, false);     // row_shl:4: lane l takes l + 4 (banks 0, 2)
        return (uint32_t)__builtin_amdgcn_update_dpp(p, x, 0x114, 0xF, 0xA, false);  // row_shr:4: lane l takes l - 4 (banks 1, 3)
    }
    const int p = __builtin_amdgcn_update_dpp(x, x, 0x108, 0xF, 0x3, false);         // row_shl:8 (banks 0, 1)
    return (uint32_t)__builtin_amdgcn_update_dpp(p, x, 0x118, 0xF, 0xC, false);      // row_shr:8 (banks 2, 3)
}
template <int J>
__device__ __forceinline__ uint32_t pd_stage_dpp(uint32_t a, int lane)
{
    constexpr uint32_t M = J == 8 ? 0x00FF00FFu : J == 4 ? 0x0F0F0F0Fu : J == 2 ? 0x33333333u : 0x55555555u;
    const uint32_t p = pd_partner_dpp<J>(a);
    const bool odd = (lane & J) != 0;
    const uint32_t s = __builtin_amdgcn_alignbit(p, p, odd ? J : 32 - J);   // bit clear: p << J, set: p >> J (as rotations: the mask drops what wraps)
    const uint32_t mk = odd ? ~M : M;
    return (mk & a) | (~mk & s);
}
__device__ __forceinline__ uint64_t transpose64(uint64_t a, int lane)
{
    uint32_t lo = (uint32_t)a, hi = (uint32_t)(a >> 32);
    {
        const auto r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
        lo = r[0]; hi = r[1];
    }
    {
        const auto r = __builtin_amdgcn_permlane16_swap(lo & 0xFFFFu, lo >> 16, false, false);
        lo = r[0] | (r[1] << 16);
        const auto q = __builtin_amdgcn_permlane16_swap(hi & 0xFFFFu, hi >> 16, false, false);
        hi = q[0] | (q[1] << 16);
    }
    lo = pd_stage_dpp<8>(lo, lane); hi = pd_stage_dpp<8>(hi, lane);
    lo = pd_stage_dpp<4>(lo, lane); hi = pd_stage_dpp<4>(hi, lane);
    lo = pd_stage_dpp<2>(lo, lane); hi = pd_stage_dpp<2>(hi, lane);
    lo = pd_stage_dpp<1>(lo, lane); hi = pd_stage_dpp<1>(hi, lane);
    return ((uint64_t)hi << 32) | lo;
}

__global__ __launch_bounds__(PD_THREADS) void pd_transpose_kernel(const uint64_t *const *__restrict__ bitmaps, int n, int wpr,
                                                                   const uint64_t *__restrict__ occ,
                                                                   const uint32_t *__restrict__ chunk_off, uint64_t word0,
                                                                   uint64_t *__restrict__ union_words, uint64_t *__restrict__ bits)
{
    __shared__ uint64_t tile[256 * PD_ROWLEN];   // 34 KB: [sample of the pass][word of the chunk]
    __shared__ uint64_t s_occ[PD_CHUNK];
    __shared__ uint32_t s_off[PD_CHUNK];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint64_t x0 = (uint64_t)blockIdx.x * PD_CHUNK;
    if (tid < PD_CHUNK) {
        const uint64_t o = occ[x0 + tid];
        s_occ[tid] = o;
        uint32_t c = (uint32_t)__popcll(o), inc = c;
#pragma unroll
        for (int d = 1; d < PD_CHUNK; d <<= 1) {
            const uint32_t t = __shfl_up(inc, d, 64);
            if (tid >= d) inc += t;
        }
        s_off[tid] = chunk_off[blockIdx.x] + inc - c;
    }
    __syncthreads();
    uint64_t any = 0;
#pragma unroll
    for (int e = 0; e < PD_CHUNK; e++) any |= s_occ[e];
    if (!any) return;   // no word of this chunk occurs in any sample
    const int n_groups = (n + 63) / 64;
    for (int gq = 0; gq * 4 < wpr; gq++) {
        // stage 256 samples x 16 words (128 B per sample: one line)
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int p = it * PD_THREADS + tid;
            const int sl = p >> 3, pr = p & 7;
            const int smp = gq * 256 + sl;
            ulonglong2 v = make_ulonglong2(0ull, 0ull);
            if (smp < n) v = *reinterpret_cast<const ulonglong2 *>(bitmaps[smp] + x0 + 2 * pr);
            tile[sl * PD_ROWLEN + 2 * pr] = v.x;
            tile[sl * PD_ROWLEN + 2 * pr + 1] = v.y;
        }
        __syncthreads();
        const int words_here = (wpr - gq * 4) < 4 ? (wpr - gq * 4) : 4;   // 2 or 4: wpr is even -- or 1 (up to 64 samples)
#pragma unroll
        for (int xi = 0; xi < PD_CHUNK / 4; xi++) {
            const int xw = wid * (PD_CHUNK / 4) + xi;
            const uint64_t o = s_occ[xw];
            if (!o) continue;   // wave-uniform
            uint64_t b[4];
#pragma unroll
            for (int gi = 0; gi < 4; gi++) {
                b[gi] = 0;
                if (gq * 4 + gi < n_groups) b[gi] = transpose64(tile[(gi * 64 + lane) * PD_ROWLEN + xw], lane);
            }
            if ((o >> lane) & 1ull) {
                const uint64_t r = (uint64_t)s_off[xw] + (uint32_t)__popcll(o & psk_lanemask_lt(lane));
                uint64_t *dst = bits + r * (uint64_t)wpr + gq * 4;
                if (words_here == 1) *dst = b[0];
                else *reinterpret_cast<ulonglong2 *>(dst) = make_ulonglong2(b[0], b[1]);
                if (words_here == 4) *reinterpret_cast<ulonglong2 *>(dst + 2) = make_ulonglong2(b[2], b[3]);
                if (gq == 0) union_words[r] = word0 + (x0 + xw) * 64 + lane;
            }
        }
    }
}

}  // namespace

int build_presence_dense(psk_ctx *ctx, uint64_t *n_kmers, int *done)
{
    *done = 0;
    if (!ctx->dense_mode || getenv("PSK_NO_DENSE_PRESENCE")) return PSK_OK;
    const int n = ctx->n_samples;
    for (int i = 0; i < n; i++)
        if (!ctx->lists[i].dense || !ctx->lists[i].bitmap) return PSK_OK;   // installed lists (exchange): the list routes
    const uint64_t nw = (uint64_t)ctx->dense_nb * DC_BUCKET_WORDS;
    const uint32_t n_chunks = (uint32_t)(nw / PD_CHUNK);
    std::vector<const uint64_t *> ptrs(n);
    for (int i = 0; i < n; i++) ptrs[i] = ctx->lists[i].bitmap;
    PSK_TRY(dev_reserve(ctx, ctx->flags, (size_t)n * 8));
    PSK_TRY(dev_reserve(ctx, ctx->keysA, nw * 8));
    PSK_TRY(dev_reserve(ctx, ctx->starts, (size_t)n_chunks * 4));
    PSK_TRY(dev_reserve(ctx, ctx->misc, 64));
    PSK_HIP(ctx, hipMemcpyAsync(ctx->flags.p, ptrs.data(), (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    const uint64_t *const *d_ptrs = reinterpret_cast<const uint64_t *const *>(ctx->flags.p);
    uint64_t *occ = ctx->keysA.as<uint64_t>();
    uint32_t *chunk_rows = ctx->starts.as<uint32_t>();
    uint32_t *d_m = ctx->misc.as<uint32_t>() + 2;
    pd_or_kernel<<<div_up(nw, 256), 256, 0, ctx->stream>>>(d_ptrs, n, nw, occ, chunk_rows);
    PSK_HIP(ctx, hipGetLastError());
    PSK_TRY(dev_exclusive_scan_u32(ctx, chunk_rows, chunk_rows, n_chunks, d_m));
    uint32_t m32 = 0;
    PSK_HIP(ctx, hipMemcpyAsync(&m32, d_m, 4, hipMemcpyDeviceToHost, ctx->stream));
    PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));   // also: `ptrs` has been uploaded
    const uint64_t M = m32;
    if (M) {
        PSK_TRY(dev_reserve(ctx, ctx->union_words, M * 8));
        PSK_TRY(dev_reserve(ctx, ctx->bits, M * (uint64_t)ctx->wpr * 8));
        pd_transpose_kernel<<<n_chunks, PD_THREADS, 0, ctx->stream>>>(d_ptrs, n, ctx->wpr, occ, chunk_rows,
                                                                      (uint64_t)ctx->dense_b0 << DC_VB,
                                                                      ctx->union_words.as<uint64_t>(), ctx->bits.as<uint64_t>());
        PSK_HIP(ctx, hipGetLastError());
        PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    *n_kmers = M;
    *done = 1;
    return PSK_OK;
}
