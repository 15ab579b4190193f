// stream_probe.hip -- the measured stream-read ceiling beside the scan's roofline fraction.
//
// SURVEY.md section 8(d) asks for the scan's achieved bandwidth "against both the 8 TB/s spec and the measured stream-read
// ceiling".  The ceiling is measured where the scan runs: the presence matrix itself -- the same bytes, the same residency
// in the 256-MiB Infinity Cache as the scan sees between back-to-back launches -- is read once per launch by a kernel that
// does nothing else (16 B per lane, four independent loads in flight per lane, XOR-folded so that the loads cannot be
// dropped), timed with HIP events on the context's stream like the scan.  No reference counterpart: measurement only.
#include "psk_internal.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Shape 0: grid-stride, four loads 1/4 of the grid's span apart per lane.
// Shape 1: the scan's own shape -- a wave owns UNR consecutive 1-KiB pieces per step (64 lanes x 16 B each), the waves of the
// grid take consecutive steps and stride by the whole grid (assoc_scan.hip chi2_scan_kernel: SC_UNROLL = 4 pieces in flight).
// NT: non-temporal loads (no allocation in L2 / the Infinity Cache) or plain ones.
template <int SHAPE, bool NT>
__global__ __launch_bounds__(256) void stream_read_kernel(const u32x4 *__restrict__ p, uint64_t n_vec, uint32_t *__restrict__ sink, const uint32_t magic)
{
    auto ld = [](const u32x4 *q) { return NT ? __builtin_nontemporal_load(q) : *q; };
    u32x4 acc = {0u, 0u, 0u, 0u};
    if (SHAPE == 0) {
        const uint64_t stride = (uint64_t)gridDim.x * 256;
        uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
        for (; i + 3 * stride < n_vec; i += 4 * stride) {
            const u32x4 a = ld(p + i), b = ld(p + i + stride), c = ld(p + i + 2 * stride), d = ld(p + i + 3 * stride);
            acc ^= a ^ b ^ c ^ d;
        }
        for (; i < n_vec; i += stride) acc ^= ld(p + i);
    } else {
        constexpr int UNR = 4;
        const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (uint64_t)gridDim.x * 4;
        const uint32_t lane = threadIdx.x & 63;
        for (uint64_t step = wave; step * (UNR * 64) < n_vec; step += n_waves) {
            const uint64_t i0 = step * (UNR * 64) + lane;
            u32x4 v[UNR];
#pragma unroll
            for (int u = 0; u < UNR; u++) v[u] = (i0 + (uint64_t)u * 64 < n_vec) ? ld(p + i0 + (uint64_t)u * 64) : (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
            for (int u = 0; u < UNR; u++) acc ^= v[u];
        }
    }
    // the fold is compared with a run-time argument (a comparison the compiler can neither predict nor hoist: r05's first cut tested
    // a loop-invariant instead and the compiler dropped the whole loop of one shape -- a 123 TB/s "ceiling"); a match, one in 2^32,
    // would rewrite one word of the matrix with itself
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == magic) *sink = p[0][0];
}

template <int SHAPE, bool NT>
int time_shape(psk_ctx *ctx, unsigned blocks, uint64_t n_vec, int reps, double *mean_ms)
{
    double total = 0;
    for (int r = 0; r < reps; r++) {
        PSK_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
        stream_read_kernel<SHAPE, NT><<<blocks, 256, 0, ctx->stream>>>(reinterpret_cast<const u32x4 *>(ctx->bits.p), n_vec,
                                                                       ctx->bits.as<uint32_t>(), 0x9e3779b9u ^ (uint32_t)r);
        PSK_HIP(ctx, hipGetLastError());
        PSK_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
        PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        float ms = 0;
        PSK_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        total += ms;
    }
    *mean_ms = total / reps;
    return PSK_OK;
}

}  // namespace

extern "C" int psk_stream_read_ceiling(psk_ctx *ctx, int reps, double *mean_ms, uint64_t *bytes_per_launch, int *shape_out)
{
    if (!ctx) return PSK_EINVAL;
    if (ctx->n_in_flight) return psk_fail(ctx, PSK_ESTATE, "a scan is in flight (psk_scan_end first)");
    if (!ctx->have_presence) return psk_fail(ctx, PSK_ESTATE, "no presence matrix (psk_build_presence first)");
    if (reps < 1) return psk_fail(ctx, PSK_EINVAL, "reps must be >= 1");
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t n_vec = ctx->n_kmers * 8ull * (uint64_t)ctx->wpr / 16;
    const unsigned cu = (unsigned)(ctx->n_cu > 0 ? ctx->n_cu : 256);
    // a ceiling is the BEST plain read there is: four shapes are timed, the fastest is reported (and named)
    double ms[4] = {0, 0, 0, 0};
    PSK_TRY((time_shape<0, true>(ctx, cu * 8, n_vec, reps, &ms[0])));
    PSK_TRY((time_shape<0, false>(ctx, cu * 8, n_vec, reps, &ms[1])));
    PSK_TRY((time_shape<1, true>(ctx, cu * 16, n_vec, reps, &ms[2])));
    PSK_TRY((time_shape<1, false>(ctx, cu * 16, n_vec, reps, &ms[3])));
    int best = 0;
    for (int i = 1; i < 4; i++) if (ms[i] < ms[best]) best = i;
    if (mean_ms) *mean_ms = ms[best];
    if (bytes_per_launch) *bytes_per_launch = n_vec * 16;
    if (shape_out) *shape_out = best;
    return PSK_OK;
}
