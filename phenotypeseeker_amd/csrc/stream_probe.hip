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

__global__ __launch_bounds__(256) void stream_read_kernel(const u32x4 *__restrict__ p, uint64_t n_vec, uint32_t *__restrict__ sink)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (; i + 3 * stride < n_vec; i += 4 * stride) {
        const u32x4 a = __builtin_nontemporal_load(p + i), b = __builtin_nontemporal_load(p + i + stride);
        const u32x4 c = __builtin_nontemporal_load(p + i + 2 * stride), d = __builtin_nontemporal_load(p + i + 3 * stride);
        acc ^= a ^ b ^ c ^ d;
    }
    for (; i < n_vec; i += stride) acc ^= __builtin_nontemporal_load(p + i);
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x5bd1e995u && n_vec == ~0ull) *sink = acc[0];   // (never true: keeps the loads)
}

}  // namespace

extern "C" int psk_stream_read_ceiling(psk_ctx *ctx, int reps, double *mean_ms, uint64_t *bytes_per_launch)
{
    if (!ctx) return PSK_EINVAL;
    if (ctx->n_in_flight) return psk_fail(ctx, PSK_ESTATE, "a scan is in flight (psk_scan_end first)");
    if (!ctx->have_presence) return psk_fail(ctx, PSK_ESTATE, "no presence matrix (psk_build_presence first)");
    if (reps < 1) return psk_fail(ctx, PSK_EINVAL, "reps must be >= 1");
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t n_vec = ctx->n_kmers * 8ull * (uint64_t)ctx->wpr / 16;
    const unsigned blocks = (unsigned)((ctx->n_cu > 0 ? ctx->n_cu : 256) * 8);
    double total = 0;
    for (int r = 0; r < reps; r++) {
        PSK_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
        stream_read_kernel<<<blocks, 256, 0, ctx->stream>>>(reinterpret_cast<const u32x4 *>(ctx->bits.p), n_vec,
                                                            ctx->bits.as<uint32_t>() /* the sink that is never written */);
        PSK_HIP(ctx, hipGetLastError());
        PSK_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
        PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        float ms = 0;
        PSK_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        total += ms;
    }
    if (mean_ms) *mean_ms = total / reps;
    if (bytes_per_launch) *bytes_per_launch = n_vec * 16;
    return PSK_OK;
}
