// f2: Mash-compatible MinHash sketch of one sample (replaces `mash sketch -r <addr>`,
// Samples.get_mash_sketches, modeling.py:386-390; bundled binary bin/mash 2.2).
// Mash 2.2 defaults, established by probing the binary and pinned by tests/golden/mash.json:
// canonical k-mer = alphabetical minimum of the k-mer and its reverse complement, hashed as its
// upper-case ASCII string with MurmurHash3_x64_128 (seed 42); the hash is the first 8 output bytes
// (first 4 when 4^k <= 2^32); the sketch is the `sketch_size` smallest DISTINCT hashes, ascending.
//
//   extract_kernel (kmer_count.hip)  canonical 2-bit words of every window (same tokeniser as a1)
//   kmer_hash_kernel                 word -> ASCII -> MurmurHash3 h1, in place
//   dev_radix_sort_u64 + run heads   ascending distinct hashes; the first `sketch_size` go to the host
#include "dev_utils.h"
#include "psk_internal.h"

namespace {

__device__ __forceinline__ uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

__device__ __forceinline__ uint64_t fmix64(uint64_t k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return k;
}

// MurmurHash3_x64_128 (public-domain algorithm by Austin Appleby) of the k-byte ASCII spelling of a
// 2-bit word; returns h1 (the first 8 bytes of the 128-bit digest).
__device__ __forceinline__ uint64_t murmur3_kmer(uint64_t word, int k, uint64_t seed)
{
    const uint64_t c1 = 0x87c37b91114253d5ull, c2 = 0x4cf5ad432745937full;
    // spell the k-mer: byte j = "ACGT"[code j], packed little-endian into four u64 lanes
    uint64_t lane[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 32; j++) {
        if (j < k) {
            const uint32_t code = (uint32_t)(word >> (2 * (k - 1 - j))) & 3u;
            const uint64_t ch = (0x54474341u >> (8 * code)) & 0xffu;  // 'A','C','G','T'
            lane[j >> 3] |= ch << (8 * (j & 7));
        }
    }
    uint64_t h1 = seed, h2 = seed;
    const int nblocks = k / 16;
    for (int b = 0; b < nblocks; b++) {
        uint64_t k1 = lane[2 * b], k2 = lane[2 * b + 1];
        k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
        h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729ull;
        k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
        h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5ull;
    }
    const int tail = k & 15;
    if (tail > 8) {
        uint64_t k2 = lane[2 * nblocks + 1];
        k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
    }
    if (tail > 0) {
        uint64_t k1 = lane[2 * nblocks];
        k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
    }
    h1 ^= (uint64_t)k; h2 ^= (uint64_t)k;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    h1 += h2;
    return h1;
}

__global__ void kmer_hash_kernel(uint64_t *__restrict__ words, uint64_t n, int k, uint64_t seed, uint64_t mask)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) words[i] = murmur3_kmer(words[i], k, seed) & mask;
}

// hash every word and keep (append, unordered) only the hashes <= limit: the bottom-s sketch of n uniform
// hashes lies below ~ s/n of the hash space, so a limit at 8 s/n keeps ~8 s candidates instead of sorting all n
// (n_dev: the word count when only the device knows it, n then bounds the launch; cap: entries `out` holds -- the
// count keeps running beyond it, so the reader sees the overflow)
__global__ void kmer_hash_filter_kernel(const uint64_t *__restrict__ words, uint64_t n, const uint32_t *__restrict__ n_dev,
                                        int k, uint64_t seed, uint64_t mask, uint64_t limit, uint64_t *__restrict__ out,
                                        uint32_t cap, uint32_t *__restrict__ n_out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n_dev) n = *n_dev;
    uint64_t h = 0;
    bool keep = false;
    if (i < n) {
        h = murmur3_kmer(words[i], k, seed) & mask;
        keep = h <= limit;
    }
    const uint64_t bal = __ballot(keep);
    if (!bal) return;
    const int lane = threadIdx.x & 63;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(n_out, (uint32_t)__popcll(bal));
    base = __shfl(base, 0, 64);
    const uint32_t at = base + (uint32_t)__popcll(bal & psk_lanemask_lt(lane));
    if (keep && at < cap) out[at] = h;
}

// One workgroup: sorts the (few thousand) candidate hashes in LDS and writes the `s` smallest distinct ones.
// res[0] = number of distinct candidates, or ~0 when the candidates do not fit / are fewer than s (the caller then
// takes the general route); res[1 .. s] = the sketch.
constexpr int SK_CAND_CAP = 8192;
constexpr int SK_THREADS = 1024;
__global__ __launch_bounds__(SK_THREADS) void sketch_select_kernel(const uint64_t *__restrict__ cand,
                                                                    const uint32_t *__restrict__ n_cand, uint32_t s,
                                                                    uint64_t *__restrict__ res)
{
    __shared__ uint64_t a[SK_CAND_CAP];
    __shared__ uint32_t wave_tot[SK_THREADS / 64];
    const uint32_t c = *n_cand;
    const int tid = threadIdx.x;
    if (c > (uint32_t)SK_CAND_CAP || c < s) {
        if (tid == 0) res[0] = ~0ull;
        return;
    }
    uint32_t P = SK_THREADS;  // power of two >= c, at least one element per thread
    while (P < c) P <<= 1;
    for (uint32_t i = tid; i < P; i += SK_THREADS) a[i] = i < c ? cand[i] : ~0ull;  // pads sort to the end
    __syncthreads();
    for (uint32_t len = 2; len <= P; len <<= 1)
        for (uint32_t inc = len >> 1; inc > 0; inc >>= 1) {
            for (uint32_t t = tid; t < P / 2; t += SK_THREADS) {
                const uint32_t lo = ((t & ~(inc - 1)) << 1) | (t & (inc - 1)), hi = lo | inc;
                const bool up = (lo & len) == 0;
                const uint64_t x = a[lo], y = a[hi];
                if ((x > y) == up) { a[lo] = y; a[hi] = x; }
            }
            __syncthreads();
        }
    // distinct values in order: thread t owns the consecutive elements [t * per, (t + 1) * per)
    const uint32_t per = P / SK_THREADS, i0 = tid * per;
    uint32_t heads = 0;
    for (uint32_t i = i0; i < i0 + per; i++) heads += (i < c && (i == 0 || a[i] != a[i - 1])) ? 1u : 0u;
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t incl = psk_wave_incl_scan_u32(heads, lane);
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t before = incl - heads, total = 0;
    for (int w = 0; w < SK_THREADS / 64; w++) {
        if (w < wave) before += wave_tot[w];
        total += wave_tot[w];
    }
    for (uint32_t i = i0; i < i0 + per; i++)
        if (i < c && (i == 0 || a[i] != a[i - 1])) {
            if (before < s) res[1 + before] = a[i];
            before++;
        }
    if (tid == 0) res[0] = total;
}

__global__ void head_flags_kernel(const uint64_t *__restrict__ keys, uint64_t n, uint32_t *__restrict__ flags)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

// heads whose rank is below `limit` are written to out[rank]
__global__ void take_first_heads_kernel(const uint64_t *__restrict__ keys, uint64_t n, const uint32_t *__restrict__ pos,
                                        uint32_t limit, uint64_t *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool head = (i == 0) || keys[i] != keys[i - 1];
    if (head && pos[i] < limit) out[pos[i]] = keys[i];
}

}  // namespace

int upload_clean_stream(psk_ctx *ctx, const uint8_t *bytes, size_t len, uint64_t *clean_len);  // kmer_count.hip

// sketch of a clean stream that already sits in device memory (stream-ordered after whatever produced it)
int sketch_from_device(psk_ctx *ctx, const uint8_t *d_clean, uint64_t clean_len, int k, int sketch_size, uint32_t seed,
                       uint64_t *hashes_out, uint64_t *n_out)
{
    *n_out = 0;
    if (clean_len == 0) return PSK_OK;
    if (clean_len >= (1ull << 32)) return psk_fail(ctx, PSK_ERANGE, "sample larger than 4 Gbases");
    PSK_TRY(dev_reserve(ctx, ctx->keysA, clean_len * 8));
    PSK_TRY(dev_reserve(ctx, ctx->keysB, clean_len * 8));
    PSK_TRY(dev_reserve(ctx, ctx->misc, 64));
    uint32_t *d_n = ctx->misc.as<uint32_t>() + 8;
    PSK_HIP(ctx, hipMemsetAsync(d_n, 0, 16, ctx->stream));
    PSK_TRY(launch_extract(ctx, d_clean, clean_len, k, 0, 0, ctx->keysA.as<uint64_t>(), d_n));
    uint32_t n32 = 0;
    PSK_HIP(ctx, hipMemcpyAsync(&n32, d_n, 4, hipMemcpyDeviceToHost, ctx->stream));
    PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const uint64_t n = n32;
    if (n == 0) return PSK_OK;
    const bool wide = (k > 16);  // Mash keeps 64-bit hashes only when 4^k exceeds 2^32
    const uint64_t mask = wide ? ~0ull : 0xffffffffull;
    // candidates: hashes below 8 s/n of the hash space (the sketch is among them unless the sample is tiny or
    // very repetitive -- then the count below comes up short and everything is sorted instead)
    const double ratio = 8.0 * (double)sketch_size / (double)n;
    uint64_t cand = 0;
    if (ratio < 0.5) {
        const uint64_t limit = wide ? (uint64_t)(ratio * 18446744073709551616.0) : (uint64_t)(ratio * 4294967296.0);
        PSK_TRY(dev_reserve(ctx, ctx->valsA, n * 8));  // worst case: every hash is a candidate
        kmer_hash_filter_kernel<<<div_up(n, 256), 256, 0, ctx->stream>>>(ctx->keysA.as<uint64_t>(), n, nullptr, k, (uint64_t)seed,
                                                                       mask, limit, ctx->valsA.as<uint64_t>(), (uint32_t)n, d_n + 1);
        PSK_HIP(ctx, hipGetLastError());
        uint32_t c32 = 0;
        PSK_HIP(ctx, hipMemcpyAsync(&c32, d_n + 1, 4, hipMemcpyDeviceToHost, ctx->stream));
        PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        cand = c32;
    }
    uint64_t *keys = ctx->keysA.as<uint64_t>(), *spare = ctx->keysB.as<uint64_t>();
    uint64_t n_sort = n;
    bool filtered = cand >= (uint64_t)sketch_size;
    for (int attempt = 0; attempt < 2; attempt++) {
        if (filtered) {
            // sort the candidates only; keysB is the spare, the unhashed words stay intact in keysA
            keys = ctx->valsA.as<uint64_t>();
            n_sort = cand;
        } else {
            kmer_hash_kernel<<<div_up(n, 256), 256, 0, ctx->stream>>>(ctx->keysA.as<uint64_t>(), n, k, (uint64_t)seed, mask);
            PSK_HIP(ctx, hipGetLastError());
            keys = ctx->keysA.as<uint64_t>();
            n_sort = n;
        }
        uint64_t *sorted = nullptr;
        PSK_TRY(dev_radix_sort_u64(ctx, keys, spare, n_sort, 0, wide ? 64 : 32, &sorted));
        uint64_t *other = (sorted == keys) ? spare : keys;
        PSK_TRY(dev_reserve(ctx, ctx->flags, n_sort * 4));
        uint32_t *flags = ctx->flags.as<uint32_t>();
        head_flags_kernel<<<div_up(n_sort, 256), 256, 0, ctx->stream>>>(sorted, n_sort, flags);
        PSK_HIP(ctx, hipGetLastError());
        PSK_TRY(dev_exclusive_scan_u32(ctx, flags, flags, n_sort, d_n + 2));
        take_first_heads_kernel<<<div_up(n_sort, 256), 256, 0, ctx->stream>>>(sorted, n_sort, flags, (uint32_t)sketch_size, other);
        PSK_HIP(ctx, hipGetLastError());
        uint32_t nu = 0;
        PSK_HIP(ctx, hipMemcpyAsync(&nu, d_n + 2, 4, hipMemcpyDeviceToHost, ctx->stream));
        PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (filtered && nu < (uint32_t)sketch_size) {  // too many repeats among the candidates: sort everything
            filtered = false;
            continue;
        }
        const uint64_t take = nu < (uint32_t)sketch_size ? nu : (uint32_t)sketch_size;
        if (take) PSK_HIP(ctx, hipMemcpy(hashes_out, other, take * 8, hipMemcpyDeviceToHost));
        *n_out = take;
        return PSK_OK;
    }
    return psk_fail(ctx, PSK_ESTATE, "sketch: unreachable");
}

// The batch counter's form of the same sketch: queued on the context's stream behind the counting chain of the
// sample and NOT waited for.  extract (sketch k) -> hash + filter at 4 s / clean_len of the hash space (~4 s
// candidates) -> one-workgroup LDS sort + distinct select -> result to pinned memory.  sketch_collect reads it one
// sample later and falls back to sketch_from_device (tiny or very repetitive samples, sketch sizes whose candidates
// do not fit the LDS sort) while the sample's clean stream is still in its lane.
int sketch_enqueue(psk_ctx *ctx, CountLane &L, const uint8_t *d_clean, uint64_t clean_len, int k, int sketch_size, uint32_t seed)
{
    L.sk_state = 2;
    if (clean_len == 0) return PSK_OK;
    if (clean_len >= (1ull << 32)) return psk_fail(ctx, PSK_ERANGE, "sample larger than 4 Gbases");
    const double ratio = 4.0 * (double)sketch_size / (double)clean_len;
    if (!(ratio < 0.5) || 4ull * (uint64_t)sketch_size + 600 > (uint64_t)SK_CAND_CAP) return PSK_OK;  // general route
    const bool wide = (k > 16);
    const uint64_t mask = wide ? ~0ull : 0xffffffffull;
    const uint64_t limit = wide ? (uint64_t)(ratio * 18446744073709551616.0) : (uint64_t)(ratio * 4294967296.0);
    const size_t res_bytes = (1 + (size_t)sketch_size) * 8;
    PSK_TRY(dev_reserve(ctx, ctx->keysA, clean_len * 8));
    PSK_TRY(dev_reserve(ctx, L.sk_cand, 16 + (size_t)SK_CAND_CAP * 8));
    PSK_TRY(dev_reserve(ctx, L.sk_out, res_bytes));
    if (res_bytes > L.sk_host_cap) {
        if (L.sk_host) (void)hipHostFree(L.sk_host);
        L.sk_host = nullptr;
        L.sk_host_cap = 0;
        PSK_HIP(ctx, hipHostMalloc(reinterpret_cast<void **>(&L.sk_host), res_bytes, hipHostMallocDefault));
        L.sk_host_cap = res_bytes;
    }
    if (!L.sk_done) {
        PSK_HIP(ctx, hipEventCreateWithFlags(&L.sk_done, hipEventDisableTiming));
        PSK_HIP(ctx, hipEventCreateWithFlags(&L.sk_filtered, hipEventDisableTiming));
    }
    if (!ctx->sketch_stream) PSK_HIP(ctx, hipStreamCreateWithFlags(&ctx->sketch_stream, hipStreamNonBlocking));
    uint32_t *d_n = L.sk_cand.as<uint32_t>();
    uint64_t *cand = L.sk_cand.as<uint64_t>() + 2;
    PSK_HIP(ctx, hipMemsetAsync(d_n, 0, 16, ctx->stream));
    PSK_TRY(launch_extract(ctx, d_clean, clean_len, k, 0, 0, ctx->keysA.as<uint64_t>(), d_n));
    kmer_hash_filter_kernel<<<div_up(clean_len, 256), 256, 0, ctx->stream>>>(ctx->keysA.as<uint64_t>(), clean_len, d_n, k,
                                                                           (uint64_t)seed, mask, limit, cand, (uint32_t)SK_CAND_CAP,
                                                                           d_n + 1);
    PSK_HIP(ctx, hipGetLastError());
    // the select is ONE workgroup sorting a few thousand hashes: on its own stream it runs beside the next sample's
    // counting chain instead of holding the whole GPU for its ~0.1 ms (this lane's candidate buffer is not written
    // again before sketch_collect has waited for sk_done)
    PSK_HIP(ctx, hipEventRecord(L.sk_filtered, ctx->stream));
    PSK_HIP(ctx, hipStreamWaitEvent(ctx->sketch_stream, L.sk_filtered, 0));
    sketch_select_kernel<<<1, SK_THREADS, 0, ctx->sketch_stream>>>(cand, d_n + 1, (uint32_t)sketch_size, L.sk_out.as<uint64_t>());
    PSK_HIP(ctx, hipGetLastError());
    PSK_HIP(ctx, hipMemcpyAsync(L.sk_host, L.sk_out.p, res_bytes, hipMemcpyDeviceToHost, ctx->sketch_stream));
    PSK_HIP(ctx, hipEventRecord(L.sk_done, ctx->sketch_stream));
    L.sk_state = 1;
    return PSK_OK;
}

int sketch_collect(psk_ctx *ctx, CountLane &L, const uint8_t *d_clean, uint64_t clean_len, int k, int sketch_size,
                   uint32_t seed, uint64_t *hashes_out, uint64_t *n_out)
{
    const int state = L.sk_state;
    L.sk_state = 0;
    *n_out = 0;
    if (state == 0) return PSK_OK;
    if (state == 1) {
        PSK_HIP(ctx, hipEventSynchronize(L.sk_done));
        const uint64_t distinct = L.sk_host[0];
        if (distinct != ~0ull && distinct >= (uint64_t)sketch_size) {
            memcpy(hashes_out, L.sk_host + 1, (size_t)sketch_size * 8);
            *n_out = (uint64_t)sketch_size;
            return PSK_OK;
        }
    }
    return sketch_from_device(ctx, d_clean, clean_len, k, sketch_size, seed, hashes_out, n_out);
}

extern "C" int psk_minhash_sketch(psk_ctx *ctx, const uint8_t *bytes, size_t len, int k, int sketch_size, uint32_t seed,
                                  uint64_t *hashes_out, uint64_t *n_out)
{
    if (!ctx) return PSK_EINVAL;
    if (k < 1 || k > 32) return psk_fail(ctx, PSK_EINVAL, "k must be 1..32");
    if (sketch_size < 1) return psk_fail(ctx, PSK_EINVAL, "sketch_size must be >= 1");
    if (!hashes_out || !n_out || (!bytes && len)) return psk_fail(ctx, PSK_EINVAL, "null buffer");
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    uint64_t clean_len = 0;
    PSK_TRY(upload_clean_stream(ctx, bytes, len, &clean_len));
    return sketch_from_device(ctx, ctx->raw.as<uint8_t>(), clean_len, k, sketch_size, seed, hashes_out, n_out);
}

// ---- pairwise sketch comparison (was `mash dist reference.msh reference.msh`, modeling.py:411-421) ----------
// One thread per pair (r >= c): the merge of Mash's distance estimate -- walk both ascending sketches until
// `s` distinct hashes of the union have been seen, count the shared ones; when a sketch runs out first the
// denominator grows by what is left of the other, capped at s.  N = 1024 samples are 524,800 pairs of <= 2000
// steps over 8 MB of sketches (L2-resident): milliseconds, against minutes for the same loops on the host.
namespace {
__global__ void mash_pairs_kernel(const uint64_t *__restrict__ sk, const uint32_t *__restrict__ lens, int n, int s,
                                  uint32_t *__restrict__ common_out, uint32_t *__restrict__ denom_out)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t n_pairs = (uint64_t)n * (n + 1) / 2;
    if (t >= n_pairs) return;
    // t -> (r, c) with c <= r, rows of the lower triangle in order
    uint32_t r = (uint32_t)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((uint64_t)r * (r + 1) / 2 > t) r--;
    while ((uint64_t)(r + 1) * (r + 2) / 2 <= t) r++;
    const uint32_t c = (uint32_t)(t - (uint64_t)r * (r + 1) / 2);
    const uint64_t *a = sk + (uint64_t)r * s, *b = sk + (uint64_t)c * s;
    const uint32_t na = lens[r], nb = lens[c];
    uint32_t i = 0, j = 0, common = 0, denom = 0;
    while (denom < (uint32_t)s && i < na && j < nb) {
        const uint64_t x = a[i], y = b[j];
        if (x < y) i++;
        else if (x > y) j++;
        else { i++; j++; common++; }
        denom++;
    }
    if (denom < (uint32_t)s) {
        if (i < na) denom += na - i;
        if (j < nb) denom += nb - j;
        if (denom > (uint32_t)s) denom = (uint32_t)s;
    }
    common_out[(uint64_t)r * n + c] = common;
    denom_out[(uint64_t)r * n + c] = denom;
    common_out[(uint64_t)c * n + r] = common;
    denom_out[(uint64_t)c * n + r] = denom;
}
}  // namespace

extern "C" int psk_mash_pairs(psk_ctx *ctx, const uint64_t *sketches, const uint32_t *lens, int n, int sketch_size,
                              uint32_t *common_out, uint32_t *denom_out)
{
    if (!ctx) return PSK_EINVAL;
    if (!sketches || !lens || !common_out || !denom_out) return psk_fail(ctx, PSK_EINVAL, "null buffer");
    if (n < 1 || sketch_size < 1) return psk_fail(ctx, PSK_EINVAL, "bad shape n=%d s=%d", n, sketch_size);
    for (int i = 0; i < n; i++)
        if (lens[i] > (uint32_t)sketch_size) return psk_fail(ctx, PSK_EINVAL, "sketch %d longer than sketch_size", i);
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t sk_bytes = (size_t)n * sketch_size * 8, nn = (size_t)n * n;
    PSK_TRY(dev_reserve(ctx, ctx->keysA, sk_bytes + (size_t)n * 4));
    PSK_TRY(dev_reserve(ctx, ctx->keysB, nn * 8));
    uint64_t *d_sk = ctx->keysA.as<uint64_t>();
    uint32_t *d_len = reinterpret_cast<uint32_t *>(ctx->keysA.as<uint8_t>() + sk_bytes);
    uint32_t *d_common = ctx->keysB.as<uint32_t>(), *d_denom = d_common + nn;
    PSK_HIP(ctx, hipMemcpyAsync(d_sk, sketches, sk_bytes, hipMemcpyHostToDevice, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(d_len, lens, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    const uint64_t n_pairs = (uint64_t)n * (n + 1) / 2;
    mash_pairs_kernel<<<div_up(n_pairs, 256), 256, 0, ctx->stream>>>(d_sk, d_len, n, sketch_size, d_common, d_denom);
    PSK_HIP(ctx, hipGetLastError());
    PSK_HIP(ctx, hipMemcpyAsync(common_out, d_common, nn * 4, hipMemcpyDeviceToHost, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(denom_out, d_denom, nn * 4, hipMemcpyDeviceToHost, ctx->stream));
    PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PSK_OK;
}
