// a2+a3: union of the per-sample lists and the bit-packed k-mer x sample presence matrix.
// Replaces the glistcompare -u tree (modeling.py:350-380) and the glistquery -l / split text
// mapping (modeling.py:317-348): every (word, sample) pair of the run is packed into one u64
// (word << sbits | sample), the pairs are radix-sorted on the word bits, run heads number the
// rows, and each pair sets its sample bit in its row.  Rows come out in ascending word order,
// i.e. in glistmaker/glistcompare list order.
#include "dev_utils.h"
#include "psk_internal.h"

#include <chrono>

namespace {

// PSK_TRACE=1 prints host-side phase timings of psk_build_presence to stderr
struct PhaseTimer {
    bool on;
    hipStream_t st;
    std::chrono::steady_clock::time_point t0;
    explicit PhaseTimer(hipStream_t s) : on(getenv("PSK_TRACE") != nullptr), st(s), t0(std::chrono::steady_clock::now()) {}
    void mark(const char *what)
    {
        if (!on) return;
        (void)hipStreamSynchronize(st);
        auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[psk] build_presence %-14s %8.3f ms\n", what,
                std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

// vals == nullptr: pair = word << sbits | sample in one u64; else key = word, payload = sample;
// every sample of a chunk in ONE launch (a launch per sample and chunk: 12,288 of them for config 3's 2,048
// samples, a quarter of the build's GPU time and as much again in launch overhead): block (s, b) packs slice b of sample s
struct PackRef {
    const uint64_t *words;   // first word of the sample's range in this chunk
    uint64_t count, offset;  // its length and where its pairs go
};
constexpr int PACK_SLICES = 8;
__global__ __launch_bounds__(256) void pack_pairs_batch_kernel(const PackRef *__restrict__ refs, int sbits, uint64_t *__restrict__ out,
                                                               uint32_t *__restrict__ vals)
{
    const int s = blockIdx.x / PACK_SLICES, b = blockIdx.x % PACK_SLICES;
    const PackRef r = refs[s];
    const uint64_t per = (r.count + PACK_SLICES - 1) / PACK_SLICES, i0 = (uint64_t)b * per;
    const uint64_t i1 = i0 + per < r.count ? i0 + per : r.count;
    for (uint64_t i = i0 + threadIdx.x; i < i1; i += 256) {
        if (vals) { out[r.offset + i] = r.words[i]; vals[r.offset + i] = (uint32_t)s; }
        else out[r.offset + i] = (r.words[i] << sbits) | (uint64_t)s;
    }
}

__global__ void pair_head_flags_kernel(const uint64_t *__restrict__ pairs, uint64_t n, int sbits,
                                       uint32_t *__restrict__ flags)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flags[i] = (i == 0 || (pairs[i] >> sbits) != (pairs[i - 1] >> sbits)) ? 1u : 0u;
}

// rowpos = exclusive scan of head flags; row of pair i = rowpos[i] + head(i) - 1.
// Pairs are sorted by word and, inside one word, by sample (the sort is stable and the samples were
// appended in order), so the matrix word a pair touches is non-decreasing along the array: each wave
// OR-combines runs of equal target words with a segmented shuffle scan and issues one atomicOr per run
// (r01 issued one per pair: 35.6 GB of memory-side atomic traffic for a 0.7 GB matrix).
__global__ __launch_bounds__(256) void presence_fill_kernel(const uint64_t *__restrict__ pairs, uint64_t n, int sbits,
                                                             const uint32_t *__restrict__ rowpos, int wpr,
                                                             uint64_t *__restrict__ words,
                                                             unsigned long long *__restrict__ bits,
                                                             const uint32_t *__restrict__ vals)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    unsigned long long key = ~0ull, val = 0ull;  // key = index of the target matrix word
    if (i < n) {
        const uint64_t pr = pairs[i];
        const uint64_t w = pr >> sbits;
        const bool head = (i == 0) || (pairs[i - 1] >> sbits) != w;
        const uint64_t row = (uint64_t)rowpos[i] + (head ? 1u : 0u) - 1u;
        const uint32_t s = vals ? vals[i] : (uint32_t)(pr & ((1ull << sbits) - 1ull));
        if (head) words[row] = w;
        key = row * (uint64_t)wpr + (s >> 6);
        val = 1ull << (s & 63);
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long k2 = __shfl_up(key, d, 64);
        const unsigned long long v2 = __shfl_up(val, d, 64);
        if (lane >= d && k2 == key) val |= v2;
    }
    const unsigned long long knext = __shfl_down(key, 1, 64);
    const bool last = (lane == 63) || (knext != key);
    if (i < n && last) atomicOr(&bits[key], val);
}

__global__ void gather_rows_kernel(const uint64_t *__restrict__ bits, int wpr, const uint64_t *__restrict__ idx,
                                   uint64_t n, uint64_t *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * (uint64_t)wpr) return;
    const uint64_t r = i / wpr, c = i % wpr;
    out[i] = bits[idx[r] * (uint64_t)wpr + c];
}

// keep[r] = 1 if words[r] occurs in the sorted db list
__global__ void db_member_kernel(const uint64_t *__restrict__ words, uint64_t m, const uint64_t *__restrict__ db,
                                 uint64_t ndb, uint32_t *__restrict__ keep)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m) return;
    const uint64_t key = words[r];
    uint64_t lo = 0, hi = ndb;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (db[mid] < key) lo = mid + 1; else hi = mid;
    }
    keep[r] = (lo < ndb && db[lo] == key) ? 1u : 0u;
}

__global__ void compact_rows_kernel(const uint64_t *__restrict__ words, const uint64_t *__restrict__ bits, int wpr,
                                    uint64_t m, const uint32_t *__restrict__ pos, uint64_t *__restrict__ words_out,
                                    uint64_t *__restrict__ bits_out)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m) return;
    if (pos[r + 1] == pos[r]) return;  // pos has m + 1 entries (caller appends a sentinel)
    const uint64_t j = pos[r];
    words_out[j] = words[r];
    for (int c = 0; c < wpr; c++) bits_out[j * (uint64_t)wpr + c] = bits[r * (uint64_t)wpr + c];
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// Benchmark-only synthetic matrix.  Row classes mimic a real union (DESIGN.md "Synthetic matrix"):
//   ~45 % of rows: k-mer of a single sample (mutation-born singletons)
//   ~35 % of rows: present in (almost) every sample (ancestral)
//   ~19 % of rows: random subset with a row-specific density
//   ~ 1 % of rows: "gene" k-mers: 90 % of the even samples, 10 % of the odd ones (bench.py and the
//                  tests use phenotype = 1 for even samples, so these rows survive the filter)
__global__ void synth_presence_kernel(uint64_t *__restrict__ bits, uint64_t m, int wpr, int n_samples, uint64_t seed)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m * (uint64_t)wpr) return;
    const uint64_t r = i / wpr;
    const int c = (int)(i % wpr);
    const uint64_t hr = splitmix64(seed ^ (r * 0xD1342543DE82EF95ull));
    const uint32_t cls = (uint32_t)(hr % 100u);
    uint64_t word = 0;
    const int base = c * 64;
    if (cls < 45) {
        const int s = (int)((hr >> 20) % (uint64_t)n_samples);
        if (s >= base && s < base + 64) word = 1ull << (s - base);
    } else if (cls < 80) {
        word = ~0ull;
        const int s = (int)((hr >> 20) % (uint64_t)n_samples);
        if (((hr >> 50) & 3) == 0 && s >= base && s < base + 64) word &= ~(1ull << (s - base));
    } else {
        // seed bits 48..63 (benchmark knob): keep only that many in 10,000 of the gene rows (0: all of them, 1 % of the matrix);
        // the others become random-density rows.  80 gives config 2's share of survivors (0.008 %).
        const uint32_t keep = (uint32_t)(seed >> 48);
        const bool gene = cls == 99 && (keep == 0 || (uint32_t)((hr >> 33) % 10000u) < keep);
        const uint32_t dens = (uint32_t)((hr >> 24) & 0xff);  // per-row density /256
        for (int b = 0; b < 64; b += 8) {
            uint64_t h = splitmix64(hr ^ ((uint64_t)(c * 8 + (b >> 3)) * 0x2545F4914F6CDD1Dull));
            for (int q = 0; q < 8; q++) {
                const uint32_t thr = gene ? (((b + q) & 1) ? 26u : 230u) : dens;
                if (((h >> (8 * q)) & 0xff) < thr) word |= 1ull << (b + q);
            }
        }
    }
    if (base + 64 > n_samples) {
        const int valid = n_samples - base;
        word &= (valid <= 0) ? 0ull : ((valid >= 64) ? ~0ull : ((1ull << valid) - 1ull));
    }
    bits[i] = word;
}

__global__ void iota_words_kernel(uint64_t *__restrict__ words, uint64_t m)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) words[i] = i;
}

int sample_bits(int n_samples)
{
    int b = 1;
    while ((1 << b) < n_samples) b++;
    return b;
}

}  // namespace

int build_presence_tiled(psk_ctx *ctx, uint64_t total_pairs, uint64_t *n_kmers, int *done);  // presence_tiled.hip
int build_presence_merge(psk_ctx *ctx, uint64_t total_pairs, uint64_t *n_kmers, int *done);  // presence_merge.hip
int build_presence_merge_wide(psk_ctx *ctx, uint64_t total_pairs, uint64_t *n_kmers, int *done);  // presence_merge.hip

static int padded_wpr(int n_samples)
{
    int w = (n_samples + 63) / 64;
    if (w == 1) return 1;  // up to 64 samples: 8-byte rows, two per 16-byte load of the scans (assoc_scan.hip, G = 0)
    return (w + 1) & ~1;   // even: rows are 16-byte aligned
}

static int build_presence_impl(psk_ctx *ctx, uint64_t *n_kmers);

extern "C" int psk_build_presence(psk_ctx *ctx, uint64_t *n_kmers)
{
    if (!ctx) return PSK_EINVAL;
    if (ctx->n_in_flight > 0) return psk_fail(ctx, PSK_ESTATE, "a scan is in flight on this matrix: psk_scan_end first");
    if (ctx->k == 0) return psk_fail(ctx, PSK_ESTATE, "psk_begin has not been called");
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    // the ingest is over: the device buffers the .gz inputs were inflated in go back before the matrix is allocated (milliseconds);
    // their host images -- gigabytes of pages: 0.5 s to unmap -- are handed to a helper thread once the build is done with its own
    // allocations (psk_internal.h: gz_release_host)
    gz_release_device(ctx);
    const int rc = build_presence_impl(ctx, n_kmers);
    gz_release_host(ctx, false);
    return rc;
}

static int build_presence_impl(psk_ctx *ctx, uint64_t *n_kmers)
{
    uint64_t total = 0;
    for (int i = 0; i < ctx->n_samples; i++) {
        if (!ctx->lists[i].done) return psk_fail(ctx, PSK_ESTATE, "sample %d has not been counted", i);
        total += ctx->lists[i].n_unique;
    }
    int sbits = sample_bits(ctx->n_samples);
    // word and sample share one u64 when they fit (k <= 26 at 4096 samples); otherwise the sample id
    // travels as a u32 payload of the sort
    const bool kv = 2 * ctx->k + sbits > 64;
    if (kv) sbits = 0;
    ctx->wpr = padded_wpr(ctx->n_samples);
    ctx->n_kmers = 0;
    ctx->have_presence = false;
    if (total == 0) {
        ctx->have_presence = true;
        ctx->dense_hint = -1;
        if (n_kmers) *n_kmers = 0;
        return PSK_OK;
    }
    PhaseTimer pt(ctx->stream);
    {   // dense list form (2k <= 26): the bitmaps are transposed into the matrix (presence_dense.hip)
        uint64_t M = 0;
        int done = 0;
        PSK_TRY(build_presence_dense(ctx, &M, &done));
        if (done) {
            pt.mark("dense build");
            ctx->n_kmers = M;
            ctx->have_presence = true;
            ctx->dense_hint = -1;
            ctx->last.valid = false;
            if (n_kmers) *n_kmers = M;
            return PSK_OK;
        }
        PSK_TRY(dense_materialize(ctx, 0, ctx->n_samples));   // the list routes below read words[]
    }
    {   // small word spaces: the sort-free tiled build (presence_tiled.hip)
        uint64_t M = 0;
        int done = 0;
        PSK_TRY(build_presence_tiled(ctx, total, &M, &done));
        if (done) {
            pt.mark("tiled build");
            ctx->n_kmers = M;
            ctx->have_presence = true;
            ctx->dense_hint = -1;
            ctx->last.valid = false;
            if (n_kmers) *n_kmers = M;
            return PSK_OK;
        }
    }
    {   // word spaces of up to 2^34 values (k <= 17): the streaming merge of the sorted lists (presence_merge.hip)
        uint64_t M = 0;
        int done = 0;
        PSK_TRY(build_presence_merge(ctx, total, &M, &done));
        if (done) {
            pt.mark("merge build");
            ctx->n_kmers = M;
            ctx->have_presence = true;
            ctx->dense_hint = -1;
            ctx->last.valid = false;
            if (n_kmers) *n_kmers = M;
            return PSK_OK;
        }
    }
    {   // wider word spaces (k >= 18): the same streaming merge without a value bitmap -- union and rows from its records (r05)
        uint64_t M = 0;
        int done = 0;
        PSK_TRY(build_presence_merge_wide(ctx, total, &M, &done));
        if (done) {
            pt.mark("wide merge build");
            ctx->n_kmers = M;
            ctx->have_presence = true;
            ctx->dense_hint = -1;
            ctx->last.valid = false;
            if (n_kmers) *n_kmers = M;
            return PSK_OK;
        }
    }
    // The sort route indexes (word, sample) pairs in u32 and holds two 8-byte buffers of them: the word range of the
    // slab is therefore cut into chunks of at most PAIR_CHUNK pairs (cut points at the quantiles of the longest list;
    // a chunk of a sorted list is a contiguous range of it), each sorted on its own.  One chunk: one pass.  Several:
    // a first pass counts the rows of every chunk, the matrix is allocated once, a second pass fills it.
    uint64_t pair_chunk = 1ull << 30;
    if (const char *e = getenv("PSK_PAIR_CHUNK")) { const uint64_t v = strtoull(e, nullptr, 10); if (v >= 1024) pair_chunk = v; }
    if (pair_chunk >= (1ull << 32)) pair_chunk = (1ull << 32) - 1;
    const int ns = ctx->n_samples;
    std::vector<uint64_t> cut;  // cut[c * ns + i]: first entry of sample i's list that belongs to chunk c; n_chunks + 1 rows
    int n_chunks = 1;
    if (total > pair_chunk) {
        int longest = 0;
        for (int i = 1; i < ns; i++) if (ctx->lists[i].n_unique > ctx->lists[longest].n_unique) longest = i;
        const SampleList &P = ctx->lists[longest];
        int want = (int)((total + pair_chunk - 1) / pair_chunk) + 1;
        for (int attempt = 0;; attempt++) {
            if (attempt == 8 || (uint64_t)want > P.n_unique)
                return psk_fail(ctx, PSK_ERANGE, "cannot cut %llu (word, sample) pairs into chunks of %llu",
                                (unsigned long long)total, (unsigned long long)pair_chunk);
            std::vector<uint64_t> bounds;  // ascending, distinct, none 0
            for (int c = 1; c < want; c++) {
                uint64_t w = 0;
                PSK_HIP(ctx, hipMemcpy(&w, P.words + (P.n_unique * (uint64_t)c) / want, 8, hipMemcpyDeviceToHost));
                if (w != 0 && (bounds.empty() || w > bounds.back())) bounds.push_back(w);
            }
            const int nb = (int)bounds.size();
            std::vector<uint64_t> offs((size_t)ns * (nb ? nb : 1));
            if (nb) PSK_TRY(psk_lists_split(ctx, 0, ns, bounds.data(), nb, offs.data()));
            n_chunks = nb + 1;
            cut.assign((size_t)(n_chunks + 1) * ns, 0);
            uint64_t worst = 0;
            for (int c = 0; c <= n_chunks; c++)
                for (int i = 0; i < ns; i++)
                    cut[(size_t)c * ns + i] = c == 0 ? 0 : c == n_chunks ? ctx->lists[i].n_unique : offs[(size_t)i * nb + (c - 1)];
            for (int c = 0; c < n_chunks; c++) {
                uint64_t t = 0;
                for (int i = 0; i < ns; i++) t += cut[(size_t)(c + 1) * ns + i] - cut[(size_t)c * ns + i];
                if (t > worst) worst = t;
            }
            if (worst <= pair_chunk + pair_chunk / 4 && worst < (1ull << 32)) break;
            want *= 2;
        }
    } else {
        cut.assign((size_t)2 * ns, 0);
        for (int i = 0; i < ns; i++) cut[(size_t)ns + i] = ctx->lists[i].n_unique;
    }
    uint64_t cap_pairs = 0;
    std::vector<uint64_t> chunk_total(n_chunks, 0);
    for (int c = 0; c < n_chunks; c++) {
        for (int i = 0; i < ns; i++) chunk_total[c] += cut[(size_t)(c + 1) * ns + i] - cut[(size_t)c * ns + i];
        if (chunk_total[c] > cap_pairs) cap_pairs = chunk_total[c];
    }
    PSK_TRY(dev_reserve(ctx, ctx->keysA, cap_pairs * 8));
    PSK_TRY(dev_reserve(ctx, ctx->keysB, cap_pairs * 8));
    if (kv) {
        PSK_TRY(dev_reserve(ctx, ctx->valsA, cap_pairs * 4));
        PSK_TRY(dev_reserve(ctx, ctx->valsB, cap_pairs * 4));
    }
    PSK_TRY(dev_reserve(ctx, ctx->misc, 64));
    uint32_t *d_m = ctx->misc.as<uint32_t>() + 2;
    pt.mark("alloc pairs");
    // pack + sort + row numbering of chunk c; the sorted pairs, their payloads and the row index of every pair (kept
    // in the sort's spare key buffer: one multi-GB allocation less) stay in the context's buffers
    uint64_t *sorted = nullptr;
    uint32_t *sorted_vals = nullptr, *flags = nullptr;
    auto sort_chunk = [&](int c, uint64_t *rows_out) -> int {
        uint64_t off = 0;
        std::vector<PackRef> refs(ns);
        for (int i = 0; i < ns; i++) {
            const SampleList &L = ctx->lists[i];
            const uint64_t lo = cut[(size_t)c * ns + i], cnt = cut[(size_t)(c + 1) * ns + i] - lo;
            refs[i] = PackRef{cnt ? L.words + lo : nullptr, cnt, off};
            off += cnt;
        }
        PSK_TRY(dev_reserve(ctx, ctx->starts, (size_t)ns * sizeof(PackRef)));
        PSK_HIP(ctx, hipMemcpyAsync(ctx->starts.p, refs.data(), (size_t)ns * sizeof(PackRef), hipMemcpyHostToDevice, ctx->stream));
        PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));   // `refs` is pageable host memory
        pack_pairs_batch_kernel<<<ns * PACK_SLICES, 256, 0, ctx->stream>>>(ctx->starts.as<PackRef>(), sbits, ctx->keysA.as<uint64_t>(),
                                                                         kv ? ctx->valsA.as<uint32_t>() : nullptr);
        PSK_HIP(ctx, hipGetLastError());
        const uint64_t tc = chunk_total[c];
        PSK_TRY(dev_radix_sort_kv(ctx, ctx->keysA.as<uint64_t>(), ctx->keysB.as<uint64_t>(),
                                  kv ? ctx->valsA.as<uint32_t>() : nullptr, kv ? ctx->valsB.as<uint32_t>() : nullptr, tc, sbits,
                                  sbits + 2 * ctx->k, &sorted, &sorted_vals));
        flags = reinterpret_cast<uint32_t *>(sorted == ctx->keysA.as<uint64_t>() ? ctx->keysB.p : ctx->keysA.p);
        pair_head_flags_kernel<<<div_up(tc, 256), 256, 0, ctx->stream>>>(sorted, tc, sbits, flags);
        PSK_HIP(ctx, hipGetLastError());
        PSK_TRY(dev_exclusive_scan_u32(ctx, flags, flags, tc, d_m));
        uint32_t m32 = 0;
        PSK_HIP(ctx, hipMemcpyAsync(&m32, d_m, 4, hipMemcpyDeviceToHost, ctx->stream));
        PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        *rows_out = m32;
        return PSK_OK;
    };
    std::vector<uint64_t> chunk_rows(n_chunks, 0);
    uint64_t M = 0;
    for (int c = 0; c < n_chunks; c++) {
        if (chunk_total[c]) PSK_TRY(sort_chunk(c, &chunk_rows[c]));
        M += chunk_rows[c];
    }
    pt.mark(n_chunks > 1 ? "count rows" : "pack+sort+heads");
    PSK_TRY(dev_reserve(ctx, ctx->union_words, M * 8));
    PSK_TRY(dev_reserve(ctx, ctx->bits, M * (uint64_t)ctx->wpr * 8));
    pt.mark("alloc matrix");
    PSK_HIP(ctx, hipMemsetAsync(ctx->bits.p, 0, M * (uint64_t)ctx->wpr * 8, ctx->stream));
    pt.mark("zero matrix");
    uint64_t base = 0;
    for (int c = 0; c < n_chunks; c++) {
        if (!chunk_total[c]) continue;
        uint64_t again = chunk_rows[c];
        if (n_chunks > 1) PSK_TRY(sort_chunk(c, &again));   // one chunk: its sorted pairs are still in place
        if (again != chunk_rows[c]) return psk_fail(ctx, PSK_ESTATE, "chunk %d numbered %llu rows, then %llu", c,
                                                    (unsigned long long)chunk_rows[c], (unsigned long long)again);
        presence_fill_kernel<<<div_up(chunk_total[c], 256), 256, 0, ctx->stream>>>(
            sorted, chunk_total[c], sbits, flags, ctx->wpr, ctx->union_words.as<uint64_t>() + base,
            reinterpret_cast<unsigned long long *>(ctx->bits.p) + base * (uint64_t)ctx->wpr, kv ? sorted_vals : nullptr);
        PSK_HIP(ctx, hipGetLastError());
        base += chunk_rows[c];
    }
    PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    pt.mark("fill");
    ctx->n_kmers = M;
    ctx->have_presence = true;
    ctx->dense_hint = -1;
    ctx->last.valid = false;
    if (n_kmers) *n_kmers = M;
    return PSK_OK;
}

extern "C" int psk_presence_shape(psk_ctx *ctx, uint64_t *n_kmers, int *words_per_row, int *n_samples)
{
    if (!ctx) return PSK_EINVAL;
    if (!ctx->have_presence) return psk_fail(ctx, PSK_ESTATE, "no presence matrix");
    if (n_kmers) *n_kmers = ctx->n_kmers;
    if (words_per_row) *words_per_row = ctx->wpr;
    if (n_samples) *n_samples = ctx->n_samples;
    return PSK_OK;
}

extern "C" int psk_get_union(psk_ctx *ctx, uint64_t *words, uint64_t cap)
{
    if (!ctx) return PSK_EINVAL;
    if (!ctx->have_presence) return psk_fail(ctx, PSK_ESTATE, "no presence matrix");
    if (cap < ctx->n_kmers) return psk_fail(ctx, PSK_ERANGE, "buffer too small");
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->n_kmers && words)
        PSK_HIP(ctx, hipMemcpy(words, ctx->union_words.p, ctx->n_kmers * 8, hipMemcpyDeviceToHost));
    return PSK_OK;
}

extern "C" int psk_get_rows(psk_ctx *ctx, const uint64_t *row_idx, uint64_t n, uint64_t *bits_out)
{
    if (!ctx) return PSK_EINVAL;
    if (!ctx->have_presence) return psk_fail(ctx, PSK_ESTATE, "no presence matrix");
    if (n == 0) return PSK_OK;
    if (!row_idx || !bits_out) return psk_fail(ctx, PSK_EINVAL, "null buffer");
    for (uint64_t i = 0; i < n; i++)
        if (row_idx[i] >= ctx->n_kmers) return psk_fail(ctx, PSK_EINVAL, "row index out of range");
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t nb = n * (uint64_t)ctx->wpr * 8;
    PSK_TRY(dev_reserve(ctx, ctx->flags, n * 8));
    PSK_TRY(dev_reserve(ctx, ctx->starts, nb));
    PSK_HIP(ctx, hipMemcpyAsync(ctx->flags.p, row_idx, n * 8, hipMemcpyHostToDevice, ctx->stream));
    gather_rows_kernel<<<div_up(n * (uint64_t)ctx->wpr, 256), 256, 0, ctx->stream>>>(
        ctx->bits.as<uint64_t>(), ctx->wpr, ctx->flags.as<uint64_t>(), n, ctx->starts.as<uint64_t>());
    PSK_HIP(ctx, hipGetLastError());
    PSK_HIP(ctx, hipMemcpyAsync(bits_out, ctx->starts.p, nb, hipMemcpyDeviceToHost, ctx->stream));
    PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PSK_OK;
}

extern "C" int psk_intersect_db(psk_ctx *ctx, const uint64_t *db_words, uint64_t n_db, uint64_t *n_kmers)
{
    if (!ctx) return PSK_EINVAL;
    if (ctx->n_in_flight > 0) return psk_fail(ctx, PSK_ESTATE, "a scan is in flight on this matrix: psk_scan_end first");
    if (!ctx->have_presence) return psk_fail(ctx, PSK_ESTATE, "no presence matrix");
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t M = ctx->n_kmers;
    if (M == 0) { if (n_kmers) *n_kmers = 0; return PSK_OK; }
    PSK_TRY(dev_reserve(ctx, ctx->keysA, (n_db ? n_db : 1) * 8));
    if (n_db) PSK_HIP(ctx, hipMemcpyAsync(ctx->keysA.p, db_words, n_db * 8, hipMemcpyHostToDevice, ctx->stream));
    PSK_TRY(dev_reserve(ctx, ctx->flags, (M + 1) * 4));
    PSK_TRY(dev_reserve(ctx, ctx->misc, 64));
    uint32_t *keep = ctx->flags.as<uint32_t>();
    uint32_t *d_m = ctx->misc.as<uint32_t>() + 4;
    db_member_kernel<<<div_up(M, 256), 256, 0, ctx->stream>>>(ctx->union_words.as<uint64_t>(), M,
                                                             ctx->keysA.as<uint64_t>(), n_db, keep);
    PSK_HIP(ctx, hipGetLastError());
    // sentinel so that pos[M] = number kept; compact_rows reads pos[r+1] != pos[r]
    PSK_HIP(ctx, hipMemsetAsync(keep + M, 0, 4, ctx->stream));
    PSK_TRY(dev_exclusive_scan_u32(ctx, keep, keep, M + 1, d_m));
    uint32_t kept = 0;
    PSK_HIP(ctx, hipMemcpyAsync(&kept, d_m, 4, hipMemcpyDeviceToHost, ctx->stream));
    PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PSK_TRY(dev_reserve(ctx, ctx->keysB, (uint64_t)(kept ? kept : 1) * (1 + ctx->wpr) * 8));
    uint64_t *w2 = ctx->keysB.as<uint64_t>();
    uint64_t *b2 = w2 + (kept ? kept : 1);
    compact_rows_kernel<<<div_up(M, 256), 256, 0, ctx->stream>>>(ctx->union_words.as<uint64_t>(),
                                                                ctx->bits.as<uint64_t>(), ctx->wpr, M, keep, w2, b2);
    PSK_HIP(ctx, hipGetLastError());
    PSK_HIP(ctx, hipMemcpyAsync(ctx->union_words.p, w2, (uint64_t)kept * 8, hipMemcpyDeviceToDevice, ctx->stream));
    PSK_HIP(ctx, hipMemcpyAsync(ctx->bits.p, b2, (uint64_t)kept * ctx->wpr * 8, hipMemcpyDeviceToDevice, ctx->stream));
    PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->n_kmers = kept;
    ctx->last.valid = false;
    if (n_kmers) *n_kmers = kept;
    return PSK_OK;
}

extern "C" int psk_set_presence(psk_ctx *ctx, const uint64_t *words, const uint64_t *bits, uint64_t n_kmers,
                                int words_per_row, int n_samples)
{
    if (!ctx) return PSK_EINVAL;
    if (ctx->n_in_flight > 0) return psk_fail(ctx, PSK_ESTATE, "a scan is in flight on this matrix: psk_scan_end first");
    if (n_samples < 1 || words_per_row != padded_wpr(n_samples))
        return psk_fail(ctx, PSK_EINVAL, "words_per_row must be %d for %d samples", padded_wpr(n_samples), n_samples);
    if (!bits && n_kmers) return psk_fail(ctx, PSK_EINVAL, "null matrix");
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    reset_lists(ctx, n_samples);
    ctx->n_samples = n_samples;
    ctx->wpr = words_per_row;
    ctx->n_kmers = n_kmers;
    const uint64_t m1 = n_kmers ? n_kmers : 1;
    PSK_TRY(dev_reserve(ctx, ctx->union_words, m1 * 8));
    PSK_TRY(dev_reserve(ctx, ctx->bits, m1 * (uint64_t)words_per_row * 8));
    if (n_kmers) {
        if (words) PSK_HIP(ctx, hipMemcpy(ctx->union_words.p, words, n_kmers * 8, hipMemcpyHostToDevice));
        else {
            iota_words_kernel<<<div_up(n_kmers, 256), 256, 0, ctx->stream>>>(ctx->union_words.as<uint64_t>(), n_kmers);
            PSK_HIP(ctx, hipGetLastError());
        }
        PSK_HIP(ctx, hipMemcpy(ctx->bits.p, bits, n_kmers * (uint64_t)words_per_row * 8, hipMemcpyHostToDevice));
    }
    PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->have_presence = true;
    ctx->dense_hint = -1;
    ctx->last.valid = false;
    return PSK_OK;
}

extern "C" int psk_synth_presence(psk_ctx *ctx, uint64_t n_kmers, int n_samples, uint64_t seed)
{
    if (!ctx) return PSK_EINVAL;
    if (ctx->n_in_flight > 0) return psk_fail(ctx, PSK_ESTATE, "a scan is in flight on this matrix: psk_scan_end first");
    if (n_samples < 1 || n_kmers < 1) return psk_fail(ctx, PSK_EINVAL, "bad shape");
    PSK_HIP(ctx, hipSetDevice(ctx->device));
    reset_lists(ctx, n_samples);
    ctx->n_samples = n_samples;
    ctx->wpr = padded_wpr(n_samples);
    ctx->n_kmers = n_kmers;
    PSK_TRY(dev_reserve(ctx, ctx->union_words, n_kmers * 8));
    PSK_TRY(dev_reserve(ctx, ctx->bits, n_kmers * (uint64_t)ctx->wpr * 8));
    iota_words_kernel<<<div_up(n_kmers, 256), 256, 0, ctx->stream>>>(ctx->union_words.as<uint64_t>(), n_kmers);
    PSK_HIP(ctx, hipGetLastError());
    synth_presence_kernel<<<div_up(n_kmers * (uint64_t)ctx->wpr, 256), 256, 0, ctx->stream>>>(
        ctx->bits.as<uint64_t>(), n_kmers, ctx->wpr, n_samples, seed);
    PSK_HIP(ctx, hipGetLastError());
    PSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->have_presence = true;
    ctx->dense_hint = -1;
    ctx->last.valid = false;
    return PSK_OK;
}
