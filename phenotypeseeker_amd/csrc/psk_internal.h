// Internal declarations shared by the translation units of libpsk.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/psk.h"

// ---- growable device buffer -------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;  // bytes
    bool borrowed = false;  // p points into a slab someone else owns (psk_ctx::lane_slab): never freed through this buffer
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// Bump allocator for the per-sample lists: thousands of odd-sized hipMallocs fragment the device
// address space (the 5 GB matrix allocated afterwards then streams at a fraction of HBM speed, r01
// cfg-3 probe), so lists are carved out of a few 1 GiB chunks instead.
struct Arena {
    std::vector<void *> chunks;
    std::vector<size_t> sizes;
    size_t cur = 0, off = 0;
    static constexpr size_t CHUNK = 1ull << 30;
};

// A sample's list in one of two forms.  Sparse: words[] + freqs[] (every k).  Dense (2k <= 26, dense_count.hip):
// a bitmap over the slab's buckets of the word space (bit v of the bitmap = word dense_b0 * 2^15 + v occurs) plus the
// sorted words with a count of two or more -- 8 MB + a few MB per 5-Mbp sample at k = 13 instead of 56 MB, and what
// the presence build transposes directly; words[] / freqs[] are materialised from it only when a caller asks for
// them (psk_get_list, the list exchange).
struct SampleList {
    uint64_t *words = nullptr;  // device, ascending canonical words (this slab)
    uint32_t *freqs = nullptr;  // device
    uint64_t n_unique = 0;
    uint64_t n_total = 0;
    bool done = false;
    bool dense = false;
    uint64_t *bitmap = nullptr;   // device, dense_nb * 512 u64
    uint32_t *mwords = nullptr;   // device, ascending words (absolute, < 2^26) with count >= 2
    uint32_t *mfreqs = nullptr;   // device, their counts
    uint64_t n_multi = 0;
};

// One of the three buffer sets of the pipelined batch counter (psk_count_kmers_batch): the chain of sample i
// runs on set i % 3 while the host finalises sample i - 1 from its set and samples i + 1, i + 2 are uploaded and
// framed into theirs on the copy stream.
struct CountLane {
    static constexpr uint32_t CNT_SLOTS = 4096;
    DevBuf raw, keysA, keysB, starts, cnt;
    DevBuf rawin, fr_scratch;        // file bytes as uploaded and the tile tables of the GPU framing (frame_gpu.hip)
    uint32_t *pinned_cnt = nullptr;  // pinned host landing: [0] windows seen by the GPU, [1] unique words
    hipEvent_t done = nullptr, raw_ready = nullptr, raw_free = nullptr, up_done = nullptr;
    bool raw_used = false;
    bool exact = true;               // n is exact (no slab filter); otherwise an upper bound
    uint32_t cnt_slot = 0;           // next unused 16-byte counter slot of `cnt` (zeroed CNT_SLOTS at a time)
    int sample = -1;                 // sample whose chain is in flight on this set (-1: none)
    uint64_t n = 0;                  // its window count (known on the host from the framing)
    uint64_t *uniq = nullptr;        // device: its unique words (one of keysA / keysB)
    // asynchronous MinHash sketch of the same sample (minhash.hip: sketch_enqueue / sketch_collect)
    DevBuf sk_cand, sk_out;          // [2 x u32 counters, pad | candidate hashes]; [distinct count | sketch]
    uint64_t *sk_host = nullptr;     // pinned landing of sk_out
    size_t sk_host_cap = 0;
    hipEvent_t sk_done = nullptr, sk_filtered = nullptr;
    int sk_state = 0;                // 0 none, 1 queued, 2 the synchronous route has to serve this sample
    // dense counting (dense_count.hip): bucketed keys, per-(tile, bucket) offsets, counter-slot ring, per-bucket
    // results of the sample in flight, multi-count entries before compaction
    DevBuf dc_part, dc_wgoff, dc_cnt, dc_meta, dc_mtemp;
    uint32_t dc_slot = 0;
    bool dense = false;              // the chain in flight on this set is a dense one
    bool dc_defer_compact = false;   // ... queued as part of a group (dense_group_enqueue): the group compacts in one launch
    bool group_pending = false;      // chain_compute left this sample's dense chain to its group (clean_len: its clean stream)
    uint64_t clean_len = 0;
    bool bs = false;                 // ... a bucketed-sort one (bucket_count.hip; it borrows the dc_* buffers)
};

struct ScanParams {  // what psk_rescan_timed needs to re-launch the last chi2 scan
    bool valid = false;
    bool weighted = false;
    int min_samples = 0, max_samples = 0;
    double pvalue_cutoff = 0;
    int omit_B = 0;
    uint64_t n_kmers_global = 0;
    int n1 = 0, n0 = 0;      // class sizes of the last chi2 scan
    int inline_masks = 0;    // phenotype masks small enough to ride in the kernel arguments
    uint64_t m1[16] = {0}, m0[16] = {0};
    double W1 = 0, W0 = 0;   // class weight totals
};

// One of the two result sets of the scans.
struct ScanSlot {
    DevBuf res;                      // SoA result arrays (setup_results)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;   // around the scan's kernels
    hipEvent_t ev_export = nullptr;  // recorded on the caller's stream after an asynchronous export of this set
    bool export_pending = false;     // the next scan that writes this set waits for ev_export on the device
    bool in_flight = false;
    uint64_t seq = 0;                // launch order
    uint64_t seg_cap = 0;            // entries per result segment of the scan that wrote this set
};

struct psk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // Two result sets, so that up to two scans can be in flight (psk_chi2_scan_begin twice before psk_scan_end) and
    // the asynchronous export of scan i's survivors overlaps scan i + 1: see ScanSlot.
    ScanSlot slot[2];
    int res_set = 0;                  // set of the last scan ENDED (= the one the result calls read)
    bool results_valid = false;       // false once a later psk_chi2_scan_begin has taken that set again
    int n_in_flight = 0;              // scans launched and not yet ended (0..2)
    uint64_t scan_seq = 0;
    std::string err;
    int n_cu = 0;

    // run configuration
    int k = 0;
    int n_samples = 0;
    uint64_t slab_lo = 0, slab_hi = 0;  // slab_hi == 0: unbounded
    std::vector<SampleList> lists;
    Arena arena;
    // dense list form (see SampleList): on for this run?  first bucket and number of buckets of the slab
    bool dense_mode = false;
    bool dense_defer = false;        // psk_count_kmers_batch is collecting genomes into launch groups (dense_group_enqueue)
    uint32_t dense_b0 = 0, dense_nb = 0;
    // bucketed sort of k = 14..32 (bucket_count.hip): 2,048 splitters taken from the first list of the run
    DevBuf bs_spl, bs_ct;             // splitters; bucket of the first word of each of 4,096 cells of the run's word range
    bool bs_ready = false;
    uint32_t bs_nb = 0, bs_shift = 0;
    uint64_t bs_lo = 0;               // first word of the cells' range (the slab's)
    double bs_keep = 1.0;             // share of a sample's windows the slab keeps (from that list)

    // scratch for per-sample counting
    DevBuf raw, keysA, keysB, valsA, valsB, hist, scan_tmp, flags, starts, misc;
    void *pinned = nullptr;   // pinned host staging for the clean stream
    size_t pinned_cap = 0;
    std::vector<void *> ring;       // pinned ring of the batch counter
    std::vector<size_t> ring_cap;
    void *scan_pinned = nullptr;  // pinned staging for the scan's masks / weights
    size_t scan_pinned_cap = 0;
    void *cnt_pinned = nullptr;   // pinned landing buffer of the scan's result counters
    static constexpr int LANES = 24;
    // .gz inputs (gz_inflate.hip): what a run of them is inflated in.  Two sets of the buffers that outlive the inflate -- the
    // compressed images in host memory, their copy and the text on the device --, because a call's runs are a pipeline: one is
    // read and uploaded while the one before it is inflated (on gz_stream) while the one before that is counted; symbols, matches and tables are the
    // inflate's own.  Kept from call to call (a hipMalloc of these costs ~30 ms per GB, fresh host pages and their release
    // 0.3 s per 2 GB); given back by psk_build_presence and psk_free.
    uint8_t *gz_host[2] = {nullptr, nullptr};
    size_t gz_host_cap[2] = {0, 0};
    std::vector<std::pair<void *, size_t>> gz_maps[2];   // r06: .gz FILES are mapped, not read into gz_host (which in-memory callers and unmappable files still use)
    std::thread gz_reaper;   // gives gz_host back off the caller's critical path (gz_release: 5 GB of host pages took 0.56 s to unmap)
    DevBuf gz_comp[2], gz_out[2], gz_sym, gz_rec, gz_tab;
    hipStream_t gz_stream = nullptr, gz_up_stream = nullptr;   // the inflate's kernels; the uploads of the run after it
    DevBuf lane_slab;        // one allocation behind the buffer sets of a grouped batch (a cold run paid 60 ms for 170 hipMallocs)
    uint32_t *lane_pinned = nullptr;   // ... and one pinned block behind their counters (16 u32 per set)
    CountLane lane[LANES];   // sample i runs on set i % 3: i + 1 and i + 2 are uploaded / framed ahead while chain i runs; in groups
                             // of G genomes (dense counting) on set i % (3 G): two groups ahead, one in flight
    hipStream_t copy_stream = nullptr;  // uploads of the batch counter overlap the previous sample's kernels
    hipStream_t copy_more[3] = {nullptr, nullptr, nullptr};   // ... and rotate over up to four streams (PSK_COPY_STREAMS, default 2): a copy is queued while one runs
    hipStream_t frame_stream = nullptr; // the GPU framing of sample i + 1 runs beside upload i + 2 and chain i
    hipStream_t sketch_stream = nullptr;  // the one-workgroup sketch select runs beside the next sample's chain

    // presence matrix
    uint64_t n_kmers = 0;
    int wpr = 0;  // u64 words per row: 1 up to 64 samples, else even
    DevBuf union_words, bits;
    bool have_presence = false;

    // scan state
    DevBuf mask1, phe, res_count, res_sorted;
    DevBuf lut;                          // nibble table of the moment scans (assoc_scan.hip row_moments_lut)
    bool lut_valid = false;              // ... holds the table of the last weighted chi2 scan
    bool lut6_valid = false;             // ... in its six-bit f32 form (row_moments_f32)
    uint64_t n_pass = 0;
    uint64_t res_seg_cap = 0;            // entries per result segment of the last scan
    std::vector<uint32_t> seg_counts;    // survivors per segment
    int last_scan_kind = 0;  // 1 chi2, 2 ttest
    int dense_hint = -1;     // did the last chi2 scan of this matrix keep > 0.1 % of the rows?  (-1: no scan yet)
    double last_scan_ms = 0;
    ScanParams last;

    void *comm = nullptr;    // PskComm (comm.hip): RCCL communicator + its stream, multi-GPU runs only
};

// ---- error helpers ----------------------------------------------------------------------------
int psk_fail(psk_ctx *ctx, int code, const char *fmt, ...);
std::string psk_error_text(psk_ctx *ctx);                        // (copies under the lock psk_fail writes under)
void psk_set_error_text(psk_ctx *ctx, const std::string &text);
void psk_forget_lane_slices(psk_ctx *ctx);   // kmer_count.hip

#define PSK_HIP(ctx, call)                                                                            \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return psk_fail((ctx), PSK_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                            __FILE__, __LINE__);                                                      \
    } while (0)

#define PSK_TRY(expr)              \
    do {                           \
        int rc_ = (expr);          \
        if (rc_ != PSK_OK) return rc_; \
    } while (0)

int dev_reserve(psk_ctx *ctx, DevBuf &b, size_t bytes);  // grow-only, contents NOT preserved
void dev_release(DevBuf &b);
void reset_lists(psk_ctx *ctx, int n_samples);
int arena_alloc(psk_ctx *ctx, size_t bytes, void **out);   // 256-byte aligned, freed by reset_lists
void arena_release(psk_ctx *ctx);  // frees the per-sample lists, resizes to n_samples
void comm_release(psk_ctx *ctx);   // comm.hip: destroys the context's RCCL communicator, if any

// gz_inflate.hip: the text of one .gz image -- in the group's device buffer, or (the device route declined it) from zlib
struct GzInflated {
    bool on_device = false, bgzf = false;
    int chunks = 0;
    uint64_t off = 0, len = 0;
    uint64_t first_nul = 0;   // on_device: where the text has its first NUL byte (len: nowhere)
    std::vector<uint8_t> host;
    std::string error;        // zlib refused the file: why
};
// r06: what psk_build_presence paid for the .gz inputs' buffers was the HOST's side -- 5.2 GB of compressed images: 0.53-0.56 s to
// unmap, of cfg5gz's 2.3 s -- not the device's (69.6 GB: 2 ms; tools/free_probe.py).  gz_release_device: at the start of the matrix
// build (the memory is wanted); gz_release_host: on a helper thread started when the build has been queued (a thread that unmaps
// holds the address-space lock the build's own allocations need); wait: join it (psk_free; whoever needs gz_host again joins too).
void gz_release_device(psk_ctx *ctx);
void gz_release_host(psk_ctx *ctx, bool wait);
void gz_release(psk_ctx *ctx);   // both, waited for (psk_free)
int gz_inflate_group(psk_ctx *ctx, int n, const uint8_t *const *data, const size_t *sizes, DevBuf &comp_buf, DevBuf &sym_buf, DevBuf &rec_buf, DevBuf &out_buf,
                     DevBuf &tab_buf, std::vector<GzInflated> &res, double *device_ms, bool host_only = false, int host_threads = 8,
                     hipStream_t on_stream = nullptr, bool images_uploaded = false);
uint64_t gz_image_layout(int n, const size_t *sizes, uint64_t *at);
bool gz_group_on_device(int n, const size_t *sizes, bool host_only, int host_threads);

// r06: pinned host buffers outlive their context in a process-wide cache (api.hip).  Unpinning is slow -- the 20 slots of the batch
// counter's ring cost psk_free 28 ms of a 0.59-s `phenotypeseeker modeling` process, and a second context of the process (bench.py's
// e2e leg, a host application that runs several analyses) pinned them all over again in its first counting call.  At most
// PSK_PINNED_CACHE_MB (default 1024; 0: no cache) are held, per device; what a process holds at its exit goes with the process.
int pinned_acquire(psk_ctx *ctx, size_t need, void **buf, size_t *cap);   // *buf of *cap >= need bytes, from the cache or hipHostMalloc
void pinned_release(psk_ctx *ctx, void *buf, size_t cap);                 // into the cache, or hipHostFree when the cache is full

static inline unsigned div_up(uint64_t a, uint64_t b) { return (unsigned)((a + b - 1) / b); }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device: "done once" has to be remembered per
// device, and contexts may count from different threads (a function-local `static bool` let a second GPU launch
// the 64-76 KB kernels without the attribute).  One of these per kernel; first(device) is true exactly once per device.
#include <atomic>
struct PerDeviceOnce {
    std::atomic<uint64_t> seen[4] = {};   // 256 devices
    bool first(int device)
    {
        const uint64_t bit = 1ull << (device & 63);
        return (seen[(device >> 6) & 3].fetch_or(bit) & bit) == 0;
    }
};

// ---- device primitives (scan.hip / radix_sort.hip) ---------------------------------------------
// Exclusive prefix sum of n u32 values (in == out allowed).  total_out (device u32*) may be null.
int dev_exclusive_scan_u32(psk_ctx *ctx, const uint32_t *in, uint32_t *out, uint64_t n, uint32_t *total_out);
// Stable LSD radix sort of n u64 keys on bits [bit_lo, bit_hi).  Sorted data ends in *sorted_out
// (either a or b).  n < 2^32.
int dev_radix_sort_u64(psk_ctx *ctx, uint64_t *a, uint64_t *b, uint64_t n, int bit_lo, int bit_hi,
                       uint64_t **sorted_out, const uint32_t *n_dev = nullptr);

// Same with a u32 payload per key (va/vb double buffer); sorted payloads end in *sorted_vals_out.
int dev_radix_sort_kv(psk_ctx *ctx, uint64_t *a, uint64_t *b, uint32_t *va, uint32_t *vb, uint64_t n, int bit_lo,
                      int bit_hi, uint64_t **sorted_out, uint32_t **sorted_vals_out, const uint32_t *n_dev = nullptr);

// ---- stages --------------------------------------------------------------------------------------
int64_t frame_sequence_host(const uint8_t *bytes, size_t len, uint8_t *out, size_t out_cap);
// clean stream (device) -> canonical words inside [lo, hi) appended to out; *n_out (device u32) counts them
int launch_extract(psk_ctx *ctx, const uint8_t *clean, uint64_t len, int k, uint64_t lo, uint64_t hi, uint64_t *out,
                   uint32_t *n_out);

// ---- dense list form (dense_count.hip, presence_dense.hip) ---------------------------------------------------
constexpr int DC_VB = 15;                  // word values per bucket: 2^15
constexpr int DC_BUCKET_WORDS = 512;       // u64 bitmap words per bucket
constexpr uint32_t DC_MAX_NB = 2048;       // 2k <= 26
void dense_configure(psk_ctx *ctx);        // psk_begin: decides dense_mode / dense_b0 / dense_nb for the run
// the dense counting chain of one sample on buffer set L (clean stream already queued for upload into L.raw);
// n = its window count.  dense_chain_finalize (one sample later) places the multi-count entries in the arena.
int dense_chain_enqueue(psk_ctx *ctx, CountLane &L, int sample_idx, uint64_t clean_len, uint64_t n);
int dense_chain_finalize(psk_ctx *ctx, CountLane &L, uint64_t *n_kept, uint64_t *n_unique);
int dense_group_size();   // samples per launch chain (PSK_DC_GROUP, default and at most 8)
bool dense_group_ok(const psk_ctx *ctx, uint64_t n);
int dense_group_enqueue(psk_ctx *ctx, CountLane *const *lanes, const int *sample_idx, const uint64_t *clean_len, const uint64_t *n,
                        int count);
int dense_group_compact(psk_ctx *ctx, CountLane *const *lanes, const int *sample_idx, int count);
// the same for the bucketed sort (k = 14..32, bucket_count.hip)
int bucket_group_enqueue(psk_ctx *ctx, CountLane *const *lanes, const int *sample_idx, const uint64_t *clean_len, const uint64_t *n,
                         int count);
int bucket_group_compact(psk_ctx *ctx, CountLane *const *lanes, const int *sample_idx, int count);
void bucket_lane_bytes(const psk_ctx *ctx, size_t max_len, size_t out[5]);
// bytes of the five dc_* buffers of one buffer set for samples of up to max_len clean bases (dc_part, dc_wgoff, dc_cnt, dc_meta, dc_mtemp)
void dense_lane_bytes(const psk_ctx *ctx, size_t max_len, size_t out[5]);
// words[] / freqs[] of samples [first, first + n) from their dense form (no-op for sparse or materialised ones)
int dense_materialize(psk_ctx *ctx, int first, int n);
int dense_lookup_counts(psk_ctx *ctx, const SampleList &L, const uint64_t *d_query, uint64_t n, uint32_t *d_out);
int build_presence_dense(psk_ctx *ctx, uint64_t *n_kmers, int *done);

// ---- bucketed sort for k = 14..32 (bucket_count.hip) ----------------------------------------------------------------
constexpr uint32_t BS_NB = 2048;           // most buckets of equal count a run uses (splitters = quantiles of one of its lists)
bool bucket_route_ok(const psk_ctx *ctx, uint64_t n);
int bucket_splitters_from(psk_ctx *ctx, const SampleList &S, uint64_t windows);
int bucket_chain_enqueue(psk_ctx *ctx, CountLane &L, int sample_idx, uint64_t clean_len, uint64_t n);
int bucket_chain_finalize(psk_ctx *ctx, CountLane &L, SampleList &S, uint64_t n_kept, uint64_t nu, bool *fell_back);
int bucket_fallback_keys(psk_ctx *ctx, CountLane &L, uint64_t n_kept, uint64_t *keys);
