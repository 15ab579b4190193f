/*
 * psk.h -- C ABI of libpsk.so, the MI355X (gfx950) k-mer association engine that replaces the
 * hot path of bioinfo-ut/PhenotypeSeeker's `modeling` / `prediction` commands.
 *
 * Plain C: opaque context, caller-owned host buffers in and out, sizes as integers, every
 * function returns 0 on success or a negative PSK_E* code (psk_last_error() has the text).
 * No torch types, no C++ types.  One context drives one GPU from one host thread; create one
 * context per rank for multi-GPU runs (the k-mer word space is range-sharded, see psk_begin).
 *
 * Each entry point cites the reference interface it replaces (paths under /root/reference;
 * modeling.py = PhenotypeSeeker/modeling.py, prediction.py = PhenotypeSeeker/prediction.py).
 * The reference's own boundary for this path is a set of subprocess command lines plus Python
 * methods; INTEGRATION.md shows the ctypes stubs that rebind those call sites to this ABI.
 */
#ifndef PSK_H
#define PSK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct psk_ctx psk_ctx;

enum {
    PSK_OK = 0,
    PSK_EINVAL = -1,   /* bad argument / call order */
    PSK_ENOMEM = -2,   /* host or device allocation failed */
    PSK_EHIP = -3,     /* a HIP runtime call failed */
    PSK_ERANGE = -4,   /* caller buffer too small / size limit exceeded */
    PSK_ESTATE = -5,   /* required earlier stage has not been run */
    PSK_EGZIP = -6     /* (with PSK_NO_GPU_GZ=1 only; r05: .gz inputs are inflated by the library) a FILE input is gzip-compressed
                          (magic bytes): the caller inflates it and uses the in-memory call */
};

/* ---- lifecycle --------------------------------------------------------------------------- */

/* Binds a context to HIP device `device` (creates its stream and timing events). */
int psk_init(int device, psk_ctx **ctx_out);
void psk_free(psk_ctx *ctx);
/* Text of the last error on this context (or of the failed psk_init when ctx is NULL). */
const char *psk_last_error(const psk_ctx *ctx);
/* ABI version of the library: (major << 16) | minor. */
int psk_version(void);
/* Device facts for reports: name (NUL-terminated, truncated to name_cap), CU count, HBM bytes. */
int psk_device_info(psk_ctx *ctx, char *name, int name_cap, int *n_cu, uint64_t *hbm_bytes);

/*
 * Starts a k-mer run: word length k (1..32; modeling.py:161 `-l`), the number of samples that
 * will be counted, and this context's shard [slab_lo, slab_hi) of the canonical 2k-bit word
 * space (slab_hi == 0 means "to the end"; one GPU: 0, 0).  Drops all state of a previous run.
 */
int psk_begin(psk_ctx *ctx, int k, int n_samples, uint64_t slab_lo, uint64_t slab_hi);

/* ---- a1: per-sample k-mer list ------------------------------------------------------------
 * Replaces `glistmaker <addr> -o K-mer_lists/<name>_0 -w <k> -c <cutoff>`
 * (Samples.get_kmer_lists, modeling.py:303-315; bundled binary bin/glistmaker 4.2.3).
 * `bytes` is the inflated FASTA/FASTQ file image.  The sorted canonical list (words + u32
 * frequencies, exactly the records of glistmaker's .list file, restricted to the slab) stays
 * in HBM under `sample_idx`; n_unique / n_total report its size.  As in the bundled binary,
 * no frequency cut-off is applied.
 */
int psk_count_kmers(psk_ctx *ctx, int sample_idx, const uint8_t *bytes, size_t len, uint64_t *n_unique,
                    uint64_t *n_total);
/* Batch form of psk_count_kmers for samples first_sample_idx .. first_sample_idx + n - 1: n_threads host
 * threads tokenise ahead into a ring of pinned buffers while the calling thread drives the GPU half
 * in order, so host framing overlaps device work.  n_unique / n_total: n entries each (may be NULL). */
int psk_count_kmers_batch(psk_ctx *ctx, int first_sample_idx, int n, const uint8_t *const *bytes, const size_t *lens,
                          uint64_t *n_unique, uint64_t *n_total, int n_threads);
/* The same call that also returns every sample's MinHash sketch (psk_minhash_sketch below) from the clean
 * stream that is already on the device for counting: the `-w` path (get_kmer_lists + get_mash_sketches,
 * modeling.py:303-315, :386-390) needs both and each file is framed and uploaded once.
 *   sketch_k == 0: no sketches (hashes_out / n_hashes_out may be NULL)
 *   hashes_out[n][sketch_size] ascending distinct hashes, n_hashes_out[n] = how many of each row are valid
 */
int psk_count_kmers_batch_sketch(psk_ctx *ctx, int first_sample_idx, int n, const uint8_t *const *bytes,
                                 const size_t *lens, uint64_t *n_unique, uint64_t *n_total, int n_threads,
                                 int sketch_k, int sketch_size, uint32_t sketch_seed, uint64_t *hashes_out,
                                 uint64_t *n_hashes_out);
/* The same for UNCOMPRESSED files on disk (paths[i] of sizes[i] bytes): the framing threads read them, so no file
 * image crosses the caller's language boundary.  A .gz file is read as it is and inflated by the library (psk_gz_inflate
 * below says how). */
int psk_count_kmers_files(psk_ctx *ctx, int first_sample_idx, int n, const char *const *paths, const size_t *sizes,
                          uint64_t *n_unique, uint64_t *n_total, int n_threads, int sketch_k, int sketch_size,
                          uint32_t sketch_seed, uint64_t *hashes_out, uint64_t *n_hashes_out);
/* .gz inputs ("FASTA/FASTQ(.gz)", glistmaker's zlib reader: SURVEY.md section 2 row 9).  psk_count_kmers /
 * psk_count_kmers_batch / psk_count_kmers_files take a gzip image (magic bytes 1f 8b) as it is: the compressed bytes
 * cross PCIe and DEFLATE is decoded on the device (csrc/gz_inflate.hip) -- all .gz samples of a call together, 8 GiB of text
 * at a time (PSK_GZ_GROUP_MB); a group of so few small files that zlib would be faster (the device route has a floor of ~40 ms) and a member the device
 * route declines go through zlib on the call's host threads.  This entry point is the inflate on its own -- tests and measurements: n gzip images ->
 * their text (out[i], of capacity out_cap[i], may be NULL: lengths only).  route[i]: 1 decoded on the device, 2 the same,
 * a BGZF file (its members found by their BSIZE fields), 0 zlib on the host, -1 zlib refused the file.  A refused file fails
 * the call (PSK_EINVAL, zlib's words for the first one, as glistmaker's reader fails) -- after every file has been tried: with
 * `route` given, the texts and lengths of the other files are delivered all the same.  device_ms: wall-clock of the device
 * route, upload included.  PSK_GZ_GUARD=1 (the fuzz test): 64-KB guard bands around the device buffers, checked after the
 * last kernel (PSK_ESTATE when one was written to). */
int psk_gz_inflate(psk_ctx *ctx, int n, const uint8_t *const *data, const size_t *sizes, uint8_t *const *out,
                   const size_t *out_cap, uint64_t *out_len, int32_t *route, double *device_ms);
/* Copies sample_idx's list to the host (for writing .list files / parity checks). */
int psk_get_list(psk_ctx *ctx, int sample_idx, uint64_t *words, uint32_t *freqs, uint64_t cap);
/* ---- multi-GPU ingest: count each sample on ONE rank, exchange the slab ranges of the sorted lists ----------
 * (no counterpart in the reference, which has no multi-process counting; see phenotypeseeker_amd/dist.py
 * ListExchange.)  A slab of the word space is a contiguous range of a sorted list, so the hand-over is three calls:
 *   psk_lists_split     offsets_out[i * n_bounds + b] = number of words of sample first + i below bounds[b]
 *                       (bounds ascending; a bound of 0 after the first entry means "end of the word space")
 *   psk_copy_list_ranges  packs ranges [start[r], start[r] + count[r]) of the lists sample_idx[r], back to back,
 *                         into DEVICE buffers (the send buffers of the all-to-all); waited for
 *   psk_set_lists_device  installs n_lists lists held back to back in DEVICE memory -- count[r] (word, count)
 *                         entries each, ascending, inside this context's slab (checked: PSK_EINVAL otherwise, and
 *                         none of them is kept) -- as the lists of sample_idx[r]; they are copied into the
 *                         context's own storage.  n_total[r] (may be NULL) is what psk_count_kmers would report.
 *   psk_release_lists   gives the device memory of this context's lists back (every sample is "not counted"
 *                       again): the counting context's lists are dead once they are packed, and at N = 2 of
 *                       config 3 they are 61 GB that the receive buffers need */
int psk_lists_split(psk_ctx *ctx, int first_sample_idx, int n, const uint64_t *bounds, int n_bounds,
                    uint64_t *offsets_out);
int psk_copy_list_ranges(psk_ctx *ctx, int n_ranges, const int32_t *sample_idx, const uint64_t *start,
                         const uint64_t *count, void *device_words_dst, void *device_freqs_dst);
int psk_release_lists(psk_ctx *ctx);
int psk_set_lists_device(psk_ctx *ctx, int n_lists, const int32_t *sample_idx, const uint64_t *count,
                         const uint64_t *n_total, const void *device_words, const void *device_freqs);
/* Frequencies of `n` given canonical words in sample_idx's list (0 if absent): the
 * `glistquery <sample>.list -l` mapping of modeling.py:324-329 restricted to the k-mers the
 * caller still needs (--real_counts columns of the ML matrix, modeling.py:693-695). */
int psk_lookup_counts(psk_ctx *ctx, int sample_idx, const uint64_t *words, uint64_t n, uint32_t *freqs);

/* ---- a2+a3: feature vector (union) and the k-mer x sample presence matrix ------------------
 * Replaces the `glistcompare -u` reduction tree (Samples.get_feature_vector / get_union,
 * modeling.py:350-380) and the per-sample `glistquery ... -l feature_vector.list` + `split`
 * text mapping (Samples.map_samples, modeling.py:317-348).  Result, resident in HBM:
 *   words[M]            ascending canonical words of the union (this slab)
 *   bits[M][wpr]        u64 words, sample i = bit (i & 63) of word (i >> 6); wpr = 1 up to 64 samples, else even
 * n_kmers = M is this slab's share of phenotypes.no_kmers_to_analyse (modeling.py:644).
 */
int psk_build_presence(psk_ctx *ctx, uint64_t *n_kmers);
int psk_presence_shape(psk_ctx *ctx, uint64_t *n_kmers, int *words_per_row, int *n_samples);
int psk_get_union(psk_ctx *ctx, uint64_t *words, uint64_t cap);
/* Gathers `n` rows (by row index) of the matrix: bits_out[n][words_per_row]. */
int psk_get_rows(psk_ctx *ctx, const uint64_t *row_idx, uint64_t n, uint64_t *bits_out);
/* a9, phenotypes.get_ML_df (modeling.py:1112-1145, rows built at :739 / :796): writes `path` = <test>_results_<pheno>.tsv --
 * `header`, then one line per k-mer that passed the scan: k-mer, round(stat, 2), "%.2E" % p, [round(mean_x, 2),
 * round(mean_y, 2) when kind = 1 (t-test),] n_with, "| " + the names of the VALID samples (valid[i] != 0: phenotype not NA)
 * whose bit is set in the k-mer's row -- and, when path_top is not NULL, its first n_top lines again as `path_top`
 * (:1133-1136).  Lines are ordered by the p-value STRINGS as the reference orders them (:1128), ties by k-mer;
 * order_out[n_rows] receives that order (row indices).  words[n_rows] 2-bit words of length k; bits[n_rows][wpr] as
 * psk_get_rows returns them; names = the sample names back to back, name i = bytes [name_off[i], name_off[i + 1]).
 * Host code (no GPU work; ctx may be NULL: it only carries the error text).  Every byte equals what the reference's
 * DataFrame.to_csv wrote for the same rows: floats as Python's repr, tests/golden/ds_* hold the files. */
int psk_write_result_tables(psk_ctx *ctx, const char *path, const char *path_top, int64_t n_top, const char *header, int kind,
                            int64_t n_rows, const uint64_t *words, int k, const double *stat, const double *p,
                            const double *mean_x, const double *mean_y, const int32_t *n_with, const uint64_t *bits,
                            int words_per_row, int n_samples, const uint8_t *valid, const char *names, const int64_t *name_off,
                            int64_t *order_out);
/* a11, write_model_coefficients_to_file (modeling.py:1414-1455): APPENDS to `path` (whose header line the caller has written)
 * one line per k-mer of the model: k-mer \t repr(coefficient) \t number of samples with it \t "| " + their names.  kmers /
 * kmer_off and names / name_off: texts back to back with their offsets; x[n_samples][n_kmers] (int64, row-major): the model
 * matrix of <pheno>_MLdf.csv, a sample carries a k-mer where it is not 0.  Host code; ctx may be NULL. */
int psk_write_model_coefficients(psk_ctx *ctx, const char *path, int64_t n_kmers, const char *kmers, const int64_t *kmer_off,
                                 const double *coefs, const int64_t *x, int64_t n_samples, const char *names,
                                 const int64_t *name_off);
/* `--kmerDB`: keeps only rows whose word occurs in the (sorted, canonical) db word list --
 * `glistcompare -i` of Samples.get_db_kmers, modeling.py:367-372. */
int psk_intersect_db(psk_ctx *ctx, const uint64_t *db_words, uint64_t n_db, uint64_t *n_kmers);
/* Loads an externally built matrix instead (tests, benchmarks): words may be NULL. */
int psk_set_presence(psk_ctx *ctx, const uint64_t *words, const uint64_t *bits, uint64_t n_kmers,
                     int words_per_row, int n_samples);
/* Fills the resident matrix with a synthetic pattern on the device (benchmark only):
 * row r is present in sample i with a probability that depends on r; see DESIGN.md.  Bits 48..63 of `seed`, when not
 * zero, thin the 1 % of "gene" rows (the rows that survive a scan against the even/odd phenotype) to that many in 10,000
 * of them: 80 gives BASELINE config 2's survivor share. */
int psk_synth_presence(psk_ctx *ctx, uint64_t n_kmers, int n_samples, uint64_t seed);

/* ---- a4-a6: chi-squared scan ---------------------------------------------------------------
 * Replaces phenotypes.get_kmers_tested + conduct_chi_squared_test and helpers
 * (modeling.py:677-714, :759-858) for one binary phenotype.
 *   pheno[n_samples]    1, 0, or -1 for 'NA' (modeling.py:122-126, :810/:817)
 *   weights[n_samples]  GSC weights, or NULL for unit weights (modeling.py:284,:812-822)
 *   min/max_samples     Samples.min_samples / max_samples (modeling.py:233-242,:770-772)
 *   pvalue_cutoff, omit_B, n_kmers_global   the filter of modeling.py:795; n_kmers_global is
 *                       the Bonferroni denominator = union size summed over all slabs
 * Surviving rows are kept on the device; fetch them with psk_get_results.
 */
int psk_chi2_scan(psk_ctx *ctx, const int8_t *pheno, const double *weights, int min_samples, int max_samples,
                  double pvalue_cutoff, int omit_B, uint64_t n_kmers_global, uint64_t *n_pass);
/* The same scan in two halves: _begin launches it and returns, psk_scan_end waits for the OLDEST scan in flight and
 * yields its number of surviving k-mers (with none in flight: the last count again).  Up to two scans may be in
 * flight -- there are two result sets -- so the multi-GPU step runs begin(i+1), end(i), export(i), begin(i+2),
 * all-gather(i): the device always has the next scan queued while the host handles the previous one's survivors.
 * The result calls (psk_get_results, psk_export_survivors*) read the last scan ENDED.  A psk_chi2_scan_begin issued
 * with no scan in flight takes the OTHER result set, so the next scan can be launched before the last one's results
 * are read (the pipeline over phenotypes: end(j), begin(j+1), read j); issued while a scan is in flight it takes
 * the set of the last scan ended (after any asynchronous export of it, on the device), and the result calls fail
 * with PSK_ESTATE until the next psk_scan_end.  A third _begin, and the one-call scans while a scan is in flight,
 * fail with PSK_ESTATE. */
int psk_chi2_scan_begin(psk_ctx *ctx, const int8_t *pheno, const double *weights, int min_samples, int max_samples,
                        double pvalue_cutoff, int omit_B, uint64_t n_kmers_global);
int psk_scan_end(psk_ctx *ctx, uint64_t *n_pass);

/* ---- a7: weighted Welch t-test scan ---------------------------------------------------------
 * Replaces conduct_t_test + get_samples_distribution_for_ttest (modeling.py:716-757) for one
 * continuous phenotype; valid[i] == 0 marks 'NA'.  Bonferroni is always applied (:738).
 */
int psk_ttest_scan(psk_ctx *ctx, const double *pheno, const uint8_t *valid, const double *weights,
                   int min_samples, int max_samples, double pvalue_cutoff, uint64_t n_kmers_global,
                   uint64_t *n_pass);

/* Results of the last scan, ascending by row index (= ascending k-mer).  Any pointer may be
 * NULL.  stat = chi2 or t; mean_x / mean_y are filled by the t-test only. */
int psk_get_results(psk_ctx *ctx, uint64_t *row_idx, uint64_t *words, double *stat, double *p, double *mean_x,
                    double *mean_y, int32_t *n_with, uint64_t cap);
/* Multi-GPU hand-off: writes the survivors of the last scan, unsorted, as records of (6 + words_per_row)
 * u64 { word, stat, p, mean_x, mean_y (f64 bit patterns), n_with (i64), bits[words_per_row] } into a
 * caller-provided DEVICE buffer of 1 + cap_records records; record 0 is a header whose first u64 is the
 * record count.  The buffer can be handed to an RCCL all-gather as is (phenotypeseeker_amd/dist.py).
 * n_records returns the count; records beyond cap_records are dropped (caller retries with a larger cap). */
int psk_export_survivors(psk_ctx *ctx, void *device_dst, uint64_t cap_records, uint64_t *n_records);
/* The same export queued on the CALLER's stream (a hipStream_t, e.g. torch's current stream) and not waited for:
 * work queued on that stream afterwards -- the RCCL all-gather -- is ordered behind it without a host
 * synchronisation; the next scan that writes the same result set waits on the device for the export to finish. */
int psk_export_survivors_async(psk_ctx *ctx, void *device_dst, uint64_t cap_records, void *stream);
/* HIP-event duration of the last scan kernel launch in milliseconds (for bench.py). */
double psk_last_scan_ms(const psk_ctx *ctx);
/* Re-launches the last chi2 scan `reps` times back to back on the context's stream and
 * returns the mean kernel duration in ms measured with HIP events on that stream. */
int psk_rescan_timed(psk_ctx *ctx, int reps, double *mean_ms);
/* The same, returning the HIP-event duration of every one of the `reps` launches in ms_each[reps] (bench.py: the
 * min / median / 95th percentile of the headline kernel; measurement only, no reference call site). */
int psk_rescan_times(psk_ctx *ctx, int reps, double *ms_each);

/* Measurement only (the scan's achieved bandwidth is quoted "against both the 8 TB/s spec and
 * the measured stream-read ceiling"; no reference call site -- the reference has no notion of bandwidth): reads the
 * presence matrix of the context once per launch with a kernel that does nothing else (16 B per lane), `reps` launches
 * timed with HIP events on the context's stream.  Four shapes of that kernel are timed (grid-stride or the scan's own
 * wave-contiguous pieces, non-temporal or plain loads): the fastest is the ceiling, *shape (0..3) says which.
 * *bytes_per_launch = rows x 8 x words per row as stored. */
int psk_stream_read_ceiling(psk_ctx *ctx, int reps, double *mean_ms, uint64_t *bytes_per_launch, int *shape);

/* ---- a10: L1 models over the selected k-mers ------------------------------------------------
 * Replaces the estimator fits behind GridSearchCV (modeling.py:994-1014, :1075-1085,
 * :1208-1216): every (grid value, fold) pair and the refits are independent problems and are
 * solved one per workgroup.
 *   X[n][p]      row-major float design matrix (presence 0/1, or counts with --real_counts)
 *   fold[n]      test-fold id of each sample in 0..n_folds-1; a fit with fold id f trains on
 *                samples whose fold != f; fit_fold[j] == -1 trains on all samples
 *   fit_param[j] C (logistic) or alpha (lasso) of fit j;  n_fits fits in total
 *   coef_out[n_fits][p], icpt_out[n_fits]
 * Logistic: liblinear's L1R_LR objective ||w||_1 + |b| + C sum log(1+exp(-y(w.x+b))).
 * Lasso: (1/2n)||y - Xw - b||^2 + alpha ||w||_1, unpenalised intercept.
 * Device scratch for the duration of the call: the design in both orientations and the per-fit state (a few MB); a 0/1
 * design of up to 4096 samples with 193..1024 columns (65..1024 from 1,024 samples on) also takes a per-fit Gram matrix, n_fits x ~4 p^2 bytes
 * (0.5 GB for 143 fits of 907 columns; the form is skipped beyond 32 GB).
 */
int psk_logreg_l1_fit(psk_ctx *ctx, const float *X, const int32_t *y01, int n, int p, const int32_t *fold,
                      const double *fit_param, const int32_t *fit_fold, int n_fits, double tol, int max_iter,
                      double *coef_out, double *icpt_out, int32_t *iters_out);
int psk_lasso_fit(psk_ctx *ctx, const float *X, const double *y, int n, int p, const int32_t *fold,
                  const double *fit_param, const int32_t *fit_fold, int n_fits, double tol, int max_iter,
                  double *coef_out, double *icpt_out, int32_t *iters_out);

/* ---- f4: the `--penalty L2` estimators --------------------------------------------------------
 * Replaces GridSearchCV over Ridge (set_model, modeling.py:1001-1002) and over
 * LogisticRegression(penalty='l2', solver=<-ls>) (modeling.py:1015-1019, get_logreg_solver :256-264).
 * Same batching and argument meaning as the L1 fits above.
 * Ridge: ||y - Xw - b||^2 + alpha ||w||^2 over the training rows, unpenalised intercept (sklearn's
 *   fit_intercept centring); solved to convergence (scikit-learn's dense solve is direct), so there is
 *   no tol / max_iter.  iters_out = conjugate-gradient steps.
 * Logistic: 0.5 (w'w [+ b^2]) + C sum log(1+exp(-y(w.x+b))); penalise_intercept = 1 for the liblinear
 *   solver (intercept is a penalised constant feature), 0 for lbfgs / newton-cg / sag / saga.  Stops on
 *   liblinear's relative gradient rule or, for the others, max|grad| <= tol * C (scikit-learn 0.22 hands
 *   tol to L-BFGS as gtol on objective / C).  iters_out = Newton steps.
 */
int psk_ridge_fit(psk_ctx *ctx, const float *X, const double *y, int n, int p, const int32_t *fold,
                  const double *fit_param, const int32_t *fit_fold, int n_fits, double *coef_out, double *icpt_out,
                  int32_t *iters_out);
int psk_logreg_l2_fit(psk_ctx *ctx, const float *X, const int32_t *y01, int n, int p, const int32_t *fold,
                      const double *fit_param, const int32_t *fit_fold, int n_fits, double tol, int max_iter,
                      int penalise_intercept, double *coef_out, double *icpt_out, int32_t *iters_out);

/* ---- f1: fixed-dictionary counting (prediction) ---------------------------------------------
 * Replaces `gmer_counter -db <txt> <addr>` (prediction.Samples.map_samples, prediction.py:72-80):
 * occurrences, both strands with multiplicity, of each dictionary k-mer (canonical words) in
 * the file image.  counts_out[n_dict], in dictionary order.
 */
int psk_count_dict(psk_ctx *ctx, const uint8_t *bytes, size_t len, int k, const uint64_t *dict_words,
                   uint64_t n_dict, uint32_t *counts_out);
/* All samples of a prediction run against one dictionary (the Pool.map over samples of prediction.py:150-163):
 * the ingest of psk_count_kmers_batch (n_threads host threads move file bytes into pinned memory, FASTA and
 * FASTQ are framed on the GPU) with one dictionary kernel per sample and one read-back.  counts_out[n][n_dict].
 * _files: paths of UNCOMPRESSED files of sizes[i] bytes, read by the framing threads.  Any dictionary size: up to
 * 2048 words the table lives in LDS, beyond that in global memory (`--n_kmers 0` models). */
int psk_count_dict_batch(psk_ctx *ctx, int n, const uint8_t *const *bytes, const size_t *lens, int k,
                         const uint64_t *dict_words, uint64_t n_dict, uint32_t *counts_out, int n_threads);
int psk_count_dict_files(psk_ctx *ctx, int n, const char *const *paths, const size_t *sizes, int k,
                         const uint64_t *dict_words, uint64_t n_dict, uint32_t *counts_out, int n_threads);

/* ---- f2: MinHash sketch for the population-structure weights ---------------------------------
 * Replaces `mash sketch -r <addr> -o K-mer_lists/<name>` (Samples.get_mash_sketches,
 * modeling.py:386-390; bundled binary bin/mash 2.2): the `sketch_size` smallest distinct
 * MurmurHash3_x64_128(seed) hashes of the sample's canonical k-mers, ascending -- exactly the hash
 * list `mash info -d` prints.  Mash's defaults are k = 21, sketch_size = 1000, seed = 42.
 * hashes_out must hold sketch_size entries; *n_out receives how many were written.
 */
int psk_minhash_sketch(psk_ctx *ctx, const uint8_t *bytes, size_t len, int k, int sketch_size, uint32_t seed,
                       uint64_t *hashes_out, uint64_t *n_out);

/* Pairwise comparison of n bottom-s sketches (was `mash dist reference.msh reference.msh`,
 * Samples.get_mash_distances, modeling.py:411-421): for every pair the number of shared hashes and the
 * denominator of Mash's Jaccard estimate (the merge stops after sketch_size distinct union hashes).
 *   sketches[n][sketch_size]  ascending distinct hashes, rows padded; lens[n] = valid entries per row
 *   common_out[n][n], denom_out[n][n]  symmetric; the host turns them into Mash distances
 */
int psk_mash_pairs(psk_ctx *ctx, const uint64_t *sketches, const uint32_t *lens, int n, int sketch_size,
                   uint32_t *common_out, uint32_t *denom_out);

/* Neighbour joining of an n x n distance matrix (was Bio.Phylo.TreeConstruction.DistanceTreeConstructor.nj in
 * Samples.get_weights, modeling.py:447-458): the n - 2 joins in order.  Join t merges the clades at POSITIONS
 * mi_out[t] and mj_out[t] of the current clade list (the joined clade takes position mj, position mi is deleted),
 * with branch lengths d1_out[t] (clade mi) and d2_out[t] (clade mj); last_out = the distance left between the final
 * two clades.  Same arithmetic, scan order and tie-breaking as the library's scalar loops.  3 <= n <= 4096.
 */
int psk_nj_merges(psk_ctx *ctx, const double *dist, int n, int32_t *mi_out, int32_t *mj_out, double *d1_out,
                  double *d2_out, double *last_out);

/* ---- e: multi-GPU collectives (RCCL over xGMI, bound directly; librccl.so is opened on first use) ------------
 * The reference has no multi-process path (its parallelism is Pool(num_threads) over text chunks,
 * modeling.py:335-342, :660-675); these calls carry the three exchanges of the range-sharded path: the
 * all-reduce of the slab union sizes (the global Bonferroni denominator of modeling.py:644/:795), the
 * all-gather of the scan survivors and the all-to-all of list ranges (multi-GPU ingest).  One communicator per
 * context, on its own HIP stream.
 *   psk_comm_unique_id   rank 0: writes the 128-byte id of ncclGetUniqueId (returns its length); the caller's
 *                        rendezvous (a file, a socket) carries it to the other ranks
 *   psk_comm_init        every rank, same id: ncclCommInitRank on the context's GPU
 *   psk_comm_size        ncclCommCount of the communicator (>= 1; < 0 on error): what a multi-GPU measurement quotes
 *                        as proof that its collectives ran on RCCL with every rank joined
 *   psk_comm_allreduce   in place on `count` host values; dtype 0 = u64, 1 = f64; op 0 = sum, 1 = max
 *   psk_comm_allgather_host     recv[world][bytes] <- every rank's send[bytes] (host buffers, waited for)
 *   psk_comm_allgather_device   the same for DEVICE buffers, queued on the communicator's stream, not waited for
 *   psk_comm_alltoallv_device   send_counts[d] elements (elem_bytes 4 or 8) to rank d, back to back in send_dev;
 *                               recv_counts[s] elements from rank s, back to back in recv_dev; waited for
 *   psk_comm_stream      the communicator's hipStream_t (hand it to psk_export_survivors_async so that the
 *                        all-gather queued next is ordered behind the export on the device)
 *   psk_comm_sync        waits for everything queued on that stream
 */
int psk_device_count(void);
int psk_comm_unique_id(psk_ctx *ctx, uint8_t *id_out, int cap);
int psk_comm_init(psk_ctx *ctx, const uint8_t *id, int id_len, int rank, int world);
int psk_comm_size(psk_ctx *ctx);
int psk_comm_free(psk_ctx *ctx);
void *psk_comm_stream(psk_ctx *ctx);
int psk_comm_sync(psk_ctx *ctx);
int psk_comm_allreduce(psk_ctx *ctx, void *vals, int count, int dtype, int op);
int psk_comm_allgather_host(psk_ctx *ctx, const void *send, void *recv, uint64_t bytes);
int psk_comm_allgather_device(psk_ctx *ctx, const void *send_dev, void *recv_dev, uint64_t bytes);
int psk_comm_alltoallv_device(psk_ctx *ctx, const void *send_dev, const uint64_t *send_counts, void *recv_dev,
                              const uint64_t *recv_counts, int elem_bytes);
/* Plain device buffers for the callers of the exchanges (send / receive buffers live outside the context), and
 * waited-for copies: kind 0 host -> device, 1 device -> host, 2 device -> device; on_comm_stream != 0 orders the
 * copy behind the collectives queued on the communicator's stream. */
int psk_dev_alloc(psk_ctx *ctx, uint64_t bytes, void **out);
int psk_dev_free(psk_ctx *ctx, void *p);
int psk_dev_copy(psk_ctx *ctx, void *dst, const void *src, uint64_t bytes, int kind, int on_comm_stream);

/* ---- helpers shared with the host side ------------------------------------------------------ */
/* Host-only: the cleaned sequence stream the tokeniser hands to the GPU (bases kept, window
 * breaks collapsed to '\n', everything else dropped).  Returns the length written (<= len), or
 * a negative code.  Exposed for tests of the tokeniser contract. */
int64_t psk_frame_sequence(const uint8_t *bytes, size_t len, uint8_t *out, size_t out_cap);
/* The same stream as the GPU framing kernels produce it (the batch counters frame FASTA and FASTQ on the device,
 * csrc/frame_gpu.hip; r06: FASTQ that is not four lines per record too -- the scan of line kinds): identical except that
 * runs of window breaks are not collapsed.  Returns the length written.  For tests of the tokeniser contract. */
int64_t psk_frame_sequence_gpu(psk_ctx *ctx, const uint8_t *bytes, size_t len, uint8_t *out, size_t out_cap);

#ifdef __cplusplus
}
#endif
#endif /* PSK_H */
