"""`phenotypeseeker modeling` on the MI355X engine.

Host-side mirror of the reference's modeling module for the hot path (names, argument meaning
and output files follow /root/reference/PhenotypeSeeker/modeling.py; line numbers below cite
it).  The four GenomeTester4 call sites and the per-k-mer Python loop are replaced by libpsk.so
calls (phenotypeseeker_amd.engine); everything here is orchestration and formatting:

    Input.get_input_data / Input.Input_args      data.pheno parsing and option plumbing (:74-264)
    Samples.get_kmer_lists                       psk_count_kmers            (was glistmaker, :303-315)
    Samples.get_feature_vector + map_samples     psk_build_presence         (was glistcompare/glistquery, :317-380)
    phenotypes.test_kmers_association_with_phenotype   psk_chi2_scan / psk_ttest_scan (:659-858)
    phenotypes.get_ML_df                         ordering, TSV/CSV writers  (:1112-1145)
    phenotypes.machine_learning_modelling        GPU grid search + reports  (:860-988)

A single process drives the GPU (no fork pools: the HIP runtime is not fork-safe).
"""
import ctypes
import math
import os
import sys
import threading
import time
from collections import OrderedDict

import numpy as np

from . import dist as _dist
from . import formats, metrics, _stats
from .engine import PskContext
from . import _lib
from ._lib import PSK_EGZIP, PskError
from .model import GridSearch, L1LogisticRegression, L2LogisticRegression, LassoRegression, RidgeRegression

RED_BANNER = "\x1b[1;1;101m%s\x1b[0m\n"
GREEN = "\x1b[1;32m%s\x1b[0m"
YELLOW = "\x1b[1;33m%s\x1b[0m\n"


def _err(text):
    sys.stderr.write(text)
    sys.stderr.flush()


def timer(f):
    """Appends 'Func <f> took <s> secs' to ./log.txt like the reference's decorator (:54-61)."""
    def wrapper(*args):
        start = time.time()
        f(*args)
        with open("log.txt", "a") as log:
            log.write("Func %s took %s secs\n" % (f, time.time() - start))
    return wrapper


class Samples:
    """One row of data.pheno (:266-301)."""
    no_samples = 0
    phenotypes = []          # header names
    take_logs = None
    kmer_length = None
    cutoff = None
    min_samples = None
    max_samples = None
    kmerDB = None
    use_weights = False
    tree = None

    def __init__(self, name, address, phenotypes, weight=1):
        self.name = name
        self.address = address
        self.phenotypes = phenotypes
        self.weight = weight
        Samples.no_samples += 1

    @classmethod
    def from_inputfile(cls, line):
        fields = line.split()
        name, address, values = fields[0], fields[1], fields[2:]
        if any(v not in ("0", "1", "NA") for v in values):
            phenotypes.pred_scale = "continuous"
        return cls(name, address, dict(zip(cls.phenotypes, values)))

    # ---- k-mer plane -----------------------------------------------------------------------
    def get_kmer_lists(self, ctx, index):
        """was: glistmaker <address> -o K-mer_lists/<name>_0 -w <k> -c <cutoff>  (:303-315).
        As with the bundled glistmaker 4.2.3, the cut-off does not change the list."""
        data = formats.read_sequence_file(self.address)
        self.n_unique, self.n_total = ctx.count_kmers(index, data)
        if Samples.use_weights:  # was: mash sketch -r <address> -o K-mer_lists/<name>  (:386-390)
            self.sketch = ctx.minhash_sketch(data).tolist()
        stderr_print.currentSampleNum += 1
        stderr_print.print_progress("lists generated.")

    @classmethod
    def get_kmer_lists_batched(cls, ctx, samples, n_threads, chunk=None):
        """All samples through the batch counter: the files are read by the library's own threads
        (psk_count_kmers_files), .gz ones as they are -- the compressed image crosses PCIe and is inflated on the device
        (csrc/gz_inflate.hip; r05).  With PSK_NO_GPU_GZ=1 compressed files are read and inflated by a thread pool here and
        handed over in memory (psk_count_kmers_batch), as until r04."""
        from concurrent.futures import ThreadPoolExecutor
        if chunk is None:  # about half a gigabyte of file images per call, two calls' worth in memory
            try:
                biggest = max(os.stat(s.address).st_size for s in samples[:: max(1, len(samples) // 16)])
            except OSError:
                biggest = 1 << 29
            chunk = int(min(64, max(1, (1 << 29) // max(biggest, 1))))
        sketch = (21, 1000, 42) if cls.use_weights else None  # was: mash sketch -r <address> (:386-390): k=21, s=1000, seed 42
        # Uncompressed inputs are read by the library's own threads (psk_count_kmers_files): no file image passes through
        # Python.  Compressed ones (gzip by its magic bytes -- two bytes per file, probed by the thread pool; the
        # suffix alone is not trusted) are inflated here and handed over in memory.  The route is chosen PER CHUNK, so a
        # compressed file in a later chunk never sends the samples already counted through the context again (ADVICE
        # r02: the old whole-set fallback recounted from sample 0 and held their list memory twice).  The library
        # checks the magic bytes too and answers PSK_EGZIP: a file that changed under the probe still ends up inflated.
        # chunk boundaries ramp up (8, 16, 32, ...): the first read is short, later calls amortise their set-up
        # the inflating pool may use every thread `-nt` grants (zlib releases the GIL: a compressed read set inflates at ~0.3 GB/s
        # per thread, twenty times below what the GPU ingests -- r04's cap of 8 was the framing threads', which these are not)
        inflaters = max(n_threads, min(int(getattr(Input, "num_threads", n_threads) or n_threads), os.cpu_count() or n_threads))
        with ThreadPoolExecutor(max_workers=inflaters) as pool:
            # (two bytes per file, read here: through the pool the 1,024 futures of a 1,024-genome run cost 45 ms, the reads 10)
            host_inflate = bool(os.environ.get("PSK_NO_GPU_GZ"))
            zipped = [host_inflate and (s.address.endswith(".gz") or formats.is_gzip(s.address)) for s in samples]
            # (plain files never sit in this process's memory -- the library streams them through its pinned ring --, so
            # their calls grow to 512 samples: a call ends with its pipeline drained, ~3 ms each at 64 samples per call, 60 ms
            # of a 1,024-genome ingest (r04, PSK_TRACE); compressed ones keep the half gigabyte per call)
            cap = chunk if any(zipped) else max(chunk, 512)
            bounds, lo, step = [], 0, min(cap, 8)
            while lo < len(samples):
                bounds.append((lo, min(len(samples), lo + step)))
                lo += step
                # (compressed chunks double: each is read and inflated while the one before is counted; plain files go from
                # the short first call -- the process's cold start -- straight to full calls: 2,048 genomes in 5 calls, not 10)
                step = min(cap, step * 2) if any(zipped) else cap

            def submit(b):   # the compressed chunks are read (and inflated) one chunk ahead of the counting
                if b is None or not any(zipped[b[0]:b[1]]):
                    return None
                return [pool.submit(formats.read_sequence_file, s.address) for s in samples[b[0]:b[1]]]
            pending = submit(bounds[0]) if bounds else None
            for bi, (lo, hi) in enumerate(bounds):
                part = samples[lo:hi]
                reads, pending = pending, submit(bounds[bi + 1] if bi + 1 < len(bounds) else None)
                if reads is None:
                    try:
                        cls._record_lists(part, ctx.count_kmers_files(lo, [s.address for s in part], n_threads, sketch=sketch))
                        continue
                    except PskError as exc:
                        if exc.code != PSK_EGZIP:
                            raise
                    reads = [pool.submit(formats.read_sequence_file, s.address) for s in part]
                cls._record_lists(part, ctx.count_kmers_batch(lo, [f.result() for f in reads], n_threads, sketch=sketch))

    @classmethod
    def _pilot_bounds(cls, group, k, cnt, counted):
        """Balanced slab bounds (dist.balanced_bounds) from up to four of the lists this rank has counted in `cnt`
        (whole word space): `counted` = [(local index, n_unique)]."""
        pilot = [cnt.get_list(j, nu)[0] for j, nu in counted[:4]]
        return _dist.balanced_bounds(group, k, pilot)

    @classmethod
    def get_kmer_lists_sharded(cls, ctx, group, samples, n_threads, exchange):
        """Several ranks.  The word space is cut at the quantiles of a pilot of the lists, so that every rank gets
        the same share of the rows (the reference's chunks are balanced too: round-robin `split -n r/<nt>`,
        :335-342); `ctx` is begun with this rank's slab.
        exchange: every sample is counted on ONE rank (round robin, no slab filter, a second context on the
        same GPU) and the slab ranges of the sorted lists are exchanged GPU to GPU (dist.ListExchange) -- each rank
        ends up with the lists it would have counted itself with its slab filter, without every rank framing and
        scanning every file.  Otherwise every rank counts every sample and keeps its slab (after a pilot of a few
        samples, counted whole, has given the bounds).  With -w the sketches of a rank's samples are all-gathered
        (every rank computes the weights, the scans need them)."""
        W, r = group.world, group.rank
        k = int(cls.kmer_length)
        n = len(samples)
        if not exchange:
            first = [i for i in range(min(n, max(W, 4))) if _dist.owner_of(i, W) == r]
            with PskContext(group.device) as cnt:
                cnt.begin(k, max(len(first), 1))
                sketch_was, cls.use_weights = cls.use_weights, False       # the pilot needs no sketches
                progress_was = stderr_print.currentSampleNum
                try:
                    if first:
                        cls.get_kmer_lists_batched(cnt, [samples[i] for i in first], n_threads)
                finally:
                    cls.use_weights = sketch_was
                    stderr_print.currentSampleNum = progress_was
                bounds = cls._pilot_bounds(group, k, cnt, [(j, samples[i].n_unique) for j, i in enumerate(first)])
            ctx.begin(k, n, bounds[r], bounds[r + 1])
            cls.get_kmer_lists_batched(ctx, samples, n_threads)
            return bounds
        own = [s for i, s in enumerate(samples) if _dist.owner_of(i, W) == r]
        cnt = PskContext(group.device)
        try:
            cnt.begin(k, max(len(own), 1))
            if own:
                cls.get_kmer_lists_batched(cnt, own, n_threads)
            bounds = cls._pilot_bounds(group, k, cnt, [(j, s.n_unique) for j, s in enumerate(own)])
            ctx.begin(k, n, bounds[r], bounds[r + 1])
            _dist.ListExchange(group, k, bounds).run(cnt, ctx, n, [s.n_total for s in own])
        finally:
            cnt.close()
        if cls.use_weights:
            mine = [np.asarray(s.sketch, dtype=np.uint64) for s in own]
            head = np.array([len(m) for m in mine], dtype=np.uint64)
            blobs = group.allgather_bytes(np.uint64(len(mine)).tobytes() + head.tobytes() + b"".join(m.tobytes() for m in mine))
            for src, blob in enumerate(blobs):
                cntb = int(np.frombuffer(blob, dtype=np.uint64, count=1)[0])
                lens = np.frombuffer(blob, dtype=np.uint64, count=cntb, offset=8).astype(np.int64)
                off = 8 + 8 * cntb
                theirs = [s for i, s in enumerate(samples) if _dist.owner_of(i, W) == src]
                for s, ln in zip(theirs, lens):
                    s.sketch = np.frombuffer(blob, dtype=np.uint64, count=int(ln), offset=off).tolist()
                    off += 8 * int(ln)
        return bounds

    @classmethod
    def _record_lists(cls, part, res):
        """(n_unique, n_total[, sketches]) of one counting call -> the Samples objects, with the progress line."""
        nu, nt = res[0], res[1]
        sk = res[2] if len(res) > 2 else None
        for j, s in enumerate(part):
            s.n_unique, s.n_total = nu[j], nt[j]
            if sk is not None:
                s.sketch = sk[j].tolist()
            stderr_print.currentSampleNum += 1
            stderr_print.print_progress("lists generated.")

    @classmethod
    def get_weights(cls, ctx, write_files=True):
        """was: mash paste / mash dist -> NJ tree -> GSC weights (:392-503); the pairwise sketch comparison runs
        on the GPU (psk_mash_pairs).  Like the reference the run leaves `distances.mat` (:415-428) and
        `tree_newick.txt` (:455-458) in the working directory (rank 0 writes them)."""
        from . import weights as _w
        names = list(Input.samples.keys())
        w, cls.tree = _w.weights_from_sketches(names, {n: Input.samples[n].sketch for n in names}, ctx=ctx,
                                               files_dir="." if write_files else None)
        for name, value in w.items():
            Input.samples[name].weight = value

    @classmethod
    def get_feature_vector(cls, ctx):
        """was: the glistcompare -u tree, optionally intersected with --kmerDB (:350-372)."""
        m = ctx.build_presence()
        if cls.kmerDB:
            with PskContext(ctx.device) as db_ctx:
                db_ctx.begin(int(cls.kmer_length), 1)
                nu, _ = db_ctx.count_kmers(0, formats.read_sequence_file(cls.kmerDB))
                db_words, _ = db_ctx.get_list(0, nu)
            m = ctx.intersect_db(db_words)
        return m


class stderr_print:
    """Progress banners on stderr (:507-537) without the lock-guarded counters."""
    currentSampleNum = 0

    def __init__(self, data):
        _err("\r\x1b[K\x1b[1;32m" + str(data) + "\x1b[0m")

    @classmethod
    def print_progress(cls, txt):
        if cls.currentSampleNum != Samples.no_samples:
            cls("\t\x1b[1;91m%d\x1b[1;32m of %d %s" % (cls.currentSampleNum, Samples.no_samples, txt))
        else:
            cls("\t%d of %d %s" % (cls.currentSampleNum, Samples.no_samples, txt))


class Input:
    samples = OrderedDict()
    phenotypes_to_analyse = OrderedDict()
    jump_to = None
    num_threads = 8

    @classmethod
    def reset(cls):
        cls.samples = OrderedDict()
        cls.phenotypes_to_analyse = OrderedDict()
        cls.jump_to = None
        Samples.no_samples = 0
        Samples.phenotypes = []
        phenotypes.pred_scale = "binary"
        phenotypes.no_results = []
        phenotypes.model_package = {}
        phenotypes._exchange = None
        stderr_print.currentSampleNum = 0
        Samples.use_weights = False
        Samples.tree = None

    @classmethod
    def get_input_data(cls, inputfilename, take_logs, mpheno):
        """data.pheno: header 'ID Addresses pheno...', then 'name address value...' (:74-97)."""
        Samples.take_logs = take_logs
        with open(inputfilename) as fh:
            header = fh.readline().split()
            Samples.phenotypes = header[2:]
            for ph in Samples.phenotypes:
                try:
                    float(ph)
                except ValueError:
                    continue
                _err(YELLOW % "Warning! It seems that the input file is missing header row!")
                break
            for line in fh:
                if line.strip():
                    cls.samples[line.split()[0]] = Samples.from_inputfile(line)
        n_ph = len(Samples.phenotypes)
        columns = range(n_ph) if not mpheno else [m - 1 for m in mpheno]
        for c in columns:
            cls.phenotypes_to_analyse[Samples.phenotypes[c]] = phenotypes(Samples.phenotypes[c])
        cls._set_phenotype_values(take_logs)

    @classmethod
    def _set_phenotype_values(cls, take_logs):
        """int() / float() (optionally log2); anything unparsable stays as its string, 'NA' in
        practice, and lowers the phenotype's sample count (:109-126)."""
        for sample in cls.samples.values():
            for ph in cls.phenotypes_to_analyse.values():
                raw = sample.phenotypes[ph.name]
                try:
                    if phenotypes.pred_scale == "continuous":
                        v = float(raw)
                        if take_logs:
                            v = math.log(v, 2)
                    else:
                        v = int(raw)
                    sample.phenotypes[ph.name] = v
                except (ValueError, TypeError):
                    ph.no_samples -= 1

    @classmethod
    def pop_phenos_out_of_kmers(cls):
        for name in phenotypes.no_results:
            cls.phenotypes_to_analyse.pop(name, None)
        if not cls.phenotypes_to_analyse:
            _err(YELLOW % "There are no k-mers left for modelling for any phenotype.")
            _err(YELLOW % "Exiting PhenotypeSeeker")
            _err("\n" + RED_BANNER % "######          PhenotypeSeeker modeling finished          ######")
            raise SystemExit()

    @classmethod
    def Input_args(cls, alphas, alpha_min, alpha_max, n_alphas, gammas, gamma_min, gamma_max, n_gammas,
                   min_samples, max_samples, kmer_length, cutoff, num_threads, pvalue_cutoff, kmer_limit,
                   binary_classifier, regressor, penalty, max_iter, tol, l1_ratio, n_splits_cv_outer, kernel,
                   n_iter, n_splits_cv_inner, testset_size, train_on_whole, logreg_solver, jump_to, pca,
                   real_counts, omit_B, kmerDB):
        """Same positional signature as the reference (:141-183).  Options that select estimators
        outside the hot path (SVM/RF/DT/NB, L2/elastic net, saga, PCA) are rejected here."""
        if alphas is None:
            phenotypes.alphas = np.logspace(math.log10(alpha_min), math.log10(alpha_max), num=n_alphas)
        else:
            phenotypes.alphas = np.array(alphas)
        mn, mx = int(min_samples), int(max_samples)
        Samples.min_samples = mn if mn != 0 else 2
        Samples.max_samples = mx if mx != 0 else Samples.no_samples - 2
        Samples.kmer_length = str(kmer_length)
        Samples.cutoff = cutoff
        Samples.kmerDB = kmerDB
        cls.num_threads = num_threads
        cls.jump_to = jump_to
        phenotypes.pvalue_cutoff = pvalue_cutoff
        phenotypes.kmer_limit = kmer_limit
        phenotypes.penalty = penalty.upper()
        phenotypes.max_iter = max_iter
        phenotypes.tol = tol
        phenotypes.n_splits_cv_outer = n_splits_cv_outer
        phenotypes.n_splits_cv_inner = n_splits_cv_inner
        phenotypes.testset_size = testset_size
        phenotypes.train_on_whole = train_on_whole
        phenotypes.real_counts = real_counts
        phenotypes.omit_B = omit_B
        phenotypes.pca = pca
        if phenotypes.pred_scale == "continuous":
            if regressor != "lin":
                raise SystemExit("Only the linear (Lasso) regressor runs on the GPU engine, got %r." % regressor)
            phenotypes.model_name_long, phenotypes.model_name_short = "linear regression", "linreg"
        else:
            if binary_classifier != "log":
                raise SystemExit("Only the logistic-regression classifier runs on the GPU engine, got %r "
                                 "(SVM/RF/DT/NB are outside the accelerated path)." % binary_classifier)
            phenotypes.model_name_long, phenotypes.model_name_short = "logistic regression", "log_reg"
            # get_logreg_solver (:245-264): L1 -> liblinear (saga's objective leaves the intercept free and
            # is not implemented); L2 -> lbfgs by default, all five names share one strictly convex optimum
            if phenotypes.penalty == "L1":
                if logreg_solver not in (None, "liblinear"):
                    raise SystemExit("Logistic Regression with L1 penalty on the GPU engine implements the "
                                     "liblinear objective only, got {}.".format(logreg_solver))
                phenotypes.logreg_solver = "liblinear"
            elif phenotypes.penalty == "L2":
                if logreg_solver is None:
                    logreg_solver = "lbfgs"
                if logreg_solver not in ("liblinear", "newton-cg", "lbfgs", "sag", "saga"):
                    raise SystemExit("Logistic Regression with L2 penalty supports only "
                                     "solvers in ['liblinear', 'newton-cg', 'lbfgs', 'sag', 'saga'], "
                                     "got {}.".format(logreg_solver))
                phenotypes.logreg_solver = logreg_solver
        if phenotypes.penalty not in ("L1", "L2"):
            # L1+L2 cannot complete in the reference either: GridSearchCV is handed {'C': ...} for an
            # SGDClassifier (:1020-1024, :1045-1047) and fit_model skips the ElasticNet fit (:1211-1214)
            raise SystemExit("Only the L1 and L2 penalties run on the GPU engine, got %r." % penalty)
        if pca:
            raise SystemExit("--pca is outside the accelerated path.")


class phenotypes:
    pred_scale = "binary"
    real_counts = False
    model_name_long = None
    model_name_short = None
    no_kmers_to_analyse = 0
    pvalue_cutoff = None
    kmer_limit = None
    omit_B = None
    penalty = None
    logreg_solver = None
    max_iter = None
    tol = None
    alphas = None
    n_splits_cv_outer = None
    n_splits_cv_inner = None
    testset_size = None
    train_on_whole = None
    pca = None
    no_results = []
    model_package = {}
    _exchange = None

    def __init__(self, name):
        self.name = name
        self.no_samples = Samples.no_samples
        self.rows = None      # surviving k-mers of the scan (dict of arrays + kmer strings)
        self.ML = None        # selected design matrix etc.
        self.model_fitted = None
        self._scan_launched = False

    # ---- association scan ----------------------------------------------------------------------
    @classmethod
    def kmer_testing_setup(cls, n_union_global):
        """The Bonferroni denominator is the size of the WHOLE union (:640-644)."""
        label = "Welch t-tests" if cls.pred_scale == "continuous" else "chi-square tests"
        _err("\n" + GREEN % ("Conducting the k-mer specific %s:" % label) + "\n")
        cls.no_kmers_to_analyse = int(n_union_global)

    def _phenotype_vectors(self):
        samples = list(Input.samples.values())
        weights = np.array([float(s.weight) for s in samples])
        unit = all(s.weight == 1 for s in samples)
        if self.pred_scale == "binary":
            ph = np.array([s.phenotypes[self.name] if s.phenotypes[self.name] in (0, 1) else -1 for s in samples],
                          dtype=np.int8)
            return ph, None, (None if unit else weights)
        vals = np.zeros(len(samples))
        valid = np.zeros(len(samples), dtype=np.uint8)
        for i, s in enumerate(samples):
            v = s.phenotypes[self.name]
            if v != "NA" and not isinstance(v, str):
                vals[i], valid[i] = float(v), 1
        return vals, valid, (None if unit else weights)

    def launch_scan(self, ctx):
        """Binary phenotypes: queues this phenotype's scan (psk_chi2_scan_begin) without waiting, so that it streams
        the matrix while the host formats the previous phenotype's survivors."""
        if self.pred_scale != "binary" or self._scan_launched:
            return
        a, _, w = self._phenotype_vectors()
        ctx.chi2_scan_begin(a, w, Samples.min_samples, Samples.max_samples, self.pvalue_cutoff, self.omit_B,
                            self.no_kmers_to_analyse)
        self._scan_launched = True

    def test_kmers_association_with_phenotype(self, ctx, group, nxt=None):
        """was: Pool.map(get_kmers_tested) over text chunks (:659-714).  One scan kernel launch
        per phenotype; with several GPUs every rank scans its slab and the survivors are
        all-gathered in slab order.  `nxt`: the phenotype scanned next, launched as soon as this one's
        scan has ended (its results live in the context's other result set)."""
        start = time.time()
        n = Samples.no_samples
        a, b, w = self._phenotype_vectors()
        if self.pred_scale == "binary":
            self.launch_scan(ctx)
            npass = ctx.scan_end()
            self._scan_launched = False
            if nxt is not None and group.world == 1:
                nxt.launch_scan(ctx)
        else:
            npass = ctx.ttest_scan(a, b, w, Samples.min_samples, Samples.max_samples, self.pvalue_cutoff,
                                   self.no_kmers_to_analyse)
        counts = None
        if group.world > 1:
            # survivors of every slab, gathered GPU-to-GPU and merged in ascending word order
            if phenotypes._exchange is None:
                phenotypes._exchange = _dist.SurvivorExchange(group, ctx.presence_shape()[1])
            local = ctx.get_results(npass) if self.real_counts else None
            t_g = time.time()
            res, bits = phenotypes._exchange.gather(ctx)
            if Phases.current is not None:
                Phases.current.move("scan", "survivor all-gather", time.time() - t_g)
            if self.real_counts:
                mine = np.stack([ctx.lookup_counts(i, local["word"]) for i in range(n)], axis=1) if npass else \
                    np.zeros((0, n), np.uint32)
                cp = group.allgather_bytes(np.ascontiguousarray(local["word"]).tobytes() +
                                           np.ascontiguousarray(mine, dtype=np.uint32).tobytes())
                cw, cc = [], []
                for blob in cp:
                    m = len(blob) // (8 + 4 * n)
                    cw.append(np.frombuffer(blob, dtype=np.uint64, count=m))
                    cc.append(np.frombuffer(blob, dtype=np.uint32, offset=8 * m).reshape(m, n))
                cw, cc = np.concatenate(cw), np.concatenate(cc)
                counts = cc[np.argsort(cw, kind="stable")]
        else:
            res = ctx.get_results(npass)
            bits = ctx.get_rows(res["row"])
            if self.real_counts and npass:
                counts = np.stack([ctx.lookup_counts(i, res["word"]) for i in range(n)], axis=1)
        # (the k-mer text and the 0/1 columns of the rows are made in get_ML_df, for the rows that are selected only: a
        # continuous phenotype leaves 10^5 rows of which 1,000 go on)
        self.rows = {"word": np.ascontiguousarray(res["word"], dtype=np.uint64), "bits": bits, "stat": res["stat"], "p": res["p"],
                     "mean_x": res["mean_x"], "mean_y": res["mean_y"], "n_with": res["n_with"],
                     "vector": counts.astype(np.int64) if counts is not None else None}
        _err("\t%s: 100%% tests conducted.\n" % self.name)
        if len(self.rows["word"]) == 0:
            self.no_results.append(self.name)
        with open("log.txt", "a") as log:
            log.write("Func test_kmers_association_with_phenotype took %s secs (scan kernel %.3f ms)\n"
                      % (time.time() - start, ctx.last_scan_ms()))

    # ---- selection (:1112-1145) ------------------------------------------------------------------
    def get_ML_df(self):
        start = time.time()
        names = list(Input.samples.keys())
        samples = list(Input.samples.values())
        if Input.jump_to == "modelling":
            self._load_MLdf()
        else:
            r = self.rows
            binary = self.pred_scale == "binary"
            valid = [s.phenotypes[self.name] != "NA" and not isinstance(s.phenotypes[self.name], str) for s in samples]
            if binary:
                head, stem = "k-mer\tchi2\tp-value\tnum_samples_w_kmer\tsamples_with_kmer", "chi2"
            else:
                head, stem = ("k-mer\tt-test\tp-value\t+_group_mean\t-_group_mean\tnum_samples_w_kmer\t"
                              "samples_with_kmer"), "t-test"
            # The tables are formatted and written by libpsk (psk_write_result_tables: host threads; a continuous phenotype
            # leaves 10^5 rows with 3 x 10^7 sample names -- 0.6 s of Python string work in r03).  It also returns the order
            # of the lines: the reference sorts the columns by the p-value STRINGS (:1128); ties, which numpy's unstable
            # sort leaves in an arbitrary order there, are broken by k-mer text
            words = r.get("word")
            if words is None:     # (rows handed over as text and 0/1 columns: tests/test_host_modeling.py)
                words = np.array([formats.kmer_to_word(km) for km in r["kmer"]], dtype=np.uint64)
            n_rows = len(words)
            bits = r.get("bits")
            if bits is None:
                pres = np.asarray(r["presence"], dtype=np.uint8).reshape(n_rows, len(names))
                wpr = (len(names) + 63) // 64
                padded = np.zeros((n_rows, wpr * 64), np.uint8)
                padded[:, :len(names)] = pres != 0
                bits = np.packbits(padded, axis=1, bitorder="little").view("<u8")
            # (no surviving row: numpy cannot infer -1 for an empty array; the table is then the header line alone)
            bits = np.ascontiguousarray(bits, dtype=np.uint64).reshape(n_rows, -1 if n_rows else (len(names) + 63) // 64)
            enc = [nm.encode() for nm in names]
            off = np.zeros(len(enc) + 1, dtype=np.int64)
            off[1:] = np.cumsum([len(e) for e in enc])
            order_arr = np.zeros(n_rows, dtype=np.int64)
            arr = {key: np.ascontiguousarray(r[key], dtype=np.float64) for key in ("stat", "p", "mean_x", "mean_y")}
            n_with = np.ascontiguousarray(r["n_with"], dtype=np.int32)
            words = np.ascontiguousarray(words, dtype=np.uint64)
            vmask = np.array(valid, dtype=np.uint8)
            top_path = ("%s_results_%s_top%s.tsv" % (stem, self.name, self.kmer_limit)).encode() if self.kmer_limit else None
            lib = _lib.load()
            vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
            rc = lib.psk_write_result_tables(None, ("%s_results_%s.tsv" % (stem, self.name)).encode(), top_path,
                                             int(self.kmer_limit or 0), head.encode(), 0 if binary else 1, n_rows, vp(words),
                                             int(Samples.kmer_length), vp(arr["stat"]), vp(arr["p"]), vp(arr["mean_x"]),
                                             vp(arr["mean_y"]), vp(n_with), vp(bits), bits.shape[1] if n_rows else 1, len(names),
                                             vp(vmask), b"".join(enc), vp(off), vp(order_arr))
            if rc != 0:
                raise PskError("psk_write_result_tables failed (%d)" % rc, rc)
            order = order_arr.tolist()
            if self.kmer_limit:
                order = order[: self.kmer_limit]
            keep = [j for j in range(len(names)) if valid[j]]
            sel = np.array(order, dtype=np.int64)
            if r.get("vector") is not None:      # --real_counts: the counts of the selected k-mers
                vec = np.asarray(r["vector"])[sel]
            else:                                # bit i of a row = sample i (little-endian u64 words)
                vec = np.unpackbits(np.ascontiguousarray(bits[sel], dtype="<u8").view(np.uint8), axis=1, bitorder="little")[:, :len(names)]
            self.ML = {"kmers": formats.words_to_kmers(np.asarray(words)[sel], int(Samples.kmer_length)),
                       "index": [names[j] for j in keep],
                       "X": vec[:, keep].T.astype(np.int64) if order else np.zeros((len(keep), 0), np.int64),
                       "weights": [samples[j].weight for j in keep],
                       "phenotype": [samples[j].phenotypes[self.name] for j in keep]}
            self._write_MLdf()
        self.model_package["kmers"] = np.array(self.ML["kmers"], dtype=object)
        with open("log.txt", "a") as log:
            log.write("Func get_ML_df took %s secs\n" % (time.time() - start))

    def _write_MLdf(self):
        """<pheno>_MLdf.csv as pandas wrote it: samples x (k-mers + weights + phenotype) (:1144)."""
        ml = self.ML
        with open(self.name + "_MLdf.csv", "w") as f:
            f.write(",".join([""] + ml["kmers"] + ["weights", "phenotype"]) + "\n")
            Xa = np.asarray(ml["X"]).astype(np.int64, copy=False)
            # 0/1 columns (anything of one digit): ",d,d,...": one byte matrix for all rows instead of a str() and a join per
            # cell (2,048 x 1,000: 74 -> 6 ms); --real_counts with counts of several digits keeps the general writer
            digits = Xa.size > 0 and Xa.ndim == 2 and int(Xa.min()) >= 0 and int(Xa.max()) <= 9
            if digits:
                cells = np.empty((Xa.shape[0], 2 * Xa.shape[1]), dtype=np.uint8)
                cells[:, 0::2] = ord(",")
                cells[:, 1::2] = Xa + ord("0")
            else:
                X = Xa.tolist()   # python ints: str() of numpy scalars is 5x slower
            for i, name in enumerate(ml["index"]):
                w, p = ml["weights"][i], ml["phenotype"][i]
                tail = [repr(w) if isinstance(w, float) else str(w), repr(p) if isinstance(p, float) else str(p)]
                if digits:
                    f.write(name + cells[i].tobytes().decode("ascii") + "," + ",".join(tail) + "\n")
                else:
                    f.write(",".join([name] + list(map(str, X[i])) + tail) + "\n")

    def _load_MLdf(self):
        with open(self.name + "_MLdf.csv") as f:
            head = f.readline().rstrip("\n").split(",")
            kmers = head[1:-2]
            idx, X, wts, ph = [], [], [], []
            for line in f:
                c = line.rstrip("\n").split(",")
                idx.append(c[0])
                X.append([int(float(v)) for v in c[1:-2]])
                wts.append(float(c[-2]) if "." in c[-2] else int(c[-2]))
                ph.append(float(c[-1]) if self.pred_scale == "continuous" else int(float(c[-1])))
        self.ML = {"kmers": kmers, "index": idx, "X": np.array(X, dtype=np.int64).reshape(len(idx), len(kmers)),
                   "weights": wts, "phenotype": ph}

    # ---- model (:860-988) -------------------------------------------------------------------------
    def assert_n_splits_cv_outer(self, n_splits_cv_outer, y):
        """(:1498-1510)"""
        if self.pred_scale == "continuous" and n_splits_cv_outer > self.no_samples // 2:
            self.n_splits_cv_outer = self.no_samples // 2
            _err(YELLOW % ("Warning! The 'n_splits_cv_outer' parameter is too high to \n"
                           "leave the required 2 samples into test set for each split!"))
            _err(YELLOW % ("Setting number of train/test splits equal to " + str(self.n_splits_cv_outer) + "!") + "\n")
        elif self.pred_scale == "binary" and np.min(np.bincount(y)) < n_splits_cv_outer:
            self.n_splits_cv_outer = int(np.min(np.bincount(y)))
            _err(YELLOW % ("Setting number of train/test splits equal to minor phenotype count - "
                           + str(self.n_splits_cv_outer) + "!") + "\n")
        else:
            self.n_splits_cv_outer = n_splits_cv_outer

    def assert_n_splits_cv_inner(self, n_splits_cv_inner, y_all, y_train=None):
        """min(what the smallest training fold allows, requested) (:1512-1524)."""
        outer = getattr(self, "n_splits_cv_outer", None)
        if self.pred_scale == "continuous":
            least = self.no_samples - math.ceil(self.no_samples / outer) if outer else len(y_train)
        else:
            if outer:
                min_class = int(np.min(np.bincount(np.asarray(y_all, dtype=np.int64))))
                least = min_class - math.ceil(min_class / outer)
            else:
                least = int(np.min(np.bincount(np.asarray(y_train, dtype=np.int64))))
        self.n_splits_cv_inner = int(min(least, n_splits_cv_inner))

    def _new_estimator(self):
        if self.pred_scale == "binary":
            grid = [1.0 / a for a in self.alphas]
            if self.penalty == "L2":
                return L2LogisticRegression(tol=self.tol, max_iter=self.max_iter, solver=self.logreg_solver), "C", grid
            return L1LogisticRegression(tol=self.tol, max_iter=self.max_iter), "C", grid
        if self.penalty == "L2":
            return RidgeRegression(tol=self.tol, max_iter=self.max_iter), "alpha", [np.float64(a) for a in self.alphas]
        # (numpy scalars, as the reference's {'alpha': np.logspace(...)} grid holds them: the summary prints the params' repr)
        return LassoRegression(tol=self.tol, max_iter=self.max_iter), "alpha", [np.float64(a) for a in self.alphas]

    def _fit(self, ctx, X, y):
        est, pname, grid = self._new_estimator()
        self.model = est
        self.model_fitted = GridSearch(est, pname, grid, self.n_splits_cv_inner).fit(X, y, ctx)

    def machine_learning_modelling(self, ctx):
        _err("\x1b[1;32m\t" + self.name + ".\x1b[0m\n")
        self.get_ML_df()
        short = self.model_name_short
        summary = open("summary_of_%s_analysis_%s.txt" % (short, self.name), "w")
        coeff_path = "k-mers_and_coefficients_in_%s_model_%s.txt" % (short, self.name)
        X = self.ML["X"].astype(np.float64)
        index = list(self.ML["index"])
        binary = self.pred_scale == "binary"
        y = np.array(self.ML["phenotype"], dtype=np.int64 if binary else np.float64)
        self.n_splits_cv_outer = None
        m_train, m_test = _metric_store(), _metric_store()
        if phenotypes.n_splits_cv_outer:
            # -cv1: outer (Stratified)KFold, a grid search per training fold (:871-922)
            self.assert_n_splits_cv_outer(phenotypes.n_splits_cv_outer, y)
            self.assert_n_splits_cv_inner(phenotypes.n_splits_cv_inner, y)
            from . import cv as _cv
            folds = _cv.stratified_kfold(y, self.n_splits_cv_outer) if binary else _cv.kfold(len(y), self.n_splits_cv_outer)
            for f in range(self.n_splits_cv_outer):
                tr, te = np.nonzero(folds != f)[0], np.nonzero(folds == f)[0]
                self._fit(ctx, X[tr], y[tr])
                summary.write("\n##### Train/test split nr.%d: #####\n" % (f + 1))
                self._cross_validation_results(summary)
                summary.write("\nTraining set:\n")
                self._predict_report(summary, [index[i] for i in tr], X[tr], y[tr], m_train)
                summary.write("\nTest set:\n")
                self._predict_report(summary, [index[i] for i in te], X[te], y[te], m_test)
            if not self.train_on_whole:
                summary.write("\n### Outputting the last model to a model file! ###\n")
            summary.write("\nMean performance metrics over all train splits: \n\n")
            self._mean_report(summary, m_train)
            summary.write("\nMean performance metrics over all test splits: \n\n")
            self._mean_report(summary, m_test)
        elif self.testset_size:
            # -ts: one hold-out split, train_test_split(random_state=55, stratified for classes) (:924-951)
            from . import cv as _cv
            tr, te = _cv.train_test_split_indices(len(y), self.testset_size, y if binary else None, 55)
            self.assert_n_splits_cv_inner(phenotypes.n_splits_cv_inner, y, y[tr])
            self._fit(ctx, X[tr], y[tr])
            self._cross_validation_results(summary)
            summary.write("\nTraining set:\n")
            self._predict_report(summary, [index[i] for i in tr], X[tr], y[tr], m_train)
            summary.write("\nTest set:\n")
            self._predict_report(summary, [index[i] for i in te], X[te], y[te], m_test)
            if not self.train_on_whole:
                summary.write("\n### Outputting the model to a file! ###\n")
        split = bool(phenotypes.n_splits_cv_outer or self.testset_size)
        if not split or self.train_on_whole:
            # the default: train on everything (:953-973)
            if split:
                summary.write("\nThe final output model training on the whole dataset:\n")
            self.assert_n_splits_cv_inner(phenotypes.n_splits_cv_inner, y, y)
            self._fit(ctx, X, y)
            self._cross_validation_results(summary)
            self._predict_report(summary, index, X, y, None)
            if split:
                summary.write("\n### Outputting the last model trained on whole data to a model file! ###\n")
            else:
                summary.write("\n### Outputting the model to a model file! ###\n")
        package = dict(self.model_package)
        # The reference's .pkl holds scikit-learn objects and its loader is a plain joblib.load (prediction.py:124-129):
        # the fitted model is written as the equivalent GridSearchCV / LogisticRegression / Lasso / Ridge objects, so
        # that the file loads where this package is absent.  PSK_NATIVE_PKL=1 keeps the package's own
        # (scikit-learn-free) classes, which is also what is written when scikit-learn is missing.
        package.update({"pca": self.pca, "pred_scale": self.pred_scale})
        blob = None
        if os.environ.get("PSK_NATIVE_PKL") != "1":
            from . import skpickle
            shell = self.model_fitted.to_sklearn_shell()          # no scikit-learn import (skpickle.py)
            if shell is not None:
                blob = skpickle.dumps(dict(package, model=shell))
            else:                                                   # a scikit-learn version without a template: import it
                try:
                    package["model"] = self.model_fitted.to_sklearn()
                except ImportError:
                    package["model"] = self.model_fitted
        else:
            package["model"] = self.model_fitted
        with open("%s_model_%s.pkl" % (short, self.name), "wb") as fh:
            if blob is not None:
                fh.write(blob)
            else:
                import joblib
                joblib.dump(package, fh)
        self._write_model_coefficients(coeff_path)
        summary.close()

    def _cross_validation_results(self, out):
        """(:1219-1237)"""
        out.write("Parameters:\n%s\n\n" % self.model)
        out.write("Grid scores (%s) on development set: \n" %
                  ("R2 score" if self.pred_scale == "continuous" else "mean accuracy"))
        cvr = self.model_fitted.cv_results_
        for mean, std, param in zip(cvr["mean_test_score"], cvr["std_test_score"], cvr["params"]):
            out.write("%0.3f (+/-%0.03f) for %r \n" % (mean, std * 2, param))
        out.write("\nBest parameters found on development set: \n")
        for key, value in self.model_fitted.best_params_.items():
            out.write(key + " : " + str(value) + "\n")

    def _predict_report(self, out, index, X, y, store):
        """(:1239-1253, :1255-1288, :1314-1380); `store` collects the per-split metrics for the means."""
        pred = self.model_fitted.predict(X)
        out.write("\nModel predictions on samples:\nSample_ID Acutal_phenotype Predicted_phenotype\n")
        # the printed value is a ONE-ROW prediction, as the reference computes it (:1245-1249): a regressor's last digit is the
        # BLAS sum's, which is not the same for a row alone and for the row inside the matrix; the metrics use `pred`
        for i, (name, actual) in enumerate(zip(index, y)):
            out.write("%s %s %s\n" % (name, actual, self.model_fitted.predict(X[i:i + 1])[0]))
        out.write("\n")

        def keep(key, value):
            if store is not None:
                store[key].append(value)
            return value
        if self.pred_scale == "continuous":
            out.write("\nMean squared error: %s\n" % keep("MSE", metrics.mean_squared_error(y, pred).round(2)))
            out.write("The coefficient of determination: %s\n" % keep("CoD", round(self.model_fitted.score(X, y), 2)))
            r, pv = _stats.spearmanr(y, pred)
            out.write("The Spearman correlation coefficient and p-value: %s, %s \n"
                      % (keep("SpCC", round(r, 2)), keep("Sp_pval", round(pv, 2))))
            r, pv = _stats.pearsonr(y, pred)
            out.write("The Pearson correlation coefficient and p-value:  %s, %s \n"
                      % (keep("PeCC", round(r, 2)), keep("Pe_pval", round(pv, 2))))
            out.write("The plus/minus 1 dilution factor accuracy (for MICs): %s \n\n"
                      % keep("DFA", metrics.within_1_tier_accuracy(y, pred)))
            return
        proba = self.model_fitted.predict_proba(X)[:, 1]
        out.write("F1-score of positive class: %s\n" % keep("F1_sc", metrics.f1(y, pred).round(2)))
        out.write("Mean accuracy: %s\n" % keep("Acc", self.model_fitted.score(X, y).round(2)))
        out.write("Sensitivity: %s\n" % keep("Sn", metrics.recall(y, pred).round(2)))
        out.write("Specificity: %s\n" % keep("Sp", metrics.recall(y, pred, positive=0).round(2)))
        out.write("AUC-ROC: %s\n" % keep("AUCROC", metrics.roc_auc(y, pred).round(2)))
        out.write("Average precision: %s\n" % keep("Pr", metrics.average_precision(y, proba).round(2)))
        out.write("MCC: %s\n" % keep("MCC", round(metrics.matthews(y, pred), 2)))
        out.write("Cohen kappa: %s\n" % keep("kappa", metrics.cohen_kappa(y, pred).round(2)))
        out.write("Very major error rate: %s\n" % keep("VME", metrics.very_major_error(y, pred)))
        out.write("Major error rate: %s\n" % keep("ME", metrics.major_error(y, pred)))
        out.write("Classification report:\n\n %s\n" % metrics.classification_report(y, pred))
        cm = metrics.confusion(y, pred)
        out.write("Confusion matrix:\n")
        out.write("Predicted\t0\t1:\n")
        out.write("Actual\n")
        out.write("0\t\t%s\t%s\n" % tuple(cm[0]))
        out.write("1\t\t%s\t%s\n\n" % tuple(cm[1]))

    def _mean_report(self, out, store):
        """(:1290-1312, :1382-1412)"""
        mean = {k: np.mean(v).round(2) for k, v in store.items() if len(v)}
        if self.pred_scale == "continuous":
            out.write("\nMean squared error: %s\n" % mean["MSE"])
            out.write("The coefficient of determination: %s\n" % mean["CoD"])
            out.write("The Spearman correlation coefficient and p-value: %s, %s \n" % (mean["SpCC"], mean["Sp_pval"]))
            out.write("The Pearson correlation coefficient and p-value:  %s, %s \n" % (mean["PeCC"], mean["Pe_pval"]))
            out.write("The plus/minus 1 dilution factor accuracy (for MICs): %s \n\n" % mean["DFA"])
            return
        for label, key in (("F1-score of positive class", "F1_sc"), ("Mean accuracy", "Acc"), ("Sensitivity", "Sn"),
                           ("Specificity", "Sp"), ("AUC-ROC", "AUCROC"), ("Average precision", "Pr"), ("MCC", "MCC"),
                           ("Cohen kappa", "kappa"), ("Very major error rate", "VME"), ("Major error rate", "ME")):
            out.write("%s: %s\n" % (label, mean[key]))

    def _write_model_coefficients(self, path):
        """(:1414-1455).  The header is written and the file CLOSED here; libpsk then appends the lines to `path` (no Python
        file object is shared with it: a stale offset could overwrite what it wrote)."""
        with open(path, "w") as out:
            out.write("K-mer\tcoef._in_" + self.model_name_short + "_model\tNo._of_samples_with_k-mer\tSamples_with_k-mer\n")
        be = self.model_fitted.best_estimator_
        coefs = be.coef_[0] if self.pred_scale == "binary" else be.coef_
        X, index, kmers = self.ML["X"], self.ML["index"], self.ML["kmers"]
        # the lines are formatted by libpsk (psk_write_model_coefficients: a 2,048-sample model names a million samples --
        # 0.07 s of joins here); it appends to the file whose header this function has just written
        kenc, nenc = [km.encode() for km in kmers], [nm.encode() for nm in index]
        koff = np.zeros(len(kenc) + 1, dtype=np.int64)
        koff[1:] = np.cumsum([len(e) for e in kenc])
        noff = np.zeros(len(nenc) + 1, dtype=np.int64)
        noff[1:] = np.cumsum([len(e) for e in nenc])
        Xa = np.ascontiguousarray(np.asarray(X).reshape(len(nenc), len(kenc)), dtype=np.int64)
        cf = np.ascontiguousarray(np.asarray(coefs, dtype=np.float64).reshape(-1)[:len(kenc)])
        vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        rc = _lib.load().psk_write_model_coefficients(None, os.fsencode(path), len(kenc), b"".join(kenc), vp(koff), vp(cf), vp(Xa),
                                                      len(nenc), b"".join(nenc), vp(noff))
        if rc != 0:
            raise PskError("psk_write_model_coefficients failed (%d)" % rc, rc)


def _metric_store():
    return {k: [] for k in ("MSE", "CoD", "SpCC", "Sp_pval", "PeCC", "Pe_pval", "DFA", "Acc", "Sn", "Sp", "AUCROC",
                            "Pr", "MCC", "kappa", "VME", "ME", "F1_sc")}


def process_start_epoch():
    """When this process was started (epoch seconds): field 22 of /proc/self/stat, clock ticks after boot."""
    try:
        with open("/proc/self/stat") as f:
            ticks = float(f.read().rsplit(")", 1)[1].split()[19])
        with open("/proc/uptime") as f:
            up = float(f.read().split()[0])
        return time.time() - (up - ticks / os.sysconf("SC_CLK_TCK"))
    except (OSError, ValueError, IndexError):
        return time.time()


class Phases:
    """Where the wall-clock of a run goes, by phase and by rank -- so that the first run on several GPUs produces a number
    somebody can read (eight ranks sharing one GPU took 5.4 s for a set one rank does in 0.2 s, and
    nothing said how much of it was eight interpreters and eight HIP start-ups).  mark(name) closes the phase that began at
    the mark before it; the table goes to phases_rank<r>.json in the working directory (and, rank 0, into log.txt); its
    entries sum to `total_s` by construction.  t0: the process's start when the CLI is the process (cli.main), else
    the call."""
    current = None

    def __init__(self, t0=None):
        self.t0 = self.last = time.time() if t0 is None else t0
        self.table = OrderedDict()

    def enter(self, name):
        """Names the phase that starts now -- what a rank that hangs in it reports (watchdog.py); mark() still books it."""
        from . import watchdog
        watchdog.enter(name)

    def snapshot(self):
        """The finished phases, for the table of a rank that is stuck in the next one."""
        return {"total_s": round(time.time() - self.t0, 4), "phases_s": OrderedDict((k, round(v, 4)) for k, v in list(self.table.items()))}

    def mark(self, name):
        now = time.time()
        self.table[name] = self.table.get(name, 0.0) + (now - self.last)
        self.last = now

    def move(self, src, dst, secs):
        """`secs` of what will be booked under `src` belong to `dst` (a step timed inside another phase)."""
        self.table[dst] = self.table.get(dst, 0.0) + secs
        self.table[src] = self.table.get(src, 0.0) - secs

    def write(self, rank, world):
        rec = {"rank": rank, "world": world, "total_s": round(self.last - self.t0, 4),
               "phases_s": OrderedDict((k, round(v, 4)) for k, v in self.table.items())}
        import json
        with open("phases_rank%d.json" % rank, "w") as f:
            json.dump(rec, f)
        if rank == 0:
            with open("log.txt", "a") as log:
                log.write("Phases of rank 0 (s): %s; total %.3f\n" % (", ".join("%s %.3f" % kv for kv in rec["phases_s"].items()), rec["total_s"]))
        return rec


def modeling(args):
    """The main function of `phenotypeseeker modeling` (:1624-1709)."""
    ph_t = Phases.current = Phases(getattr(args, "_t0", None))
    ph_t.mark("process start, interpreter, imports")
    ours = os.environ.get("PSK_LAUNCHER") == "psk"
    if (ours or getattr(args, "_t0", None) is not None) and threading.current_thread() is threading.main_thread():
        # a rank of this package's own launcher, or the CLI as a process of its own: a launch that runs into its deadline
        # (launch.spawn_ranks: PSK_LAUNCH_TIMEOUT) asks every rank where it is.  (Only there: the answer takes over SIGUSR1 and
        # the interpreter's wake-up descriptor, which a host application that merely calls modeling() may be using itself.)
        from . import watchdog
        watchdog.install(os.environ.get("RANK", "0"), os.environ.get("WORLD_SIZE", "1"), ph_t.snapshot)
        if not ours and int(os.environ.get("WORLD_SIZE", "1")) > 1:    # somebody else's launcher: every rank keeps the deadline itself
            from . import launch as _launch
            watchdog.self_deadline(_launch.launch_timeout(default=0))     # (only when PSK_LAUNCH_TIMEOUT is set)
    ph_t.enter("arguments, data.pheno")
    _err(RED_BANNER % "######                   PhenotypeSeeker                   ######")
    _err(RED_BANNER % "######                      modeling                       ######" + "\n")
    Input.reset()
    Input.get_input_data(args.inputfile, args.take_logs, args.mpheno)
    Input.Input_args(
        args.alphas, args.alpha_min, args.alpha_max, args.n_alphas, args.gammas, args.gamma_min, args.gamma_max,
        args.n_gammas, args.min, args.max, args.kmer_length, args.cutoff, args.num_threads, args.pvalue,
        args.n_kmers, args.binary_classifier, args.regressor, args.penalty, args.max_iter, args.tolerance,
        args.l1_ratio, args.n_splits_cv_outer, args.kernel, args.n_iter, args.n_splits_cv_inner, args.testset_size,
        args.train_on_whole, args.logreg_solver, args.jump_to, args.pca, args.real_counts, args.omit_B_correction,
        args.kmerDB)
    Samples.use_weights = bool(getattr(args, "weights", False))
    ph_t.mark("arguments, data.pheno")
    ph_t.enter("rendezvous, communicator (ncclCommInitRank)")
    group = _dist.Group().init()
    ph_t.mark("rendezvous, communicator (ncclCommInitRank)")
    for name, secs in _dist.init_times.items():     # the bring-up's own steps (the GPU probe in it pays the HIP start-up)
        ph_t.move("rendezvous, communicator (ncclCommInitRank)", name if name.startswith("HIP") else "rendezvous: " + name, secs)
    ph_t.enter("HIP runtime, context")
    ctx = PskContext(group.device)
    ph_t.mark("HIP runtime, context")
    try:
        if not Input.jump_to:
            k = int(Samples.kmer_length)
            _err(GREEN % "Generating the k-mer lists for input samples:" + "\n")
            n_thr = max(1, min(int(Input.num_threads), 8))
            ph_t.enter("ingest: k-mer lists")
            t_lists = time.time()
            # several ranks: the list exchange when the collectives run GPU to GPU (RCCL); with the host transport of the tests they are
            # staged through the host (7 GB through TCP loopback for 2 x 128 genomes: 5.7 s against 0.12 s), so
            # every rank counts everything there.  PSK_REDUNDANT_INGEST = 0 / 1 forces either.
            knob = os.environ.get("PSK_REDUNDANT_INGEST")
            exchange = group.world > 1 and (knob == "0" or (knob != "1" and getattr(group, "backend", None) == "rccl"))
            if group.world > 1:
                Samples.get_kmer_lists_sharded(ctx, group, list(Input.samples.values()), n_thr, exchange)
            else:
                ctx.begin(k, Samples.no_samples)
                Samples.get_kmer_lists_batched(ctx, list(Input.samples.values()), n_thr)
            if group.rank == 0:
                with open("log.txt", "a") as log:
                    log.write("Func get_kmer_lists took %s secs (%d rank(s)%s)\n"
                              % (time.time() - t_lists, group.world, ", list exchange" if exchange else ""))
            ph_t.mark("ingest: k-mer lists" + (" (list exchange)" if exchange else " (every rank filters its slab)" if group.world > 1 else ""))
            _err("\n" + GREEN % "Generating the k-mer feature vector." + "\n")
            ph_t.enter("presence matrix")
            m_local = Samples.get_feature_vector(ctx)
            ph_t.mark("presence matrix")
            _err(GREEN % "Mapping samples to the feature vector space:" + "\n")
            stderr_print("\t%d of %d samples mapped." % (Samples.no_samples, Samples.no_samples))
            if Samples.use_weights:
                _err("\n" + GREEN % "Estimating the Mash distances between samples..." + "\n")
                stderr_print(GREEN % "Calculating the GSC weights from mash distance matrix...")
                ph_t.enter("weights: sketches, distances, NJ, GSC")
                Samples.get_weights(ctx, write_files=group.rank == 0)
                ph_t.mark("weights: sketches, distances, NJ, GSC")
            ph_t.enter("all-reduce of the union size")
            phenotypes.kmer_testing_setup(group.allreduce_sum(int(m_local)))
            ph_t.mark("all-reduce of the union size")
            phs = list(Input.phenotypes_to_analyse.values())
            ph_t.enter("scan")
            for j, ph in enumerate(phs):
                ph.test_kmers_association_with_phenotype(ctx, group, phs[j + 1] if j + 1 < len(phs) else None)
            ph_t.mark("scan")      # (the survivors' all-gather inside it is moved to its own line: Phases.move)
            Input.pop_phenos_out_of_kmers()
        # 'modelling' / 'modeling' both run the model stage; 'PCA' runs nothing (:1689)
        if not Input.jump_to or Input.jump_to in ("modelling", "modeling"):
            if Input.jump_to:
                Input.jump_to = "modelling"
            _err(GREEN % ("Generating the " + phenotypes.model_name_long + " model for phenotype: ") + "\n")
            ph_t.enter("model: result tables, grid search, model files")
            if group.rank == 0:
                for ph in Input.phenotypes_to_analyse.values():
                    ph.machine_learning_modelling(ctx)
            ph_t.mark("model: result tables, grid search, model files")
            ph_t.enter("waiting for rank 0's model")
            group.barrier()
            ph_t.mark("waiting for rank 0's model")
        if getattr(args, "assembly", False):
            _err(YELLOW % "-a/--assembly is outside the accelerated path and is skipped.")
    finally:
        ph_t.enter("teardown: buffers, communicator")
        ctx.close()
        group.close()
        ph_t.mark("teardown: buffers, communicator")
        ph_t.write(group.rank, group.world)
    _err("\n" + RED_BANNER % "######          PhenotypeSeeker modeling finished          ######")
