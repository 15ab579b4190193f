"""PskContext: numpy-level wrapper over the C ABI (one context = one GPU = one rank)."""
import ctypes
import os

import numpy as np

from . import _lib
from ._lib import PskError


def words_per_row(n_samples):
    """u64 words per presence row: 1 up to 64 samples, else ceil(n/64) rounded up to even (rows are 16-byte aligned)."""
    w = (n_samples + 63) // 64
    return 1 if w <= 1 else (w + 1) & ~1


def _ptr(a):
    return None if a is None else a.ctypes.data


def _file_size(path):
    try:
        return os.stat(path).st_size
    except OSError:
        return 0


class PskContext:
    def __init__(self, device=0):
        self._lib = _lib.load()
        h = ctypes.c_void_p()
        rc = self._lib.psk_init(int(device), ctypes.byref(h))
        if rc != 0:
            raise PskError("psk_init(%d) failed: %s" % (device, self._lib.psk_last_error(None).decode()))
        self._h = h
        self.device = device
        self.k = None
        self.n_samples = None

    # -- plumbing -------------------------------------------------------------------------------
    def _check(self, rc, what):
        if rc != 0:
            raise PskError("%s failed (%d): %s" % (what, rc, self._lib.psk_last_error(self._h).decode()), code=rc)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.psk_free(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_info(self):
        name = ctypes.create_string_buffer(256)
        ncu = ctypes.c_int()
        mem = ctypes.c_uint64()
        self._check(self._lib.psk_device_info(self._h, name, 256, ctypes.byref(ncu), ctypes.byref(mem)), "device_info")
        return {"name": name.value.decode(), "n_cu": ncu.value, "hbm_bytes": mem.value}

    # -- k-mer plane ----------------------------------------------------------------------------
    def begin(self, k, n_samples, slab_lo=0, slab_hi=0):
        self._check(self._lib.psk_begin(self._h, int(k), int(n_samples), int(slab_lo), int(slab_hi)), "psk_begin")
        self.k, self.n_samples = int(k), int(n_samples)

    def count_kmers(self, sample_idx, data):
        data = bytes(data)
        nu, nt = ctypes.c_uint64(), ctypes.c_uint64()
        self._check(self._lib.psk_count_kmers(self._h, int(sample_idx), data, len(data), ctypes.byref(nu),
                                              ctypes.byref(nt)), "psk_count_kmers")
        return nu.value, nt.value

    def count_kmers_batch(self, first_idx, datas, n_threads=4, sketch=None):
        """Counts several samples in one call; host tokenisation runs ahead on n_threads threads.
        sketch=(k, size, seed): also returns every sample's MinHash sketch (third element: list of uint64 arrays),
        computed from the clean stream that is on the device for counting anyway."""
        datas = [bytes(d) for d in datas]
        n = len(datas)
        arr = (ctypes.c_char_p * n)(*datas)
        lens = (ctypes.c_size_t * n)(*[len(d) for d in datas])
        nu = np.zeros(n, dtype=np.uint64)
        nt = np.zeros(n, dtype=np.uint64)
        if sketch is None:
            self._check(self._lib.psk_count_kmers_batch(self._h, int(first_idx), n, arr, lens, _ptr(nu), _ptr(nt),
                                                        int(n_threads)), "psk_count_kmers_batch")
            return nu.astype(np.int64).tolist(), nt.astype(np.int64).tolist()
        k, size, seed = sketch
        hashes = np.zeros((n, int(size)), dtype=np.uint64)
        nh = np.zeros(n, dtype=np.uint64)
        self._check(self._lib.psk_count_kmers_batch_sketch(self._h, int(first_idx), n, arr, lens, _ptr(nu), _ptr(nt),
                                                           int(n_threads), int(k), int(size), int(seed), _ptr(hashes),
                                                           _ptr(nh)), "psk_count_kmers_batch_sketch")
        return (nu.astype(np.int64).tolist(), nt.astype(np.int64).tolist(),
                [hashes[i, : int(nh[i])].copy() for i in range(n)])

    def count_kmers_files(self, first_idx, paths, n_threads=4, sketch=None):
        """count_kmers_batch for UNCOMPRESSED files: the library's framing threads read them (psk_count_kmers_files).
        Returns (n_unique, n_total[, sketches])."""
        enc = [os.fsencode(p) for p in paths]
        n = len(enc)
        arr = (ctypes.c_char_p * n)(*enc)
        sizes = (ctypes.c_size_t * n)(*[_file_size(p) for p in paths])
        nu = np.zeros(n, dtype=np.uint64)
        nt = np.zeros(n, dtype=np.uint64)
        k, size, seed = sketch if sketch is not None else (0, 1, 0)
        hashes = np.zeros((n, int(size)), dtype=np.uint64) if sketch is not None else None
        nh = np.zeros(n, dtype=np.uint64) if sketch is not None else None
        self._check(self._lib.psk_count_kmers_files(self._h, int(first_idx), n, arr, sizes, _ptr(nu), _ptr(nt), int(n_threads),
                                                    int(k), int(size), int(seed), _ptr(hashes), _ptr(nh)),
                    "psk_count_kmers_files")
        out = (nu.astype(np.int64).tolist(), nt.astype(np.int64).tolist())
        if sketch is not None:
            out += ([hashes[i, : int(nh[i])].copy() for i in range(n)],)
        return out

    def get_list(self, sample_idx, n_unique):
        words = np.empty(n_unique, dtype=np.uint64)
        freqs = np.empty(n_unique, dtype=np.uint32)
        self._check(self._lib.psk_get_list(self._h, int(sample_idx), _ptr(words), _ptr(freqs), n_unique), "psk_get_list")
        return words, freqs

    def gz_inflate(self, images, want_text=True, per_file=False):
        """The text of gzip images, inflated on the device (csrc/gz_inflate.hip).  Returns (texts or None, lengths, routes,
        device_ms); routes: 1 device, 2 device (BGZF), 0 zlib on the host.  want_text=False: lengths only (measurements).
        A file zlib refuses raises PskError (zlib's words) -- per_file=True instead returns route -1 and text None for it."""
        images = [bytes(b) for b in images]
        n = len(images)
        ptrs = (ctypes.c_char_p * max(n, 1))(*images)
        sizes = (ctypes.c_size_t * max(n, 1))(*[len(b) for b in images])
        lens = np.zeros(max(n, 1), dtype=np.uint64)
        route = np.zeros(max(n, 1), dtype=np.int32)
        ms = ctypes.c_double()

        def call(outp, caps):
            route[:] = 0
            rc = self._lib.psk_gz_inflate(self._h, n, ptrs, sizes, outp, caps, _ptr(lens), _ptr(route), ctypes.byref(ms))
            if not (per_file and rc == -1 and (route[:n] == -1).any()):     # PSK_EINVAL with the refused files marked
                self._check(rc, "psk_gz_inflate")
        call(None, None)
        if not want_text:
            return None, lens[:n].tolist(), route[:n].tolist(), ms.value
        bufs = [np.empty(max(int(l), 1), dtype=np.uint8) for l in lens[:n]]
        outp = (ctypes.c_void_p * max(n, 1))(*[b.ctypes.data for b in bufs])
        caps = (ctypes.c_size_t * max(n, 1))(*[len(b) for b in bufs])
        call(outp, caps)
        return ([None if r == -1 else b[:int(l)].tobytes() for b, l, r in zip(bufs, lens[:n], route[:n])], lens[:n].tolist(),
                route[:n].tolist(), ms.value)

    # -- multi-GPU ingest: slab ranges of sorted lists (dist.ListExchange) ---------------------------
    def lists_split(self, first_idx, n, bounds):
        """offsets[i][b] = number of words of sample first_idx + i below bounds[b] (a 0 after the first bound = end)."""
        b = np.ascontiguousarray(bounds, dtype=np.uint64)
        out = np.zeros((int(n), len(b)), dtype=np.uint64)
        self._check(self._lib.psk_lists_split(self._h, int(first_idx), int(n), _ptr(b), len(b), _ptr(out)), "psk_lists_split")
        return out.astype(np.int64)

    def copy_list_ranges(self, sample_idx, start, count, dev_words_ptr, dev_freqs_ptr):
        """Ranges [start[r], start[r] + count[r]) of the lists sample_idx[r], packed back to back into DEVICE buffers
        given as raw pointers (torch's data_ptr())."""
        si = np.ascontiguousarray(sample_idx, dtype=np.int32)
        st = np.ascontiguousarray(start, dtype=np.uint64)
        ct = np.ascontiguousarray(count, dtype=np.uint64)
        self._check(self._lib.psk_copy_list_ranges(self._h, len(si), _ptr(si), _ptr(st), _ptr(ct), ctypes.c_void_p(dev_words_ptr),
                                                   ctypes.c_void_p(dev_freqs_ptr)), "psk_copy_list_ranges")

    def release_lists(self):
        """Gives the device memory of the lists back; every sample is "not counted" again."""
        self._check(self._lib.psk_release_lists(self._h), "psk_release_lists")

    def set_lists_device(self, sample_idx, count, n_total, dev_words_ptr, dev_freqs_ptr):
        """Installs len(sample_idx) lists held back to back in DEVICE memory as the lists of those samples."""
        si = np.ascontiguousarray(sample_idx, dtype=np.int32)
        ct = np.ascontiguousarray(count, dtype=np.uint64)
        tot = np.ascontiguousarray(n_total, dtype=np.uint64)
        self._check(self._lib.psk_set_lists_device(self._h, len(si), _ptr(si), _ptr(ct), _ptr(tot), ctypes.c_void_p(dev_words_ptr),
                                                   ctypes.c_void_p(dev_freqs_ptr)), "psk_set_lists_device")

    def lookup_counts(self, sample_idx, words):
        words = np.ascontiguousarray(words, dtype=np.uint64)
        out = np.zeros(len(words), dtype=np.uint32)
        self._check(self._lib.psk_lookup_counts(self._h, int(sample_idx), _ptr(words), len(words), _ptr(out)),
                    "psk_lookup_counts")
        return out

    def build_presence(self):
        m = ctypes.c_uint64()
        self._check(self._lib.psk_build_presence(self._h, ctypes.byref(m)), "psk_build_presence")
        return m.value

    def presence_shape(self):
        m, w, n = ctypes.c_uint64(), ctypes.c_int(), ctypes.c_int()
        self._check(self._lib.psk_presence_shape(self._h, ctypes.byref(m), ctypes.byref(w), ctypes.byref(n)),
                    "psk_presence_shape")
        return m.value, w.value, n.value

    def get_union(self):
        m, _, _ = self.presence_shape()
        words = np.empty(m, dtype=np.uint64)
        self._check(self._lib.psk_get_union(self._h, _ptr(words), m), "psk_get_union")
        return words

    def get_rows(self, row_idx):
        row_idx = np.ascontiguousarray(row_idx, dtype=np.uint64)
        _, wpr, _ = self.presence_shape()
        out = np.zeros((len(row_idx), wpr), dtype=np.uint64)
        self._check(self._lib.psk_get_rows(self._h, _ptr(row_idx), len(row_idx), _ptr(out)), "psk_get_rows")
        return out

    def intersect_db(self, db_words):
        db = np.ascontiguousarray(np.unique(np.asarray(db_words, dtype=np.uint64)))
        m = ctypes.c_uint64()
        self._check(self._lib.psk_intersect_db(self._h, _ptr(db), len(db), ctypes.byref(m)), "psk_intersect_db")
        return m.value

    def set_presence(self, bits, n_samples, words=None):
        bits = np.ascontiguousarray(bits, dtype=np.uint64)
        m, wpr = bits.shape
        if words is not None:
            words = np.ascontiguousarray(words, dtype=np.uint64)
        self._check(self._lib.psk_set_presence(self._h, _ptr(words), _ptr(bits), m, wpr, int(n_samples)),
                    "psk_set_presence")
        self.n_samples = int(n_samples)

    def synth_presence(self, n_kmers, n_samples, seed=1):
        self._check(self._lib.psk_synth_presence(self._h, int(n_kmers), int(n_samples), int(seed)), "psk_synth_presence")
        self.n_samples = int(n_samples)

    # -- association scans ----------------------------------------------------------------------
    def chi2_scan(self, pheno, weights, min_samples, max_samples, pvalue_cutoff, omit_B, n_kmers_global=0):
        """pheno: int8 array, 1 / 0 / -1 (NA).  Returns the number of surviving k-mers."""
        ph = np.ascontiguousarray(pheno, dtype=np.int8)
        w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
        n = ctypes.c_uint64()
        self._check(self._lib.psk_chi2_scan(self._h, _ptr(ph), _ptr(w), int(min_samples), int(max_samples),
                                            float(pvalue_cutoff), int(bool(omit_B)), int(n_kmers_global),
                                            ctypes.byref(n)), "psk_chi2_scan")
        return n.value

    def chi2_scan_begin(self, pheno, weights, min_samples, max_samples, pvalue_cutoff, omit_B, n_kmers_global=0):
        """Launches the scan and returns at once; scan_end() waits for it (psk_chi2_scan_begin / psk_scan_end)."""
        ph = np.ascontiguousarray(pheno, dtype=np.int8)
        w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
        self._check(self._lib.psk_chi2_scan_begin(self._h, _ptr(ph), _ptr(w), int(min_samples), int(max_samples),
                                                  float(pvalue_cutoff), int(bool(omit_B)), int(n_kmers_global)),
                    "psk_chi2_scan_begin")

    def scan_end(self):
        n = ctypes.c_uint64()
        self._check(self._lib.psk_scan_end(self._h, ctypes.byref(n)), "psk_scan_end")
        return n.value

    def ttest_scan(self, values, valid, weights, min_samples, max_samples, pvalue_cutoff, n_kmers_global=0):
        v = np.ascontiguousarray(values, dtype=np.float64)
        ok = np.ascontiguousarray(valid, dtype=np.uint8)
        w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
        n = ctypes.c_uint64()
        self._check(self._lib.psk_ttest_scan(self._h, _ptr(v), _ptr(ok), _ptr(w), int(min_samples), int(max_samples),
                                             float(pvalue_cutoff), int(n_kmers_global), ctypes.byref(n)),
                    "psk_ttest_scan")
        return n.value

    def get_results(self, n_pass):
        n = int(n_pass)
        out = {"row": np.zeros(n, np.uint64), "word": np.zeros(n, np.uint64), "stat": np.zeros(n), "p": np.zeros(n),
               "mean_x": np.zeros(n), "mean_y": np.zeros(n), "n_with": np.zeros(n, np.int32)}
        self._check(self._lib.psk_get_results(self._h, _ptr(out["row"]), _ptr(out["word"]), _ptr(out["stat"]),
                                              _ptr(out["p"]), _ptr(out["mean_x"]), _ptr(out["mean_y"]),
                                              _ptr(out["n_with"]), n), "psk_get_results")
        return out

    def export_survivors(self, device_ptr, cap_records):
        """Packs the last scan's survivors into a caller DEVICE buffer (see psk_export_survivors)."""
        n = ctypes.c_uint64()
        self._check(self._lib.psk_export_survivors(self._h, ctypes.c_void_p(int(device_ptr)), int(cap_records),
                                                   ctypes.byref(n)), "psk_export_survivors")
        return n.value

    def export_survivors_async(self, device_ptr, cap_records, stream_handle):
        """psk_export_survivors_async: the export is queued on `stream_handle` (a hipStream_t as an integer) and
        not waited for.  The number of records is the n_pass the scan returned."""
        self._check(self._lib.psk_export_survivors_async(self._h, ctypes.c_void_p(int(device_ptr)), int(cap_records),
                                                         ctypes.c_void_p(int(stream_handle))), "psk_export_survivors_async")

    # -- multi-GPU collectives (RCCL, comm.hip) and plain device buffers ---------------------------
    def comm_unique_id(self):
        buf = ctypes.create_string_buffer(128)
        n = self._lib.psk_comm_unique_id(self._h, buf, 128)
        if n < 0:
            self._check(n, "psk_comm_unique_id")
        return buf.raw[:n]

    def comm_init(self, uid, rank, world):
        self._check(self._lib.psk_comm_init(self._h, bytes(uid), len(uid), int(rank), int(world)), "psk_comm_init")

    def comm_size(self):
        """ncclCommCount of this context's communicator."""
        n = self._lib.psk_comm_size(self._h)
        if n < 0:
            self._check(n, "psk_comm_size")
        return n

    def comm_stream(self):
        return self._lib.psk_comm_stream(self._h) or 0

    def comm_sync(self):
        self._check(self._lib.psk_comm_sync(self._h), "psk_comm_sync")

    def comm_allreduce(self, arr, op="sum"):
        """In place on a contiguous uint64 or float64 array."""
        assert arr.dtype in (np.uint64, np.float64) and arr.flags.c_contiguous
        self._check(self._lib.psk_comm_allreduce(self._h, _ptr(arr), arr.size, 0 if arr.dtype == np.uint64 else 1,
                                                 {"sum": 0, "max": 1}[op]), "psk_comm_allreduce")
        return arr

    def comm_allgather_host(self, send, world):
        send = np.ascontiguousarray(send).view(np.uint8).ravel()
        recv = np.empty((world, send.size), dtype=np.uint8)
        self._check(self._lib.psk_comm_allgather_host(self._h, _ptr(send), _ptr(recv), send.size), "psk_comm_allgather_host")
        return recv

    def comm_allgather_device(self, send_ptr, recv_ptr, nbytes):
        self._check(self._lib.psk_comm_allgather_device(self._h, ctypes.c_void_p(int(send_ptr)), ctypes.c_void_p(int(recv_ptr)),
                                                        int(nbytes)), "psk_comm_allgather_device")

    def comm_alltoallv_device(self, send_ptr, send_counts, recv_ptr, recv_counts, elem_bytes):
        sc = np.ascontiguousarray(send_counts, dtype=np.uint64)
        rc = np.ascontiguousarray(recv_counts, dtype=np.uint64)
        self._check(self._lib.psk_comm_alltoallv_device(self._h, ctypes.c_void_p(int(send_ptr)), _ptr(sc),
                                                        ctypes.c_void_p(int(recv_ptr)), _ptr(rc), int(elem_bytes)),
                    "psk_comm_alltoallv_device")

    def dev_alloc(self, nbytes):
        p = ctypes.c_void_p()
        self._check(self._lib.psk_dev_alloc(self._h, int(nbytes), ctypes.byref(p)), "psk_dev_alloc")
        return p.value

    def dev_free(self, ptr):
        if ptr and getattr(self, "_h", None):
            self._check(self._lib.psk_dev_free(self._h, ctypes.c_void_p(int(ptr))), "psk_dev_free")

    def dev_download(self, ptr, nbytes, behind_collectives=False):
        out = np.empty(int(nbytes), dtype=np.uint8)
        self._check(self._lib.psk_dev_copy(self._h, _ptr(out), ctypes.c_void_p(int(ptr)), int(nbytes), 1,
                                           1 if behind_collectives else 0), "psk_dev_copy")
        return out

    def dev_upload(self, ptr, arr):
        a = np.ascontiguousarray(arr)
        self._check(self._lib.psk_dev_copy(self._h, ctypes.c_void_p(int(ptr)), _ptr(a), a.nbytes, 0, 0), "psk_dev_copy")

    def last_scan_ms(self):
        return self._lib.psk_last_scan_ms(self._h)

    def rescan_timed(self, reps):
        ms = ctypes.c_double()
        self._check(self._lib.psk_rescan_timed(self._h, int(reps), ctypes.byref(ms)), "psk_rescan_timed")
        return ms.value

    def rescan_times(self, reps):
        """HIP-event duration in ms of each of `reps` repeats of the last chi2 scan (psk_rescan_times)."""
        ms = np.zeros(int(reps), dtype=np.float64)
        self._check(self._lib.psk_rescan_times(self._h, int(reps), _ptr(ms)), "psk_rescan_times")
        return ms

    def stream_read_ceiling(self, reps):
        """(mean ms, bytes, shape) of the fastest plain 16-B-per-lane read of the presence matrix: the stream-read ceiling
        the scan's achieved bandwidth is quoted against (psk_stream_read_ceiling)."""
        ms, nb, shape = ctypes.c_double(), ctypes.c_uint64(), ctypes.c_int()
        self._check(self._lib.psk_stream_read_ceiling(self._h, int(reps), ctypes.byref(ms), ctypes.byref(nb), ctypes.byref(shape)),
                    "psk_stream_read_ceiling")
        return ms.value, nb.value, ("grid-stride, non-temporal", "grid-stride", "wave-contiguous, non-temporal", "wave-contiguous")[shape.value]

    # -- models -----------------------------------------------------------------------------------
    def _fit(self, fn, name, X, y, ydtype, fold, fit_param, fit_fold, tol, max_iter):
        X = np.ascontiguousarray(X, dtype=np.float32)
        n, p = X.shape
        y = np.ascontiguousarray(y, dtype=ydtype)
        fold = np.ascontiguousarray(fold, dtype=np.int32)
        fit_param = np.ascontiguousarray(fit_param, dtype=np.float64)
        fit_fold = np.ascontiguousarray(fit_fold, dtype=np.int32)
        nf = len(fit_param)
        coef = np.zeros((nf, p))
        icpt = np.zeros(nf)
        iters = np.zeros(nf, dtype=np.int32)
        self._check(fn(self._h, _ptr(X), _ptr(y), n, p, _ptr(fold), _ptr(fit_param), _ptr(fit_fold), nf, float(tol),
                       int(max_iter), _ptr(coef), _ptr(icpt), _ptr(iters)), name)
        return coef, icpt, iters

    def logreg_l1_fit(self, X, y01, fold, fit_param, fit_fold, tol=1e-4, max_iter=1000):
        return self._fit(self._lib.psk_logreg_l1_fit, "psk_logreg_l1_fit", X, y01, np.int32, fold, fit_param, fit_fold,
                         tol, max_iter)

    def lasso_fit(self, X, y, fold, fit_param, fit_fold, tol=1e-4, max_iter=1000):
        return self._fit(self._lib.psk_lasso_fit, "psk_lasso_fit", X, y, np.float64, fold, fit_param, fit_fold, tol,
                         max_iter)

    def ridge_fit(self, X, y, fold, fit_param, fit_fold):
        """Ridge fits (psk_ridge_fit): same batching as lasso_fit, solved to convergence."""
        X = np.ascontiguousarray(X, dtype=np.float32)
        n, p = X.shape
        y = np.ascontiguousarray(y, dtype=np.float64)
        fold = np.ascontiguousarray(fold, dtype=np.int32)
        fit_param = np.ascontiguousarray(fit_param, dtype=np.float64)
        fit_fold = np.ascontiguousarray(fit_fold, dtype=np.int32)
        nf = len(fit_param)
        coef, icpt, iters = np.zeros((nf, p)), np.zeros(nf), np.zeros(nf, dtype=np.int32)
        self._check(self._lib.psk_ridge_fit(self._h, _ptr(X), _ptr(y), n, p, _ptr(fold), _ptr(fit_param), _ptr(fit_fold),
                                            nf, _ptr(coef), _ptr(icpt), _ptr(iters)), "psk_ridge_fit")
        return coef, icpt, iters

    def logreg_l2_fit(self, X, y01, fold, fit_param, fit_fold, tol=1e-4, max_iter=1000, penalise_intercept=False):
        X = np.ascontiguousarray(X, dtype=np.float32)
        n, p = X.shape
        y = np.ascontiguousarray(y01, dtype=np.int32)
        fold = np.ascontiguousarray(fold, dtype=np.int32)
        fit_param = np.ascontiguousarray(fit_param, dtype=np.float64)
        fit_fold = np.ascontiguousarray(fit_fold, dtype=np.int32)
        nf = len(fit_param)
        coef, icpt, iters = np.zeros((nf, p)), np.zeros(nf), np.zeros(nf, dtype=np.int32)
        self._check(self._lib.psk_logreg_l2_fit(self._h, _ptr(X), _ptr(y), n, p, _ptr(fold), _ptr(fit_param),
                                                _ptr(fit_fold), nf, float(tol), int(max_iter), int(bool(penalise_intercept)),
                                                _ptr(coef), _ptr(icpt), _ptr(iters)), "psk_logreg_l2_fit")
        return coef, icpt, iters

    # -- population-structure weights --------------------------------------------------------------
    def minhash_sketch(self, data, k=21, sketch_size=1000, seed=42):
        data = bytes(data)
        out = np.zeros(int(sketch_size), dtype=np.uint64)
        n = ctypes.c_uint64()
        self._check(self._lib.psk_minhash_sketch(self._h, data, len(data), int(k), int(sketch_size), int(seed),
                                                 _ptr(out), ctypes.byref(n)), "psk_minhash_sketch")
        return out[: n.value].copy()

    def mash_pairs(self, sketches, sketch_size=1000):
        """sketches: list of ascending hash lists.  Returns (common[n][n], denom[n][n]) of Mash's Jaccard estimate."""
        n = len(sketches)
        sk = np.zeros((n, int(sketch_size)), dtype=np.uint64)
        lens = np.zeros(n, dtype=np.uint32)
        for i, h in enumerate(sketches):
            h = np.asarray(h, dtype=np.uint64)[: int(sketch_size)]
            sk[i, : len(h)] = h
            lens[i] = len(h)
        common = np.zeros((n, n), dtype=np.uint32)
        denom = np.zeros((n, n), dtype=np.uint32)
        self._check(self._lib.psk_mash_pairs(self._h, _ptr(sk), _ptr(lens), n, int(sketch_size), _ptr(common), _ptr(denom)),
                    "psk_mash_pairs")
        return common, denom

    def nj_merges(self, dist):
        """Neighbour-joining merge list of a symmetric distance matrix (psk_nj_merges)."""
        d = np.ascontiguousarray(dist, dtype=np.float64)
        n = d.shape[0]
        mi = np.zeros(n - 2, dtype=np.int32)
        mj = np.zeros(n - 2, dtype=np.int32)
        d1 = np.zeros(n - 2)
        d2 = np.zeros(n - 2)
        last = np.zeros(1)
        self._check(self._lib.psk_nj_merges(self._h, _ptr(d), n, _ptr(mi), _ptr(mj), _ptr(d1), _ptr(d2), _ptr(last)),
                    "psk_nj_merges")
        return mi, mj, d1, d2, float(last[0])

    # -- prediction -------------------------------------------------------------------------------
    def count_dict(self, data, k, dict_words):
        data = bytes(data)
        d = np.ascontiguousarray(dict_words, dtype=np.uint64)
        out = np.zeros(len(d), dtype=np.uint32)
        self._check(self._lib.psk_count_dict(self._h, data, len(data), int(k), _ptr(d), len(d), _ptr(out)),
                    "psk_count_dict")
        return out


    def count_dict_batch(self, datas, k, dict_words, n_threads=4):
        """psk_count_dict_batch: every sample (file image) against one dictionary -> counts[n][n_dict]."""
        datas = [bytes(d) for d in datas]
        n = len(datas)
        d = np.ascontiguousarray(dict_words, dtype=np.uint64)
        out = np.zeros((n, len(d)), dtype=np.uint32)
        if n and len(d):
            arr = (ctypes.c_char_p * n)(*datas)
            lens = (ctypes.c_size_t * n)(*[len(x) for x in datas])
            self._check(self._lib.psk_count_dict_batch(self._h, n, arr, lens, int(k), _ptr(d), len(d), _ptr(out), int(n_threads)),
                        "psk_count_dict_batch")
        return out

    def count_dict_files(self, paths, k, dict_words, n_threads=4):
        """psk_count_dict_files: the same for UNCOMPRESSED files, read by the library's framing threads."""
        enc = [os.fsencode(p) for p in paths]
        n = len(enc)
        d = np.ascontiguousarray(dict_words, dtype=np.uint64)
        out = np.zeros((n, len(d)), dtype=np.uint32)
        if n and len(d):
            arr = (ctypes.c_char_p * n)(*enc)
            sizes = (ctypes.c_size_t * n)(*[os.path.getsize(p) for p in paths])
            self._check(self._lib.psk_count_dict_files(self._h, n, arr, sizes, int(k), _ptr(d), len(d), _ptr(out), int(n_threads)),
                        "psk_count_dict_files")
        return out


    def frame_sequence_gpu(self, data):
        """psk_frame_sequence_gpu: the clean stream as the device framing produces it (r06: FASTQ that is not four lines per
        record included -- the scan of line kinds of frame_gpu.hip; until r05 such input came back as None: the host's)."""
        data = bytes(data)
        out = np.empty(len(data) + 128, dtype=np.uint8)
        n = self._lib.psk_frame_sequence_gpu(self._h, data, len(data), _ptr(out), out.size)
        if n == -5:
            return None
        if n < 0:
            self._check(int(n), "psk_frame_sequence_gpu")
        return out[:n].tobytes()


def frame_sequence(data):
    """Host-only tokeniser framing (psk_frame_sequence): the clean stream handed to the GPU."""
    lib = _lib.load()
    data = bytes(data)
    out = np.zeros(max(len(data), 1), dtype=np.uint8)
    n = lib.psk_frame_sequence(data, len(data), out.ctypes.data, len(out))
    if n < 0:
        raise PskError("psk_frame_sequence failed: %d" % n)
    return out[:n].tobytes()
