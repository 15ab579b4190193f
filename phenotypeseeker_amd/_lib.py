"""ctypes binding of libpsk.so (include/psk.h).  Loading fails loudly: there is no CPU path."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PSK_LIB") or os.path.join(_HERE, "libpsk.so")  # PSK_LIB: A/B builds (tools/)


class PskError(RuntimeError):
    """`code`: the negative PSK_E* value of include/psk.h the call returned (None when the failure is the loader's)."""

    def __init__(self, msg, code=None):
        super().__init__(msg)
        self.code = code


PSK_EGZIP = -6    # include/psk.h: (PSK_NO_GPU_GZ=1 only, since r05) a file input is gzip-compressed; the caller inflates it and uses the in-memory call


c = ctypes
_u64p = c.POINTER(c.c_uint64)
_SIGNATURES = {
    # name: (restype, argtypes)   -- one entry per declaration in include/psk.h
    "psk_init": (c.c_int, [c.c_int, c.POINTER(c.c_void_p)]),
    "psk_free": (None, [c.c_void_p]),
    "psk_last_error": (c.c_char_p, [c.c_void_p]),
    "psk_version": (c.c_int, []),
    "psk_device_info": (c.c_int, [c.c_void_p, c.c_char_p, c.c_int, c.POINTER(c.c_int), _u64p]),
    "psk_begin": (c.c_int, [c.c_void_p, c.c_int, c.c_int, c.c_uint64, c.c_uint64]),
    "psk_count_kmers": (c.c_int, [c.c_void_p, c.c_int, c.c_char_p, c.c_size_t, _u64p, _u64p]),
    "psk_count_kmers_batch": (c.c_int, [c.c_void_p, c.c_int, c.c_int, c.POINTER(c.c_char_p), c.POINTER(c.c_size_t),
                                        c.c_void_p, c.c_void_p, c.c_int]),
    "psk_count_kmers_batch_sketch": (c.c_int, [c.c_void_p, c.c_int, c.c_int, c.POINTER(c.c_char_p), c.POINTER(c.c_size_t),
                                               c.c_void_p, c.c_void_p, c.c_int, c.c_int, c.c_int, c.c_uint32, c.c_void_p,
                                               c.c_void_p]),
    "psk_get_list": (c.c_int, [c.c_void_p, c.c_int, c.c_void_p, c.c_void_p, c.c_uint64]),
    "psk_lookup_counts": (c.c_int, [c.c_void_p, c.c_int, c.c_void_p, c.c_uint64, c.c_void_p]),
    "psk_write_result_tables": (c.c_int, [c.c_void_p, c.c_char_p, c.c_char_p, c.c_int64, c.c_char_p, c.c_int, c.c_int64, c.c_void_p,
                                          c.c_int, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_int,
                                          c.c_int, c.c_void_p, c.c_char_p, c.c_void_p, c.c_void_p]),
    "psk_write_model_coefficients": (c.c_int, [c.c_void_p, c.c_char_p, c.c_int64, c.c_char_p, c.c_void_p, c.c_void_p, c.c_void_p,
                                               c.c_int64, c.c_char_p, c.c_void_p]),
    "psk_lists_split": (c.c_int, [c.c_void_p, c.c_int, c.c_int, c.c_void_p, c.c_int, c.c_void_p]),
    "psk_gz_inflate": (c.c_int, [c.c_void_p, c.c_int, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p,
                                  c.POINTER(c.c_double)]),
    "psk_release_lists": (c.c_int, [c.c_void_p]),
    "psk_copy_list_ranges": (c.c_int, [c.c_void_p, c.c_int, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p]),
    "psk_set_lists_device": (c.c_int, [c.c_void_p, c.c_int, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p]),
    "psk_build_presence": (c.c_int, [c.c_void_p, _u64p]),
    "psk_presence_shape": (c.c_int, [c.c_void_p, _u64p, c.POINTER(c.c_int), c.POINTER(c.c_int)]),
    "psk_get_union": (c.c_int, [c.c_void_p, c.c_void_p, c.c_uint64]),
    "psk_get_rows": (c.c_int, [c.c_void_p, c.c_void_p, c.c_uint64, c.c_void_p]),
    "psk_intersect_db": (c.c_int, [c.c_void_p, c.c_void_p, c.c_uint64, _u64p]),
    "psk_set_presence": (c.c_int, [c.c_void_p, c.c_void_p, c.c_void_p, c.c_uint64, c.c_int, c.c_int]),
    "psk_synth_presence": (c.c_int, [c.c_void_p, c.c_uint64, c.c_int, c.c_uint64]),
    "psk_chi2_scan": (c.c_int, [c.c_void_p, c.c_void_p, c.c_void_p, c.c_int, c.c_int, c.c_double, c.c_int,
                                c.c_uint64, _u64p]),
    "psk_ttest_scan": (c.c_int, [c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_int, c.c_int, c.c_double,
                                 c.c_uint64, _u64p]),
    "psk_get_results": (c.c_int, [c.c_void_p] + [c.c_void_p] * 7 + [c.c_uint64]),
    "psk_export_survivors": (c.c_int, [c.c_void_p, c.c_void_p, c.c_uint64, _u64p]),
    "psk_last_scan_ms": (c.c_double, [c.c_void_p]),
    "psk_rescan_timed": (c.c_int, [c.c_void_p, c.c_int, c.POINTER(c.c_double)]),
    "psk_rescan_times": (c.c_int, [c.c_void_p, c.c_int, c.c_void_p]),
    "psk_stream_read_ceiling": (c.c_int, [c.c_void_p, c.c_int, c.POINTER(c.c_double), _u64p, c.POINTER(c.c_int)]),
    "psk_logreg_l1_fit": (c.c_int, [c.c_void_p, c.c_void_p, c.c_void_p, c.c_int, c.c_int, c.c_void_p, c.c_void_p,
                                    c.c_void_p, c.c_int, c.c_double, c.c_int, c.c_void_p, c.c_void_p, c.c_void_p]),
    "psk_lasso_fit": (c.c_int, [c.c_void_p, c.c_void_p, c.c_void_p, c.c_int, c.c_int, c.c_void_p, c.c_void_p,
                                c.c_void_p, c.c_int, c.c_double, c.c_int, c.c_void_p, c.c_void_p, c.c_void_p]),
    "psk_ridge_fit": (c.c_int, [c.c_void_p, c.c_void_p, c.c_void_p, c.c_int, c.c_int, c.c_void_p, c.c_void_p,
                                c.c_void_p, c.c_int, c.c_void_p, c.c_void_p, c.c_void_p]),
    "psk_logreg_l2_fit": (c.c_int, [c.c_void_p, c.c_void_p, c.c_void_p, c.c_int, c.c_int, c.c_void_p, c.c_void_p,
                                    c.c_void_p, c.c_int, c.c_double, c.c_int, c.c_int, c.c_void_p, c.c_void_p,
                                    c.c_void_p]),
    "psk_count_dict": (c.c_int, [c.c_void_p, c.c_char_p, c.c_size_t, c.c_int, c.c_void_p, c.c_uint64, c.c_void_p]),
    "psk_count_dict_batch": (c.c_int, [c.c_void_p, c.c_int, c.POINTER(c.c_char_p), c.POINTER(c.c_size_t), c.c_int, c.c_void_p,
                                       c.c_uint64, c.c_void_p, c.c_int]),
    "psk_count_dict_files": (c.c_int, [c.c_void_p, c.c_int, c.POINTER(c.c_char_p), c.POINTER(c.c_size_t), c.c_int, c.c_void_p,
                                       c.c_uint64, c.c_void_p, c.c_int]),
    "psk_minhash_sketch": (c.c_int, [c.c_void_p, c.c_char_p, c.c_size_t, c.c_int, c.c_int, c.c_uint32, c.c_void_p, _u64p]),
    "psk_mash_pairs": (c.c_int, [c.c_void_p, c.c_void_p, c.c_void_p, c.c_int, c.c_int, c.c_void_p, c.c_void_p]),
    "psk_nj_merges": (c.c_int, [c.c_void_p, c.c_void_p, c.c_int, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p]),
    "psk_export_survivors_async": (c.c_int, [c.c_void_p, c.c_void_p, c.c_uint64, c.c_void_p]),
    "psk_chi2_scan_begin": (c.c_int, [c.c_void_p, c.c_void_p, c.c_void_p, c.c_int, c.c_int, c.c_double, c.c_int, c.c_uint64]),
    "psk_scan_end": (c.c_int, [c.c_void_p, _u64p]),
    "psk_count_kmers_files": (c.c_int, [c.c_void_p, c.c_int, c.c_int, c.POINTER(c.c_char_p), c.POINTER(c.c_size_t),
                                        c.c_void_p, c.c_void_p, c.c_int, c.c_int, c.c_int, c.c_uint32, c.c_void_p,
                                        c.c_void_p]),
    "psk_frame_sequence": (c.c_int64, [c.c_char_p, c.c_size_t, c.c_void_p, c.c_size_t]),
    "psk_frame_sequence_gpu": (c.c_int64, [c.c_void_p, c.c_char_p, c.c_size_t, c.c_void_p, c.c_size_t]),
    "psk_device_count": (c.c_int, []),
    "psk_comm_unique_id": (c.c_int, [c.c_void_p, c.c_void_p, c.c_int]),
    "psk_comm_init": (c.c_int, [c.c_void_p, c.c_void_p, c.c_int, c.c_int, c.c_int]),
    "psk_comm_size": (c.c_int, [c.c_void_p]),
    "psk_comm_free": (c.c_int, [c.c_void_p]),
    "psk_comm_stream": (c.c_void_p, [c.c_void_p]),
    "psk_comm_sync": (c.c_int, [c.c_void_p]),
    "psk_comm_allreduce": (c.c_int, [c.c_void_p, c.c_void_p, c.c_int, c.c_int, c.c_int]),
    "psk_comm_allgather_host": (c.c_int, [c.c_void_p, c.c_void_p, c.c_void_p, c.c_uint64]),
    "psk_comm_allgather_device": (c.c_int, [c.c_void_p, c.c_void_p, c.c_void_p, c.c_uint64]),
    "psk_comm_alltoallv_device": (c.c_int, [c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_int]),
    "psk_dev_alloc": (c.c_int, [c.c_void_p, c.c_uint64, c.POINTER(c.c_void_p)]),
    "psk_dev_free": (c.c_int, [c.c_void_p, c.c_void_p]),
    "psk_dev_copy": (c.c_int, [c.c_void_p, c.c_void_p, c.c_void_p, c.c_uint64, c.c_int, c.c_int]),
}

_lib = None


def load():
    """Returns the bound library; raises PskError if libpsk.so is absent or unloadable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PskError("libpsk.so is missing (%s): build it with `python -c 'import __graft_entry__ as g; "
                       "g.build()'` or `make -C phenotypeseeker_amd/csrc`; there is no CPU fallback" % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise PskError("cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise PskError("libpsk.so does not export %s (stale build?)" % name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def exported_names():
    return sorted(_SIGNATURES)
