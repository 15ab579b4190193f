"""Worker for tests/test_dist_gloo.py (one rank of a world_size-2 gloo group on CPU).
Each rank plays one slab of the range-sharded path with the oracle standing in for the GPU
engine (no GPU in the CPU suite); what is under test is phenotypeseeker_amd.dist: slab bounds,
the all-reduce of the union size, the all-gather(v)/merge of the per-slab survivors, and the all-to-all of list
ranges of the sample-parallel ingest (dist.ListExchange, host stand-ins for the two contexts)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import load_dataset  # noqa: E402
from oracle import oracle as O  # noqa: E402
from phenotypeseeker_amd import dist  # noqa: E402


def slab_result(ds, lo, hi, m_global):
    k, names, n = ds["meta"]["k"], ds["names"], len(ds["names"])
    wl = []
    for nm in names:
        w = O.count_kmers(ds["files"][nm], k)[0]
        wl.append(w[(w >= lo) & ((w < hi) if hi else np.ones(len(w), bool))])
    uw = O.union(wl)
    bits = O.presence_bits(wl, uw)
    if m_global is None:
        return len(uw), None, None
    res = O.chi2_scan(bits, ds["pheno"], np.ones(n), n, 2, n - 2, 0.05, True, m_global)
    keep = np.nonzero(res["keep"])[0]
    out = {"word": uw[keep], "stat": res["stat"][keep], "p": res["p"][keep], "mean_x": np.zeros(len(keep)),
           "mean_y": np.zeros(len(keep)), "n_with": res["n_with"][keep]}
    return len(uw), out, bits[keep]


class HostLists:
    """Stands in for the counting context of dist.ListExchange: sorted lists in host memory."""

    def __init__(self, lists):
        self.lists = lists

    def lists_split(self, first, n, bounds):
        out = np.zeros((n, len(bounds)), dtype=np.int64)
        for i in range(n):
            w = self.lists[first + i][0]
            for b, key in enumerate(bounds):
                out[i, b] = len(w) if (b > 0 and key == 0) else int(np.searchsorted(w, np.uint64(key)))
        return out

    def copy_list_ranges(self, sample_idx, start, count, words_ptr, freqs_ptr):
        import ctypes
        off = 0
        for j, st, c in zip(sample_idx, start, count):
            w, f = self.lists[j]
            ctypes.memmove(words_ptr + 8 * off, np.ascontiguousarray(w[st:st + c]).ctypes.data, c * 8)
            ctypes.memmove(freqs_ptr + 4 * off, np.ascontiguousarray(f[st:st + c]).ctypes.data, c * 4)
            off += c


class HostSlab:
    """Stands in for the slab context: keeps what psk_set_lists_device would install."""

    def __init__(self):
        self.got = {}

    def set_lists_device(self, sample_idx, count, n_total, words_ptr, freqs_ptr):
        import ctypes
        off = 0
        for i, n, tot in zip(sample_idx, count, n_total):
            w = np.ctypeslib.as_array((ctypes.c_uint64 * n).from_address(words_ptr + 8 * off)).copy() if n else np.zeros(0, np.uint64)
            f = np.ctypeslib.as_array((ctypes.c_uint32 * n).from_address(freqs_ptr + 4 * off)).copy() if n else np.zeros(0, np.uint32)
            self.got[i] = (w, f, tot)
            off += n


def check_list_exchange(grp, ds):
    """dist.ListExchange over gloo: every sample's list lives on ONE rank, after the all-to-all each rank must hold
    the slab range of every sample's list."""
    k, names, n = ds["meta"]["k"], ds["names"], len(ds["names"])
    full = [O.count_kmers(ds["files"][nm], k)[:2] for nm in names]
    own = [i for i in range(n) if dist.owner_of(i, grp.world) == grp.rank]
    slab = HostSlab()
    bounds = dist.balanced_bounds(grp, k, [full[i][0] for i in own[:4]])
    pairs = dist.ListExchange(grp, k, bounds).run(HostLists([full[i] for i in own]), slab, n, [int(full[i][1].sum()) for i in own])
    lo, hi = bounds[grp.rank], bounds[grp.rank + 1]
    assert sorted(slab.got) == list(range(n))
    for i in range(n):
        w, f = full[i]
        sel = (w >= lo) & ((w < hi) if hi else np.ones(len(w), bool))
        assert np.array_equal(slab.got[i][0], w[sel]) and np.array_equal(slab.got[i][1], f[sel]), i
        assert slab.got[i][2] == int(f.sum())
    assert pairs == sum(len(v[0]) for v in slab.got.values())
    return pairs, bounds


def main():
    out_path = sys.argv[1]
    grp = dist.Group().init()       # PSK_DIST_TRANSPORT names the gloo transport of tests/
    assert grp.backend == "gloo"
    ds = load_dataset(os.environ.get("PSK_TEST_DATASET", "ds_omitB"))
    k = ds["meta"]["k"]
    pairs, bounds = check_list_exchange(grp, ds)
    lo, hi = bounds[grp.rank], bounds[grp.rank + 1]
    m_local, _, _ = slab_result(ds, lo, hi, None)
    shares = grp.allgather_i64(np.array([m_local]))[:, 0]
    m_global = grp.allreduce_sum(int(m_local))
    _, res, bits = slab_result(ds, lo, hi, m_global)
    merged, mbits = dist.merge_candidates(grp.allgather_bytes(dist.pack_candidates(res, bits)))
    pairs = grp.allreduce_sum(int(pairs))
    tmax = grp.allreduce_max(float(grp.rank + 1))
    grp.barrier()
    if grp.rank == 0:
        np.savez(out_path, m_global=m_global, word=merged["word"], stat=merged["stat"], p=merged["p"],
                 n_with=merged["n_with"], bits=mbits, tmax=tmax, world=grp.world, pairs=pairs, shares=shares,
                 bounds=np.array(bounds, dtype=np.uint64))
    grp.close()


if __name__ == "__main__":
    main()
