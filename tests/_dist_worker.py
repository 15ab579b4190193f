"""Worker for tests/test_dist_gloo.py (one rank of a world_size-2 gloo group on CPU).
Each rank plays one slab of the range-sharded path with the oracle standing in for the GPU
engine (no GPU in the CPU suite); what is under test is phenotypeseeker_amd.dist: slab bounds,
the all-reduce of the union size and the all-gather(v)/merge of the per-slab survivors."""
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
    bits = O.presence_bits(wl, uw, wpr=(((n + 63) // 64) + 1) & ~1)
    if m_global is None:
        return len(uw), None, None
    res = O.chi2_scan(bits, ds["pheno"], np.ones(n), n, 2, n - 2, 0.05, True, m_global)
    keep = np.nonzero(res["keep"])[0]
    out = {"word": uw[keep], "stat": res["stat"][keep], "p": res["p"][keep], "mean_x": np.zeros(len(keep)),
           "mean_y": np.zeros(len(keep)), "n_with": res["n_with"][keep]}
    return len(uw), out, bits[keep]


def main():
    out_path = sys.argv[1]
    grp = dist.Group().init("gloo")
    ds = load_dataset("ds_omitB")
    k = ds["meta"]["k"]
    lo, hi = dist.slab_bounds(k, grp.world, grp.rank)
    m_local, _, _ = slab_result(ds, lo, hi, None)
    m_global = grp.allreduce_sum(int(m_local))
    _, res, bits = slab_result(ds, lo, hi, m_global)
    merged, mbits = dist.merge_candidates(grp.allgather_bytes(dist.pack_candidates(res, bits)))
    tmax = grp.allreduce_max(float(grp.rank + 1))
    grp.barrier()
    if grp.rank == 0:
        np.savez(out_path, m_global=m_global, word=merged["word"], stat=merged["stat"], p=merged["p"],
                 n_with=merged["n_with"], bits=mbits, tmax=tmax, world=grp.world)
    grp.close()


if __name__ == "__main__":
    main()
