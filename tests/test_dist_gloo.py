"""N > 1 path on CPU: two gloo ranks, each owning one slab of the k-mer word space, must
reproduce the single-rank result byte for byte (SURVEY.md 8(e) invariant)."""
import os
import subprocess
import sys

import numpy as np

from helpers import ROOT, load_dataset


def test_slab_bounds_tile_the_word_space():
    from phenotypeseeker_amd.dist import slab_bounds
    for k in (1, 5, 13, 16, 31, 32):
        for world in (1, 2, 3, 4, 8):
            if world > 4 ** k:
                continue
            edges = [slab_bounds(k, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == 0
            for (lo, hi), (lo2, _) in zip(edges[:-1], edges[1:]):
                assert hi == lo2 and hi > lo
            assert all(lo < (1 << 64) and hi < (1 << 64) for lo, hi in edges)


def test_pack_merge_round_trip():
    from phenotypeseeker_amd import dist
    rng = np.random.default_rng(0)
    parts = []
    for n in (0, 3, 5):
        res = {"word": np.sort(rng.integers(0, 1 << 40, n)).astype(np.uint64), "stat": rng.random(n), "p": rng.random(n),
               "mean_x": rng.random(n), "mean_y": rng.random(n), "n_with": rng.integers(0, 99, n).astype(np.int32)}
        parts.append((res, rng.integers(0, 1 << 62, (n, 4)).astype(np.uint64)))
    merged, bits = dist.merge_candidates([dist.pack_candidates(r, b) for r, b in parts])
    assert len(merged["word"]) == 8 and bits.shape == (8, 4)
    assert np.array_equal(merged["stat"], np.concatenate([p[0]["stat"] for p in parts]))
    assert np.array_equal(bits, np.concatenate([p[1] for p in parts]))


def test_two_rank_gloo_run_equals_single_rank(tmp_path, oracle):
    out = os.path.join(tmp_path, "merged.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29617", os.path.join(ROOT, "tests", "_dist_worker.py"), out]
    r = subprocess.run(cmd, env=env, cwd=ROOT, timeout=300, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    z = np.load(out)
    assert int(z["world"]) == 2 and float(z["tmax"]) == 2.0
    ds = load_dataset("ds_omitB")
    k, names, n = ds["meta"]["k"], ds["names"], len(ds["names"])
    wl = [oracle.count_kmers(ds["files"][nm], k)[0] for nm in names]
    uw = oracle.union(wl)
    assert int(z["m_global"]) == len(uw) == ds["meta"]["n_union"]
    assert int(z["pairs"]) == sum(len(w) for w in wl)   # the list exchange moved every (word, sample) pair once
    bits = oracle.presence_bits(wl, uw, wpr=z["bits"].shape[1])
    ref = oracle.chi2_scan(bits, ds["pheno"], np.ones(n), n, 2, n - 2, 0.05, True, len(uw))
    keep = np.nonzero(ref["keep"])[0]
    assert np.array_equal(z["word"], uw[keep])
    assert np.array_equal(z["stat"], ref["stat"][keep]) and np.array_equal(z["p"], ref["p"][keep])
    assert np.array_equal(z["n_with"], ref["n_with"][keep]) and np.array_equal(z["bits"], bits[keep])
