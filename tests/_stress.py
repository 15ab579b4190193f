#!/usr/bin/env python3
"""Randomised end-to-end check of the GPU path against the oracle (checker only): random sample sets (FASTA with
breaks / lower case / empty samples, some FASTQ), random k, random slab; lists, union, presence bits and the
chi2 survivors must be identical; also the batch sketches, two scans in flight, the weighted chi2, the Welch scan,
the dictionary counting of `prediction`, the list cut points of the multi-GPU ingest and (r05) samples that arrive
gzip-compressed (inflated on the device).  Test infrastructure (it links the oracle): run by
tests/test_gpu_parity.py::test_randomised_pipeline_against_oracle, or by hand: python tests/_stress.py SECONDS [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from oracle import oracle_weights as OW  # noqa: E402
from phenotypeseeker_amd.engine import PskContext  # noqa: E402
from phenotypeseeker_amd.synth import GenomeSet, fastq_reads  # noqa: E402

import gzip  # noqa: E402

os.environ.setdefault("PSK_GZ_DEVICE_MIN_MB", "0")     # (.gz samples, however small, through the device inflate)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
t_end = time.time() + budget
rounds = 0
with PskContext(0) as ctx:
    while time.time() < t_end:
        rounds += 1
        n = int(rng.integers(1, 70)) if rng.random() < 0.8 else int(rng.integers(70, 200))     # (beyond 64 / 128: several waves per merge range)
        length = int(rng.choice([300, 3000, 40_000, 150_000]))
        k = int(rng.choice([1, 2, 5, 9, 11, 13, 13, 13, 14, 16, 17, 18, 21, 24, 27, 31, 32]))
        gs = GenomeSet(n, length, seed=int(rng.integers(1, 1 << 30)), gene_len=min(200, length // 3))
        datas = []
        for i in range(n):
            roll = rng.random()
            if roll < 0.05:
                datas.append(b"")
                continue
            if roll < 0.15:
                datas.append(fastq_reads(gs.codes(i), int(rng.integers(1, 60)), 150, seed=[3, i]))
                continue
            b = bytearray(gs.sample(i)[1])
            for pos in rng.integers(10, len(b), int(rng.integers(0, 30))):
                if b[pos] != 10:
                    b[pos] = ord("N") if pos % 3 else ord("a")
            datas.append(bytes(b))
        space = 1 << (2 * k)
        if rng.random() < 0.5 or space < 8:
            lo, hi = 0, 0
        else:
            world = int(rng.integers(2, 6))
            rank = int(rng.integers(0, world))
            lo, hi = (space * rank) // world, (0 if rank == world - 1 else (space * (rank + 1)) // world)
        ctx.begin(k, n, lo, hi)
        do_sketch = length <= 3000 and rng.random() < 0.5   # the oracle's sketch is a pure-Python loop
        sk_par = (int(rng.choice([21, 16, 11])), int(rng.choice([1000, 100, 30])), 42)
        # some samples arrive gzip-compressed (r05: inflated on the device; the lists are those of the text)
        sent = [gzip.compress(d, int(rng.choice([1, 6, 9]))) if rng.random() < 0.15 else d for d in datas]
        if rng.random() < 0.3:
            os.environ["PSK_GZ_CHUNK"] = str(int(rng.choice([2048, 8192, 30000])))
        else:
            os.environ.pop("PSK_GZ_CHUNK", None)
        if rng.random() < 0.3:      # the call cut into runs of ~1 MB of text: read | inflated | counted as a pipeline of threads
            os.environ["PSK_GZ_GROUP_MB"] = "1"
        else:
            os.environ.pop("PSK_GZ_GROUP_MB", None)
        if do_sketch:
            nu, nt, sks = ctx.count_kmers_batch(0, sent, int(rng.integers(1, 9)), sketch=sk_par)
            for i in rng.choice(n, min(n, 3), replace=False):
                want = OW.sketch(datas[i], k=sk_par[0], sketch_size=sk_par[1]) if not datas[i].startswith(b"@") else None
                if want is not None:
                    assert sks[i].tolist() == want, ("sketch", rounds, int(i), sk_par)
        else:
            nu, nt = ctx.count_kmers_batch(0, sent, int(rng.integers(1, 9)))
        ref_lists = []
        for i in range(n):
            w, f = ctx.get_list(i, nu[i])
            ow, of = O.count_kmers(datas[i], k)[:2]
            sel = (ow >= lo) & ((ow < hi) if hi else np.ones(len(ow), bool))
            assert np.array_equal(w, ow[sel]) and np.array_equal(f, of[sel]), ("list", rounds, i, k, lo, hi)
            ref_lists.append(ow[sel])
        if rounds % 4 == 0 and n >= 2:
            # prediction's dictionary counting: a dictionary drawn from two samples' own words plus absent ones,
            # counted in a third sample (distinct canonical words: duplicates corrupt gmer_counter's counts too)
            pool = np.unique(np.concatenate([O.count_kmers(datas[0], k)[0][:40], O.count_kmers(datas[n - 1], k)[0][-40:],
                                             rng.integers(0, 1 << min(2 * k, 62), 20).astype(np.uint64)]))
            pool = np.array(sorted({int(O.canonical_word(int(w), k)) for w in pool}), dtype=np.uint64)
            if len(pool) >= 3:
                who = int(rng.integers(0, n))
                assert np.array_equal(ctx.count_dict(datas[who], k, pool), O.count_dict(datas[who], k, pool)), ("dict", rounds, k)
        if rounds % 3 == 0 and n >= 2 and not (lo or hi):
            # the multi-GPU ingest's cut points: psk_lists_split against searchsorted on the lists just checked
            world = int(rng.integers(2, 6))
            if world <= space:
                bounds = [(space * d) // world for d in range(world)] + [0]
                cuts = ctx.lists_split(0, n, bounds)
                for i in range(n):
                    want = [int(np.searchsorted(ref_lists[i], np.uint64(b))) for b in bounds[:-1]] + [len(ref_lists[i])]
                    assert cuts[i].tolist() == want, ("split", rounds, i, k)
        m = ctx.build_presence()
        uw = O.union(ref_lists)
        assert m == len(uw), ("union size", rounds, k, m, len(uw))
        if m == 0:
            continue
        assert np.array_equal(ctx.get_union(), uw), ("union", rounds)
        wpr = ctx.presence_shape()[1]
        bits = O.presence_bits(ref_lists, uw, wpr=wpr)
        assert np.array_equal(ctx.get_rows(np.arange(m, dtype=np.uint64)), bits), ("bits", rounds, k, n)
        if n >= 4:
            ph = rng.integers(-1, 2, n).astype(np.int8)
            omit = bool(rng.random() < 0.5)
            npass = ctx.chi2_scan(ph, None, 2, n - 2, 0.05, omit, m)
            ref = O.chi2_scan(bits, ph.tolist(), np.ones(n), n, 2, n - 2, 0.05, omit, m)
            keep = np.nonzero(ref["keep"])[0]
            res = ctx.get_results(npass)
            assert np.array_equal(np.sort(res["row"]), keep.astype(np.uint64)), ("scan rows", rounds, k, n)
            order = np.argsort(res["row"])
            assert np.array_equal(res["stat"][order], ref["stat"][keep]), ("scan stat", rounds)
            # two scans in flight (two result sets): unweighted again + a weighted one, ended in launch order
            wts = np.round(rng.uniform(0.2, 3.0, n), 6)
            ph2 = rng.integers(-1, 2, n).astype(np.int8)
            ctx.chi2_scan_begin(ph, None, 2, n - 2, 0.05, omit, m)
            ctx.chi2_scan_begin(ph2, wts, 1, n - 1, 0.2, True, m)
            c1 = ctx.scan_end()
            r1 = ctx.get_results(c1)
            assert np.array_equal(r1["row"], res["row"]) and np.array_equal(r1["stat"], res["stat"]), ("pipelined scan", rounds)
            c2 = ctx.scan_end()
            r2 = ctx.get_results(c2)
            refw = O.chi2_scan(bits, ph2.tolist(), wts, n, 1, n - 1, 0.2, True, m)
            keepw = np.nonzero(refw["keep"])[0]
            gotw = dict(zip(r2["row"].tolist(), range(c2)))
            both = [r for r in keepw.tolist() if r in gotw]   # a p within rounding of the cut-off may fall either way
            assert len(both) >= len(keepw) - 2 and c2 - len(both) <= 2, ("weighted rows", rounds, len(keepw), c2)
            gi = [gotw[r] for r in both]
            assert np.allclose(r2["stat"][gi], refw["stat"][both], rtol=1e-9, atol=1e-12), ("weighted stat", rounds)
            # Welch t-test on the same matrix
            vals = np.round(rng.normal(3.0, 1.5, n), 4)
            okv = rng.random(n) > 0.1
            if okv.sum() >= 4:
                pheno = [float(v) if o else "NA" for v, o in zip(vals, okv)]
                tw = wts if rng.random() < 0.5 else np.ones(n)
                reft = O.ttest_scan(bits, pheno, tw, n, 2, n - 2, 0.5, 1)
                ct = ctx.ttest_scan(vals, okv, tw, 2, n - 2, 0.5, 1)
                rt = ctx.get_results(ct)
                keept = np.nonzero(reft["keep"])[0]
                gott = dict(zip(rt["row"].tolist(), range(ct)))
                botht = [r for r in keept.tolist() if r in gott]
                assert len(botht) >= len(keept) - 2 and ct - len(botht) <= 2, ("t rows", rounds, len(keept), ct)
                gi = [gott[r] for r in botht]
                assert np.allclose(rt["stat"][gi], reft["stat"][botht], rtol=1e-7, atol=1e-10), ("t stat", rounds)
                assert np.array_equal(rt["n_with"][gi], reft["n_with"][botht]), ("t n_with", rounds)
print("stress ok: %d rounds in %.0f s (seed %d)" % (rounds, budget, seed))
