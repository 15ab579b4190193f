"""-w path.  MinHash sketches and Mash distances against the bundled mash binary's output (tests/golden/mash.json);
the distance-matrix plumbing and the GSC recursion against what the REFERENCE'S OWN functions returned
(tests/golden/gsc_kat.json: modeling.py:415-444 and :461-503 run through oracle/ref_shim.py, generator
oracle/gen_golden.py::gen_gsc_kat); neighbour joining against the oracle's plain O(n^3) restatement (orc_nj) --
Biopython / ete3 themselves are absent, so the joins' tie-breaking, the "%1.5f" newick text and the parser's defaults
stay PARITY UNPINNED."""
import base64
import gzip
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN


@pytest.fixture(scope="module")
def mash():
    with open(os.path.join(GOLDEN, "mash.json")) as f:
        d = json.load(f)
    for s in d["samples"]:
        s["fasta"] = gzip.decompress(base64.b64decode(s["fasta_gz_b64"]))
    return d


def test_oracle_sketch_equals_mash(mash):
    from oracle import oracle_weights as OW
    for s in mash["samples"]:
        assert OW.sketch(s["fasta"]) == s["hashes"], s["name"]
    assert OW.sketch(mash["samples"][0]["fasta"], k=17, sketch_size=50) == mash["k17_s50_hashes"]
    assert mash["k15_bits"] == 32
    assert OW.sketch(mash["samples"][0]["fasta"], k=15, sketch_size=40) == mash["k15_s40_hashes"]


def test_mash_distance_matches_mash_dist(mash):
    from phenotypeseeker_amd import weights as W
    sk = {s["name"] + ".fasta": s["hashes"] for s in mash["samples"]}
    n = 0
    for line in mash["dist_table"].strip().splitlines():
        ref, qry, dist, _, shared = line.split("\t")
        d, common, denom = W.mash_distance(sk[ref], sk[qry], 21, 1000)
        assert "%d/%d" % (common, denom) == shared
        assert d == float(dist), (ref, qry)
        n += 1
    assert n == 36


def test_distance_matrix_reproduces_the_glob_order_labelling():
    from phenotypeseeker_amd import weights as W
    sk = {"b": [1, 2, 3, 4], "a": [1, 2, 3, 9], "c": [5, 6, 7, 8]}
    labels, mat = W.distance_matrix(["b", "a", "c"], sk, k=21, sketch_size=4)
    assert labels == ["b", "a", "c"]           # labels in data.pheno order ...
    d_ab = W.mash_distance(sk["a"], sk["b"], 21, 4)[0]
    assert mat[0][1] == d_ab and mat[0][2] == 1.0 and mat[1][2] == 1.0   # ... rows in file-name order a, b, c


@pytest.fixture(scope="module")
def gsc_kat():
    with open(os.path.join(GOLDEN, "gsc_kat.json")) as f:
        return json.load(f)


def _rnd_matrix(n, rng, kind):
    if kind == "ties":
        half = rng.choice([0.0, 0.001, 0.002, 0.0153, 1.0], (n, n))
    else:
        half = np.round(rng.random((n, n)) * 0.1, 6)
    mat = np.tril(half, -1)
    return mat + mat.T


def test_distance_plumbing_equals_the_reference(gsc_kat):
    """distances.mat byte for byte and the lower triangle handed to the tree builder `==` what
    Samples._mash_output_to_distance_matrix / _distance_matrix_modifier (modeling.py:415-444) produced from the real
    `mash dist` table -- including the set whose pheno order is not the glob order (permuted labels, the reference's
    behaviour) and whose names order differently with the '.msh' suffix."""
    from phenotypeseeker_amd import weights as W
    assert {r["tag"] for r in gsc_kat["plumbing"]} >= {"sorted6", "shuffled9", "ds_omitB"}
    for r in gsc_kat["plumbing"]:
        labels, mat = W.distance_matrix(r["names"], r["hashes"])
        assert W.distances_mat_text(labels, mat) == r["distances_mat"], r["tag"]
        assert W.lower_triangle(mat) == r["lower_triangle"], r["tag"]
        # the third column of the mash table, row-major, is the matrix
        col3 = [float(l.split("\t")[2]) for l in r["mash_distances_mat"].strip().split("\n")]
        assert np.array_equal(np.asarray(mat, dtype=np.float64).ravel(), np.array(col3)), r["tag"]
    shuffled = next(r for r in gsc_kat["plumbing"] if r["tag"] == "shuffled9")
    assert W.glob_order(shuffled["names"]) != sorted(shuffled["names"]) != shuffled["names"]


def test_gsc_weights_equal_the_reference_bit_for_bit(gsc_kat):
    """Samples.GSC_weights_from_newick(normalize='mean1') (modeling.py:461-503) on 60 trees -- 2 ... 200 leaves, a
    three-child root, zero / negative / huge lengths (the clip), caterpillars: the same doubles (the tolerance the verdict
    asked for is 1e-12 relative; the sums are taken in the reference's order, so it is 0)."""
    from phenotypeseeker_amd import weights as W
    assert len(gsc_kat["gsc"]) >= 20
    sizes = set()
    for c in gsc_kat["gsc"]:
        got = W.gsc_weights(W.from_newick(c["newick"]))
        assert got == c["weights"], c["note"]
        assert list(got) == list(c["weights"])          # leaves in the reference's iteration order
        sizes.add(len(got))
    assert min(sizes) == 2 and max(sizes) == 200


def test_whole_weight_chain_equals_the_fixture(gsc_kat):
    """hashes -> distances -> neighbour joining -> newick -> GSC: the tree text and the weights of the three genome
    sets equal the chain [reference plumbing -> oracle NJ + newick -> reference GSC] of the fixture; and the two files a
    `-w` run leaves behind are the reference's."""
    import tempfile
    from phenotypeseeker_amd import weights as W
    for r in gsc_kat["plumbing"]:
        c = gsc_kat["chains"][r["tag"]]
        with tempfile.TemporaryDirectory() as tmp:
            w, tree = W.weights_from_sketches(r["names"], r["hashes"], files_dir=tmp)
            assert open(os.path.join(tmp, "distances.mat")).read() == r["distances_mat"]
            assert open(os.path.join(tmp, "tree_newick.txt")).read() == c["newick"] + "\n"
        assert W.to_newick(tree) == c["newick"], r["tag"]
        assert w == c["weights"], r["tag"]


def test_host_neighbour_joining_equals_the_oracle():
    """weights.nj (numpy, the host form of the product) against orc_nj (plain C loops, oracle/psk_oracle.c section 7):
    pairs, branch lengths and the last distance as written to newick, with ties everywhere and without."""
    from oracle import oracle_weights as OW
    from phenotypeseeker_amd import weights as W
    rng = np.random.default_rng(12)
    for n in (3, 4, 5, 6, 9, 17, 33, 64, 130, 257):
        for kind in ("plain", "ties"):
            mat = _rnd_matrix(n, rng, kind)
            names = ["s%d" % i for i in range(n)]
            assert W.to_newick(W.newick_round_trip(W.nj(names, mat.tolist()))) == OW.nj_newick(names, mat), (n, kind)


def test_newick_reader_round_trips_and_rejects_garbage():
    from phenotypeseeker_amd import weights as W
    t = W.from_newick("((A:1,B:2)Inner1:1.5,(C:1,D:3)Inner2:0.5)Inner3:0.00000;\n")
    assert W.to_newick(t) == "((A:1.00000,B:2.00000)Inner1:1.50000,(C:1.00000,D:3.00000)Inner2:0.50000)Inner3:0.00000;"
    assert [c.name for c in t.children] == ["Inner1", "Inner2"] and t.children[0].up is t
    for bad in ("(A:1,B:2)", "(A:1,B:2));", "((A:1,B:2);", "A:1,B:2;"):
        with pytest.raises(ValueError):
            W.from_newick(bad)


def _tree_leaves(root):
    from phenotypeseeker_amd.weights import _walk
    return [n for n in _walk(root) if not n.children]


def test_nj_recovers_additive_tree_and_gsc_properties():
    from phenotypeseeker_amd import weights as W
    # additive distances of the tree ((A:1,B:2):1.5,(C:1,D:3):0.5)
    names = ["A", "B", "C", "D"]
    D = [[0, 3, 4, 6], [3, 0, 5, 7], [4, 5, 0, 4], [6, 7, 4, 0]]
    root = W.nj(names, [[float(x) for x in r] for r in D])
    leaves = {n.name: n for n in _tree_leaves(root)}
    assert set(leaves) == set(names)

    def path(a, b):
        def up(n):
            out = []
            while n is not None:
                out.append(n)
                n = n.up
            return out
        pa, pb = up(leaves[a]), up(leaves[b])
        common = next(x for x in pa if x in pb)
        return sum(x.dist for x in pa[:pa.index(common)]) + sum(x.dist for x in pb[:pb.index(common)])
    for i, a in enumerate(names):
        for j, b in enumerate(names):
            if i < j:
                assert path(a, b) == pytest.approx(D[i][j])
    w = W.gsc_weights(W.newick_round_trip(root))
    assert sum(w.values()) == pytest.approx(4.0)                  # mean 1
    assert w["D"] > w["C"] and w["B"] > w["A"]                    # longer private branch, larger weight
    assert W.to_newick(root).endswith(";") and W.to_newick(root).count("(") >= 2


def test_gsc_identical_samples_share_weight_and_negative_branches_clip():
    from phenotypeseeker_amd import weights as W
    names = ["s1", "s2", "s3", "s4", "s5"]
    sk = {"s1": list(range(0, 1000)), "s2": list(range(0, 1000)), "s3": list(range(100, 1100)),
          "s4": list(range(500, 1500)), "s5": list(range(5000, 6000))}
    w, tree = W.weights_from_sketches(names, sk)
    assert sum(w.values()) == pytest.approx(5.0)
    assert w["s1"] == pytest.approx(w["s2"], rel=1e-6) and w["s5"] > w["s1"]
    from phenotypeseeker_amd.weights import _walk
    assert all(n.dist >= 1e-9 for n in _walk(tree))


@pytest.mark.gpu
def test_gpu_sketch_equals_mash(mash):
    from phenotypeseeker_amd.engine import PskContext
    with PskContext(0) as ctx:
        for s in mash["samples"]:
            got = ctx.minhash_sketch(s["fasta"])
            assert got.tolist() == s["hashes"], s["name"]
        assert ctx.minhash_sketch(mash["samples"][0]["fasta"], k=17, sketch_size=50).tolist() == mash["k17_s50_hashes"]
        assert ctx.minhash_sketch(mash["samples"][0]["fasta"], k=15, sketch_size=40).tolist() == mash["k15_s40_hashes"]
        assert len(ctx.minhash_sketch(b">e\nACGT\n")) == 0


@pytest.mark.gpu
def test_gpu_pairwise_distances_equal_mash_and_host(mash):
    """psk_mash_pairs: the sketch merges of `mash dist` for every pair at once.  Against the distances the
    bundled mash binary printed (tests/golden/mash.json) and against the host loop on short / empty /
    overlapping sketches (the merge's early-exhaustion rule)."""
    import numpy as np
    from phenotypeseeker_amd import weights as W
    from phenotypeseeker_amd.engine import PskContext
    names = [s["name"] for s in mash["samples"]]
    sk = {s["name"]: s["hashes"] for s in mash["samples"]}
    with PskContext(0) as ctx:
        labels, mat = W.distance_matrix(names, sk, ctx=ctx)
        labels_h, mat_h = W.distance_matrix(names, sk)
        assert labels == labels_h and np.array_equal(np.array(mat), np.array(mat_h))
        rng = np.random.default_rng(4)
        pool = np.unique(rng.integers(1, 1 << 40, 5000).astype(np.uint64))
        odd = {"e": [], "one": [int(pool[3])], "few": sorted(int(v) for v in rng.choice(pool, 40, replace=False))}
        for i in range(9):
            odd["r%d" % i] = sorted(int(v) for v in rng.choice(pool, int(rng.integers(50, 1000)), replace=False))
        odd["dup_of_r0"] = list(odd["r0"])
        nm = list(odd)
        _, got = W.distance_matrix(nm, odd, k=21, sketch_size=1000, ctx=ctx)
        _, want = W.distance_matrix(nm, odd, k=21, sketch_size=1000)
        assert np.array_equal(np.array(got), np.array(want))
        common, denom = ctx.mash_pairs([odd[n] for n in sorted(nm)], 1000)
        for a, na in enumerate(sorted(nm)):
            for b, nb in enumerate(sorted(nm)):
                _, c, d = W.mash_distance(odd[na], odd[nb], 21, 1000)
                assert (int(common[a, b]), int(denom[a, b])) == (c, d), (na, nb)


@pytest.mark.gpu
def test_gpu_batch_sketch_equals_single_sketch(mash):
    """psk_count_kmers_batch_sketch: sketches from the clean stream already on the device equal psk_minhash_sketch
    (and therefore mash), also for an empty sample and one shorter than k."""
    import numpy as np
    from phenotypeseeker_amd.engine import PskContext
    datas = [s["fasta"] for s in mash["samples"]] + [b"", b">tiny\nACGTACGT\n"]
    with PskContext(0) as ctx:
        ctx.begin(13, len(datas))
        nu, nt, sk = ctx.count_kmers_batch(0, datas, 3, sketch=(21, 1000, 42))
        for i, s in enumerate(mash["samples"]):
            assert sk[i].tolist() == s["hashes"], s["name"]
        assert len(sk[-1]) == 0 and len(sk[-2]) == 0
        nu2, nt2 = ctx.count_kmers_batch(0, datas, 3)
        assert (nu, nt) == (nu2, nt2)
        one = ctx.minhash_sketch(datas[0], k=17, sketch_size=50)
        ctx.begin(13, 1)
        _, _, sk17 = ctx.count_kmers_batch(0, datas[:1], 1, sketch=(17, 50, 42))
        assert np.array_equal(sk17[0], one)
        # the batch path's one-workgroup candidate sort against the general route (psk_minhash_sketch) where it has
        # to give up: a 40-kbp unit repeated 30 times (the candidates hold fewer than s distinct hashes), a sketch
        # size whose candidates do not fit it, samples around the size where filtering starts, N runs
        rng = np.random.default_rng(8)
        def genome(n):
            return b">g\n" + bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), n)) + b"\n"
        unit = genome(40_000)[3:-1]
        odd = [b">rep\n" + unit * 30 + b"\n", genome(1_500_000), genome(11_000), genome(13_000), genome(2_000),
               b">n\n" + unit[:30_000] + b"N" * 5000 + unit[30_000:] + b"\n"]
        for kk, ss in ((21, 1000), (21, 2000), (16, 300), (32, 1000)):
            ctx.begin(13, len(odd))
            _, _, got = ctx.count_kmers_batch(0, odd, 3, sketch=(kk, ss, 42))
            for i, d in enumerate(odd):
                want = ctx.minhash_sketch(d, k=kk, sketch_size=ss)
                assert np.array_equal(got[i], want), (kk, ss, i)


@pytest.mark.gpu
def test_gpu_neighbour_joining_is_bit_identical_to_the_oracle_and_the_host_loop():
    """psk_nj_merges against the ORACLE's neighbour joining (orc_nj: plain C loops, nothing of the product in it) and
    against weights.nj on the host: random matrices (Mash-like 6-digit values), matrices made of a few repeated values
    (ties everywhere), sizes that cross the 1024-thread workgroup."""
    from oracle import oracle_weights as OW
    import random
    import time
    from phenotypeseeker_amd import weights as W
    from phenotypeseeker_amd.engine import PskContext

    def rnd(n, seed, ties):
        rng = random.Random(seed)
        m = [[0.0] * n for _ in range(n)]
        for i in range(n):
            for j in range(i):
                v = float("%g" % (rng.random() * 0.1)) if not ties else rng.choice([0.0, 0.001, 0.002, 0.0153, 1.0])
                m[i][j] = m[j][i] = v
        return m
    with PskContext(0) as ctx:
        for n in (3, 4, 5, 9, 33, 130):
            for seed in range(4):
                for ties in (False, True):
                    names = ["s%d" % i for i in range(n)]
                    mat = rnd(n, seed, ties)
                    a = W.to_newick(W.newick_round_trip(W.nj(names, mat)))
                    b = W.to_newick(W.newick_round_trip(W.nj(names, mat, ctx)))
                    assert a == b == OW.nj_newick(names, mat), (n, seed, ties)
                    for u, v in zip(ctx.nj_merges(mat), OW.nj_merges(mat)):
                        assert np.array_equal(np.asarray(u), np.asarray(v)), (n, seed, ties)
                    assert W.gsc_weights(W.newick_round_trip(W.nj(names, mat))) == \
                        W.gsc_weights(W.newick_round_trip(W.nj(names, mat, ctx)))
        n = 1100
        names = ["s%d" % i for i in range(n)]
        mat = rnd(n, 5, False)
        t = time.time()
        b = W.nj(names, mat, ctx)
        t_gpu = time.time() - t
        a = W.nj(names, mat)
        assert W.to_newick(a) == W.to_newick(b)
        for u, v in zip(ctx.nj_merges(mat), OW.nj_merges(mat)):
            assert np.array_equal(np.asarray(u), np.asarray(v))
        print("nj 1100 leaves: gpu path %.2f s" % t_gpu)


@pytest.mark.gpu
def test_gpu_neighbour_joining_on_many_workgroups_equals_the_one_workgroup_kernel(monkeypatch):
    """r04: nj_grid_kernel (a workgroup per 64 columns of the matrix, three exchanges per join through agent-scope atomics;
    what psk_nj_merges runs from 512 leaves on, forced here for the small sizes) against nj_kernel (PSK_NJ_ONE_WG=1): the
    merge lists -- pairs, both branch lengths, the last distance -- bit for bit, at sizes on either side of the 64-column
    blocks and of the load batches, with ties everywhere and without; and faster where it is the default (1,024 leaves:
    67 against 92 ms; 2,500: 0.4 against 1.2 s)."""
    import time
    from oracle import oracle_weights as OW
    from phenotypeseeker_amd.engine import PskContext
    rng = np.random.default_rng(77)
    monkeypatch.setenv("PSK_NJ_GRID", "1")
    with PskContext(0) as ctx:
        for n in (3, 4, 31, 33, 64, 65, 129, 200, 513, 1024, 1100, 2500):
            for ties in (False, True):
                if ties:
                    half = rng.choice([0.0, 0.001, 0.002, 0.0153, 1.0], (n, n))
                else:
                    half = np.round(rng.random((n, n)) * 0.1, 6)
                mat = np.tril(half, -1)
                mat = mat + mat.T
                t0 = time.time()
                a = ctx.nj_merges(mat)
                t_grid = time.time() - t0
                monkeypatch.setenv("PSK_NJ_ONE_WG", "1")
                t0 = time.time()
                b = ctx.nj_merges(mat)
                t_one = time.time() - t0
                monkeypatch.delenv("PSK_NJ_ONE_WG")
                for u, v in zip(a, b):
                    assert np.array_equal(np.asarray(u), np.asarray(v)), (n, ties)
                if n <= 1100:        # and both equal the oracle's plain loops (orc_nj: 0.6 s at 1,100 leaves)
                    for u, w_ in zip(a, OW.nj_merges(mat)):
                        assert np.array_equal(np.asarray(u), np.asarray(w_)), (n, ties, "oracle")
                if n >= 1024:
                    assert t_grid < t_one, (n, t_grid, t_one)


@pytest.mark.gpu
def test_gpu_neighbour_joining_in_lds_equals_the_one_workgroup_kernel(monkeypatch):
    """r04: nj_lds_kernel (a workgroup per C = 64 ... 8 columns of the matrix, the slices resident in LDS for the whole tree,
    joined-away rows as rows of -0.0, exchanges that carry the join number in every word; what psk_nj_merges runs from 288 to
    2,048 leaves) against nj_kernel (PSK_NJ_ONE_WG=1): pairs, both branch lengths and the last distance bit for bit -- at
    sizes around every slice width (64 / 32 / 16 / 8 columns: up to 256 / 512 / 1,024 / 2,048 leaves), with ties everywhere
    and without, forced below its threshold for the smallest trees -- and several times faster where it is the default."""
    import time
    from oracle import oracle_weights as OW
    from phenotypeseeker_amd.engine import PskContext
    rng = np.random.default_rng(99)
    monkeypatch.setenv("PSK_NJ_LDS_MIN", "3")
    monkeypatch.setenv("PSK_TRACE", "1")
    with PskContext(0) as ctx:
        for n in (3, 4, 5, 17, 63, 64, 65, 129, 256, 257, 300, 512, 513, 777, 1024, 1025, 1500, 2048):
            for ties in (False, True):
                if ties:
                    half = rng.choice([0.0, 0.001, 0.002, 0.0153, 1.0], (n, n))
                else:
                    half = np.round(rng.random((n, n)) * 0.1, 6)
                mat = np.tril(half, -1)
                mat = mat + mat.T
                a = ctx.nj_merges(mat)      # (the first call of a size also grows the context's buffers)
                t0 = time.time()
                a2 = ctx.nj_merges(mat)
                t_lds = time.time() - t0
                monkeypatch.setenv("PSK_NJ_ONE_WG", "1")
                t0 = time.time()
                b = ctx.nj_merges(mat)
                t_one = time.time() - t0
                monkeypatch.delenv("PSK_NJ_ONE_WG")
                for u, v, u2 in zip(a, b, a2):
                    assert np.array_equal(np.asarray(u), np.asarray(v)), (n, ties)
                    assert np.array_equal(np.asarray(u2), np.asarray(v)), (n, ties)
                if n <= 1100 or (n == 2048 and not ties):     # the oracle's plain loops: 5 s at 2,048 leaves
                    for u, w_ in zip(a, OW.nj_merges(mat)):
                        assert np.array_equal(np.asarray(u), np.asarray(w_)), (n, ties, "oracle")
                print("nj %d leaves%s: lds %.1f ms, one workgroup %.1f ms" % (n, " (ties)" if ties else "", 1e3 * t_lds, 1e3 * t_one))
                if n >= 1024:
                    assert 3 * t_lds < t_one, (n, t_lds, t_one)


@pytest.mark.gpu
def test_form_knobs_of_neighbour_joining_and_of_the_merge_build_are_validated(monkeypatch):
    """A knob that does not parse is an error that names it, not atoi's idea of it."""
    from phenotypeseeker_amd._lib import PskError
    from phenotypeseeker_amd.engine import PskContext
    from phenotypeseeker_amd.synth import GenomeSet
    mat = np.ones((5, 5)) - np.eye(5)
    with PskContext(0) as ctx:
        for bad in ("abc", "2", "12x", "99999"):
            monkeypatch.setenv("PSK_NJ_LDS_MIN", bad)
            with pytest.raises(PskError, match="PSK_NJ_LDS_MIN"):
                ctx.nj_merges(mat)
        monkeypatch.setenv("PSK_NJ_LDS_MIN", "3")
        assert len(ctx.nj_merges(mat)[0]) == 3
        monkeypatch.delenv("PSK_NJ_LDS_MIN")
        gs = GenomeSet(4, 5000, seed=1, gene_len=100)
        ctx.begin(16, 4)
        ctx.count_kmers_batch(0, [gs.sample(i)[1] for i in range(4)], 2)
        monkeypatch.setenv("PSK_NO_TILED_PRESENCE", "1")
        for bad in ("x", "1.5", "12 "):
            monkeypatch.setenv("PSK_MERGE_REC_DIV", bad)
            with pytest.raises(PskError, match="PSK_MERGE_REC_DIV"):
                ctx.build_presence()
        monkeypatch.setenv("PSK_MERGE_REC_DIV", "7")
        assert ctx.build_presence() > 0


def test_oracle_neighbour_joining_on_the_textbook_example():
    """orc_nj (the checker of the GPU neighbour joining) on the five-taxon example of the neighbour-joining literature
    (Saitou & Nei's method as tabulated in the Wikipedia article: a, b joined first with branches 2 and 3 -- although (d, e) ties
    with them in Q --, then c at 4 from the new node, the inner branches 3 and 2, d and e at 2 and 1): the joins, the branch
    lengths, and every leaf-to-leaf path of the written tree equal to the input distances (the matrix is additive)."""
    from oracle import oracle_weights as OW
    D = np.array([[0, 5, 9, 9, 8], [5, 0, 10, 10, 9], [9, 10, 0, 8, 7], [9, 10, 8, 0, 3], [8, 9, 7, 3, 0]], dtype=np.float64)
    mi, mj, d1, d2, last = OW.nj_merges(D)
    assert (int(mi[0]), int(mj[0])) == (0, 1) and (d1[0], d2[0]) == (2.0, 3.0)          # a, b first: the tie with (d, e) goes to the earlier pair
    names = list("abcde")
    t = OW.Tree(OW.nj_newick(names, D))
    leaves = {n.name: n for n in t.iter_leaves()}

    def up(n):
        out = []
        while n is not None:
            out.append(n)
            n = n.up
        return out
    for i, a in enumerate(names):
        for j, b in enumerate(names):
            if i < j:
                pa, pb = up(leaves[a]), up(leaves[b])
                common = next(x for x in pa if x in pb)
                assert sum(x.dist for x in pa[:pa.index(common)]) + sum(x.dist for x in pb[:pb.index(common)]) == D[i, j], (a, b)
    assert sorted(round(l.dist, 5) for l in leaves.values()) == [1.0, 2.0, 2.0, 3.0, 4.0]
