"""CPU tests (-m "not gpu"): the C-ABI library loads and exports every symbol include/psk.h
declares (no compute calls without a GPU), the host-side tokeniser framing reproduces
glistmaker's lists, and the product fails loudly -- never falls back -- when no GPU exists."""
import os
import re

import numpy as np
import pytest

from helpers import ROOT, tokenizer_cases


def _declared():
    with open(os.path.join(ROOT, "include", "psk.h")) as f:
        txt = f.read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(psk_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from phenotypeseeker_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(_lib.exported_names())
    assert lib.psk_version() >> 16 == 1


def test_no_cpu_fallback_without_gpu():
    import torch  # only to learn whether this box has a GPU
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from phenotypeseeker_amd.engine import PskContext
    from phenotypeseeker_amd._lib import PskError
    with pytest.raises(PskError):
        PskContext(0)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "phenotypeseeker_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                with open(os.path.join(dirpath, fn)) as f:
                    src = f.read()
                assert "oracle" not in src.replace("oracle/gen_golden.py", ""), os.path.join(dirpath, fn)


def test_product_never_imports_torch():
    """north_star: Python host code over a ctypes C ABI, no PyTorch -- the collectives are RCCL calls made by
    libpsk.so (csrc/comm.hip); torch appears only in tests/ (the gloo transport) and as the driver's launcher."""
    import subprocess
    import sys
    pkg = os.path.join(ROOT, "phenotypeseeker_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(".py"):
                with open(os.path.join(dirpath, fn)) as f:
                    src = f.read()
                assert not re.search(r"^\s*(import|from)\s+torch\b", src, flags=re.M), os.path.join(dirpath, fn)
    with open(os.path.join(ROOT, "bench.py")) as f:
        assert not re.search(r"^\s*(import|from)\s+torch\b", f.read(), flags=re.M)
    code = ("import sys; sys.path.insert(0, %r); import phenotypeseeker_amd.dist, phenotypeseeker_amd.modeling, "
            "phenotypeseeker_amd.prediction, phenotypeseeker_amd.cli; assert 'torch' not in sys.modules" % ROOT)
    assert subprocess.run([sys.executable, "-c", code], timeout=120).returncode == 0


def _roll_clean(clean, k):
    """what the extract kernel computes from the clean stream, in plain Python"""
    mask = (1 << (2 * k)) - 1
    fw = rc = run = 0
    out = []
    for c in clean:
        if c == 10:
            run = 0
            continue
        code = ((c >> 1) ^ (c >> 2)) & 3
        fw = ((fw << 2) | code) & mask
        rc = (rc >> 2) | ((3 - code) << (2 * (k - 1)))
        run = min(run + 1, k)
        if run == k:
            out.append(min(fw, rc))
    return out


def test_host_framing_reproduces_glistmaker(oracle):
    from phenotypeseeker_amd.engine import frame_sequence
    n = 0
    for data, k, ref in tokenizer_cases():
        clean = frame_sequence(data)
        assert set(clean) <= set(b"ACGTUacgtu\n")
        words = sorted(_roll_clean(clean, k))
        uw, cnt = np.unique(np.array(words, dtype=np.uint64), return_counts=True)
        if ref is None:
            assert len(uw) == 0
        else:
            assert oracle.list_bytes(k, uw, cnt.astype(np.uint32)) == ref, data[:60]
            n += 1
    assert n > 100


def test_words_per_row_is_one_or_even():
    """8-byte rows up to 64 samples (the reference's example set: ~30 genomes), 16-byte aligned rows beyond"""
    from phenotypeseeker_amd.engine import words_per_row
    assert [words_per_row(n) for n in (1, 30, 64, 65, 128, 129, 256, 2048)] == [1, 1, 1, 2, 2, 4, 4, 32]


def test_cv_splitters_match_sklearn_fixtures():
    from phenotypeseeker_amd import cv
    from helpers import GOLDEN
    z = np.load(os.path.join(GOLDEN, "model_kat.npz"))
    for tag in ("1", "2"):
        y = z["y" + tag]
        k = int(min(np.bincount(y).min(), 10))
        assert np.array_equal(cv.stratified_kfold(y, k), z["skf_folds" + tag])
    assert np.array_equal(cv.kfold(len(z["yc2"]), 10), z["kf_folds2"])
    rng = np.random.default_rng(0)
    from oracle import oracle_model as OM
    for _ in range(50):
        n = int(rng.integers(8, 90))
        y = (rng.random(n) < rng.uniform(0.2, 0.8)).astype(int)
        if min(np.bincount(y, minlength=2)) < 2:
            continue
        k = int(min(np.bincount(y).min(), rng.integers(2, 11)))
        assert np.array_equal(cv.stratified_kfold(y, k), OM.stratified_kfold(y, k))
        assert np.array_equal(cv.kfold(n, k), OM.kfold(n, k))


def test_train_test_split_matches_sklearn_fixture():
    import json
    from phenotypeseeker_amd import cv
    from helpers import GOLDEN
    with open(os.path.join(GOLDEN, "split_kat.json")) as f:
        cases = json.load(f)["cases"]
    assert len(cases) >= 60
    for c in cases:
        tr, te = cv.train_test_split_indices(c["n"], c["test_size"], None if c["y"] is None else np.array(c["y"]), 55)
        assert list(tr) == c["train"] and list(te) == c["test"]


def _python_result_table(head, kind, kmers, stat, p, mx, my, n_with, presence, valid, names):
    """The writer of r01-r03 (pure Python, the DataFrame.to_csv text of modeling.py:1112-1145): the reference the C writer
    is held to byte for byte."""
    import numpy as np
    pstr = ["%.2E" % v for v in p]
    order = sorted(range(len(pstr)), key=lambda i: (pstr[i], kmers[i]))
    lines = []
    for i in order:
        who = [names[j] for j in range(len(names)) if presence[i][j] and valid[j]]
        tail = " ".join(["|"] + who)
        f = [kmers[i], repr(float(np.round(np.float64(stat[i]), 2))), pstr[i]]
        if kind == 1:
            f += [repr(float(np.round(np.float64(mx[i]), 2))), repr(float(np.round(np.float64(my[i]), 2)))]
        lines.append("\t".join(f + [str(int(n_with[i])), tail]))
    return order, ("\n".join([head] + lines) + "\n").encode()


def test_result_tables_written_by_libpsk_equal_the_python_writer(tmp_path):
    """psk_write_result_tables (a9, host code in libpsk): the files and the line order equal the Python writer's on random
    rows with the awkward doubles in them -- whole numbers ('3.0'), values that switch repr() to exponent notation
    (1e-05, 1e+16, 1.5e-07), negative zero, huge and tiny statistics, p-values down to subnormals, inf and nan --, with NA
    samples, sample names of several bytes per character, both table kinds, with and without the top file, zero rows."""
    import ctypes
    import numpy as np
    from phenotypeseeker_amd import _lib, formats
    lib = _lib.load()
    rng = np.random.default_rng(12)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    for case in range(12):
        n = int(rng.integers(1, 200))
        m = 0 if case == 0 else int(rng.integers(1, 6000 if case == 11 else 300))
        k = int(rng.choice([5, 13, 16, 32]))
        kind = case & 1
        names = [("s%d" % i) if i % 7 else ("Tõnu_%d·x" % i) for i in range(n)]
        valid = (rng.random(n) < 0.9).astype(np.uint8)
        words = rng.integers(0, 1 << min(2 * k, 62), m, dtype=np.uint64)
        special = np.array([0.0, -0.0, 3.0, 100.0, 1e-5, 1.5e-7, 1e16, 123456789012345678.0, 0.005, 0.015, 2.675, -7.125, 1e15,
                            1234.5678, 9.999e-5, 1e22, float("inf"), float("-inf"), float("nan"), 0.00049, 5e-324])
        def vals():
            v = rng.normal(0, 1, m) * 10.0 ** rng.integers(-8, 20, m)
            pick = rng.random(m) < 0.3
            v[pick] = rng.choice(special, pick.sum())
            return v
        stat, mx, my = vals(), vals(), vals()
        p = 10.0 ** -rng.uniform(0, 320, m)
        pp = rng.random(m) < 0.3
        p[pp] = rng.choice([1.23e-5, 1.234e-5, 1.0, 9.995e-3, 5e-324, 0.0], pp.sum())
        pres = (rng.random((m, n)) < rng.uniform(0.05, 0.9)).astype(np.uint8)
        n_with = pres.sum(axis=1).astype(np.int32)
        wpr = (n + 63) // 64
        pad = np.zeros((m, wpr * 64), np.uint8)
        pad[:, :n] = pres
        bits = np.ascontiguousarray(np.packbits(pad, axis=1, bitorder="little").view("<u8")).reshape(m, wpr)
        enc = [s.encode() for s in names]
        off = np.zeros(n + 1, np.int64)
        off[1:] = np.cumsum([len(e) for e in enc])
        head = "k-mer\tstat\tp-value\tn\tsamples" if kind == 0 else "k-mer\tt\tp\t+\t-\tn\tsamples"
        n_top = int(rng.integers(0, 50))
        order = np.zeros(m, np.int64)
        path, top = str(tmp_path / ("t%d.tsv" % case)).encode(), str(tmp_path / ("t%d_top.tsv" % case)).encode()
        rc = lib.psk_write_result_tables(None, path, top if case % 3 else None, n_top, head.encode(), kind, m, vp(words), k, vp(stat), vp(p),
                                         vp(mx), vp(my), vp(n_with), vp(bits), wpr, n, vp(valid), b"".join(enc), vp(off), vp(order))
        assert rc == 0
        kmers = formats.words_to_kmers(words, k)
        want_order, want = _python_result_table(head, kind, kmers, stat, p, mx, my, n_with, pres, valid, names)
        assert order.tolist() == want_order, case
        with open(path, "rb") as f:
            assert f.read() == want, case
        if case % 3:
            with open(top, "rb") as f:
                assert f.read() == b"\n".join(want.split(b"\n")[: 1 + min(n_top, m)]) + b"\n", case
    assert lib.psk_write_result_tables(None, None, None, 0, b"h", 0, 0, None, 13, None, None, None, None, None, None, 1, 1, None, None, None,
                                       None) == -1


def test_model_coefficient_writer_equals_the_python_lines(tmp_path):
    """r04: psk_write_model_coefficients (host code in libpsk) appends the lines of the coefficient file exactly as the Python
    writer formatted them: "%s\\t%s\\t%d\\t| %s\\n" % (k-mer, repr(coef), count, " ".join(names)) -- awkward doubles, k-mers no
    sample carries ("| " with its trailing space), names of every length."""
    import ctypes
    from phenotypeseeker_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(11)
    for n, p in ((1, 1), (7, 13), (130, 40), (0, 3), (5, 0)):
        names = ["s%d%s" % (i, "x" * int(rng.integers(0, 9))) for i in range(n)]
        kmers = ["".join(rng.choice(list("ACGT"), 13)) for _ in range(p)]
        X = (rng.random((n, p)) < 0.3).astype(np.int64) * rng.integers(1, 5, (n, p))
        if p > 1:
            X[:, 1] = 0
        coefs = np.concatenate([[0.0, -0.0, 1e-5, 123456789012345680.0, 1e16, 0.1 + 0.2, -3.5e-310][:p], rng.normal(0, 1, max(p - 7, 0))])[:p]
        want = "".join("%s\t%s\t%d\t| %s\n" % (kmers[j], repr(float(coefs[j])), int((X[:, j] != 0).sum()),
                                              " ".join(names[i] for i in range(n) if X[i, j] != 0)) for j in range(p))
        path = tmp_path / ("coef_%d_%d.txt" % (n, p))
        path.write_text("header\n")
        kenc, nenc = [k.encode() for k in kmers], [m.encode() for m in names]
        koff = np.zeros(p + 1, dtype=np.int64); koff[1:] = np.cumsum([len(e) for e in kenc])
        noff = np.zeros(n + 1, dtype=np.int64); noff[1:] = np.cumsum([len(e) for e in nenc])
        Xa, cf = np.ascontiguousarray(X), np.ascontiguousarray(coefs, dtype=np.float64)
        vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        rc = lib.psk_write_model_coefficients(None, str(path).encode(), p, b"".join(kenc), vp(koff), vp(cf), vp(Xa), n, b"".join(nenc), vp(noff))
        assert rc == 0
        assert path.read_text() == "header\n" + want
