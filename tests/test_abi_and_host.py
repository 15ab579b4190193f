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


def test_words_per_row_is_even():
    from phenotypeseeker_amd.engine import words_per_row
    assert [words_per_row(n) for n in (1, 64, 65, 128, 129, 256, 2048)] == [2, 2, 2, 2, 4, 4, 32]


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
