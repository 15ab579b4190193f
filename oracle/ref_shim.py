"""Launcher that imports the UNMODIFIED reference Python in the build container.

TEST INFRASTRUCTURE (fixture generation only; never runs on the GPU box, where
/root/reference does not exist).  The reference pins 2020-era library versions with
pkg_resources.require (modeling.py:49-52) and imports Bio / ete3 / statsmodels, none of which
are installed here; this shim stubs those modules, turns the version pin into a no-op and
wraps the sklearn metric functions so that they return np.float64 (the reference calls
.round() on them, modeling.py:1316-1356).  No reference code is altered or copied.

Usage (inside a scratch directory, with /root/reference/bin on PATH):
    python ref_shim.py modeling data.pheno -nt 4 --omit_B_correction
or  import ref_shim; M = ref_shim.load_modeling()
"""
import runpy
import sys
import types

REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def load_modeling(ttest_ind=None):
    _stub("Bio")
    _stub("Bio.Phylo")
    _stub("Bio.Phylo.TreeConstruction", DistanceTreeConstructor=object, _DistanceMatrix=object)
    _stub("ete3", Tree=object)
    _stub("statsmodels")
    _stub("statsmodels.stats")
    _stub("statsmodels.stats.weightstats", ttest_ind=ttest_ind)
    import pkg_resources
    pkg_resources.require = lambda *a, **k: None
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import numpy as np
    import PhenotypeSeeker.modeling as M

    def f64(f):
        return lambda *a, **k: np.float64(f(*a, **k))

    for n in ("f1_score", "recall_score", "roc_auc_score", "average_precision_score", "cohen_kappa_score",
              "mean_squared_error", "matthews_corrcoef", "accuracy_score"):
        if hasattr(M, n):
            setattr(M, n, f64(getattr(M, n)))
    from sklearn.model_selection._search import BaseSearchCV
    if not getattr(BaseSearchCV.score, "_psk_wrapped", False):
        wrapped = f64(BaseSearchCV.score)
        wrapped._psk_wrapped = True
        BaseSearchCV.score = wrapped
    return M


if __name__ == "__main__":
    load_modeling()
    sys.argv[0] = "phenotypeseeker"
    runpy.run_path(REF + "/scripts/phenotypeseeker", run_name="__main__")
