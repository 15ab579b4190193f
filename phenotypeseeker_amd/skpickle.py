"""Model files as the reference writes them -- joblib-loadable pickles of scikit-learn objects (modeling.py:975-988;
loaded by a plain joblib.load, prediction.py:124-129) -- written WITHOUT importing scikit-learn.

Importing scikit-learn costs 0.3-0.5 s, as much as all the GPU work of a 256-genome `modeling` run.  What a fitted
GridSearchCV / LogisticRegression / Lasso / Ridge pickles as is its class (module + name) and its attribute
dictionary; the dictionaries of default-constructed estimators are recorded per scikit-learn version in
sklearn_shells.json (tools/make_sklearn_shells.py), the fitted attributes come from this package's own solver.  The
writer emits the pickle opcodes for "object of class <module>.<name> with this state" directly (GLOBAL / NEWOBJ /
BUILD, protocol 2) and lets the standard pickler serialise the leaves (numbers, strings, numpy arrays).  When the
installed scikit-learn has no template the caller falls back to GridSearch.to_sklearn(), which imports it.
"""
import json
import math
import os
import pickle


class Shell:
    """Stands for an instance of module.name whose __dict__ will be `state`."""

    def __init__(self, module, name, state):
        self.module, self.name, self.state = module, name, state


_templates = None


def installed_sklearn_version():
    try:
        from importlib import metadata
        return metadata.version("scikit-learn")
    except Exception:
        return None


def template(cls_name):
    """(module, name, default state) of `cls_name` for the installed scikit-learn, or None."""
    global _templates
    if _templates is None:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "sklearn_shells.json")) as f:
            _templates = json.load(f)
    t = _templates.get(installed_sklearn_version() or "", {}).get(cls_name)
    if t is None:
        return None
    state = {k: (math.nan if v == "nan" else v) for k, v in t["state"].items()}
    return t["module"], t["name"], state


def make(cls_name, **attrs):
    t = template(cls_name)
    if t is None:
        return None
    module, name, state = t
    state.update(attrs)
    return Shell(module, name, state)


def _emit(obj, out):
    if isinstance(obj, Shell):
        out.append(b"c" + obj.module.encode() + b"\n" + obj.name.encode() + b"\n")   # GLOBAL
        out.append(b")\x81")                                                          # EMPTY_TUPLE, NEWOBJ
        _emit(obj.state, out)
        out.append(b"b")                                                              # BUILD
    elif isinstance(obj, dict) and _has_shell(obj):
        out.append(b"}(")                                                             # EMPTY_DICT, MARK
        for k, v in obj.items():
            _emit(k, out)
            _emit(v, out)
        out.append(b"u")                                                              # SETITEMS
    else:
        # a leaf: the standard pickler's stream without its PROTO header and STOP; its memo indices start from 0 again,
        # which is harmless -- every GET of a fragment refers to a PUT of the same fragment, executed just before
        frag = pickle.dumps(obj, protocol=2)
        assert frag[:2] == b"\x80\x02" and frag[-1:] == b"."
        out.append(frag[2:-1])


def _has_shell(obj):
    if isinstance(obj, Shell):
        return True
    if isinstance(obj, dict):
        return any(_has_shell(v) for v in obj.values())
    return False


def dumps(obj):
    out = [b"\x80\x02"]
    _emit(obj, out)
    out.append(b".")
    return b"".join(out)


# ---- reading a model file for `prediction` without importing scikit-learn ----------------------------------------
class _Stub:
    """Stands in for any scikit-learn class while a model file is unpickled: keeps the attribute dictionary."""

    def __init__(self, *args, **kwargs):
        pass

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        elif isinstance(state, tuple) and len(state) == 2:       # (dict, slots)
            for part in state:
                if isinstance(part, dict):
                    self.__dict__.update(part)


class _StubUnpickler(pickle.Unpickler):
    _made = {}

    def find_class(self, module, name):
        if module.split(".")[0] == "joblib":
            # a joblib-wrapped file carries raw array bytes inline behind a NumpyArrayWrapper: read as pickle opcodes they
            # are garbage -- stop here, the caller takes joblib.load
            raise pickle.UnpicklingError("joblib-wrapped arrays: not a plain pickle")
        if module.split(".")[0] == "sklearn":
            key = (module, name)
            if key not in self._made:
                self._made[key] = type(name, (_Stub,), {"__module__": module})
            return self._made[key]
        return super().find_class(module, name)


class LinearModel:
    """What `prediction` needs of a fitted (GridSearchCV of a) linear estimator: predict / predict_proba with the
    expressions scikit-learn evaluates (linear_model/_base.py: X @ coef_.T + intercept_; classes_[scores > 0];
    expit(scores) for the liblinear / one-vs-rest probability of a binary problem)."""

    def __init__(self, coef, intercept, classes):
        import numpy as np
        self.coef_, self.intercept_ = np.asarray(coef, dtype=np.float64), np.asarray(intercept, dtype=np.float64)
        self.classes_ = None if classes is None else np.asarray(classes)

    def _scores(self, X):
        import numpy as np
        X = np.asarray(X, dtype=np.float64)
        if self.classes_ is None:
            return X @ self.coef_.T + self.intercept_ if self.coef_.ndim == 2 else X @ self.coef_ + self.intercept_
        s = X @ self.coef_.T + self.intercept_
        return s.ravel() if s.ndim == 2 and s.shape[1] == 1 else s

    def predict(self, X):
        import numpy as np
        s = self._scores(X)
        if self.classes_ is None:
            return s
        return self.classes_[(s > 0).astype(np.int64)]

    def predict_proba(self, X):
        import numpy as np
        s = self._scores(X)
        # scipy.special.expit, without the overflow warning of exp(-s) at strongly negative scores
        e = np.exp(-np.abs(s))
        p = np.where(s >= 0, 1.0 / (1.0 + e), e / (1.0 + e))
        return np.vstack([1.0 - p, p]).T


def load_linear_package(path):
    """The model package of `path` ({'model', 'kmers', 'pca', 'pred_scale'}) with 'model' as a LinearModel -- for files
    that are plain pickles of a (grid search over a) binary linear classifier or a linear regressor, which is what
    `modeling` writes.  None for anything else (joblib-wrapped arrays, multi-class models, other estimators, a PCA
    pipeline): the caller then takes joblib.load and scikit-learn itself."""
    try:
        with open(path, "rb") as f:
            pkg = _StubUnpickler(f).load()
        if not isinstance(pkg, dict) or pkg.get("pca") or "model" not in pkg:
            return None
        m = pkg["model"]
        est = getattr(m, "best_estimator_", m)
        kind = type(est).__name__
        coef, icpt = getattr(est, "coef_", None), getattr(est, "intercept_", None)
        if coef is None or icpt is None:
            return None
        if kind in ("LogisticRegression",):
            classes = getattr(est, "classes_", None)
            if classes is None or len(classes) != 2 or getattr(coef, "shape", (0,))[0] != 1:
                return None
            # a binary model saved with multi_class='multinomial' has softmax([-d, d]) = expit(2 d) probabilities in
            # scikit-learn, not expit(d): leave it to joblib.load + scikit-learn
            if getattr(est, "multi_class", "auto") not in ("auto", "ovr", "deprecated", "warn"):
                return None
            model = LinearModel(coef, icpt, classes)
        elif kind in ("Lasso", "Ridge"):
            model = LinearModel(coef, icpt, None)
        else:
            return None
        out = dict(pkg)
        out["model"] = model
        return out
    except Exception:   # noqa: BLE001 -- any surprise: the ordinary loader decides
        return None
