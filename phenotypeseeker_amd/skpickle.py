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
