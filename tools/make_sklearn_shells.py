#!/usr/bin/env python3
"""Records, for the installed scikit-learn, what a default-constructed LogisticRegression / Lasso / Ridge /
GridSearchCV pickles as (module, class name, state) into phenotypeseeker_amd/sklearn_shells.json, keyed by the
scikit-learn version.  phenotypeseeker_amd/skpickle.py writes model files from these templates without importing
scikit-learn (0.3-0.5 s, as long as the rest of a 256-genome `modeling` run); a version without a template takes the
import.  usage: tools/make_sklearn_shells.py"""
import json
import math
import os

import sklearn
from sklearn.linear_model import Lasso, LogisticRegression, Ridge
from sklearn.model_selection import GridSearchCV

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(ROOT, "phenotypeseeker_amd", "sklearn_shells.json")


def plain(v):
    if v is None or isinstance(v, (bool, int, str)):
        return v
    if isinstance(v, float):
        return "nan" if math.isnan(v) else v
    raise TypeError("template value %r is not plain" % (v,))


def shell(obj, drop=()):
    st = obj.__getstate__()
    return {"module": type(obj).__module__, "name": type(obj).__qualname__,
            "state": {k: plain(v) for k, v in st.items() if k not in drop}}


out = {}
if os.path.exists(path):
    with open(path) as f:
        out = json.load(f)
out[sklearn.__version__] = {
    "LogisticRegression": shell(LogisticRegression()),
    "Lasso": shell(Lasso()),
    "Ridge": shell(Ridge()),
    "GridSearchCV": shell(GridSearchCV(LogisticRegression(), {"C": [1.0]}), drop=("estimator", "param_grid")),
}
with open(path, "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
print("templates for scikit-learn", sorted(out))
