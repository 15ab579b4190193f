"""Cross-validation splitters GridSearchCV applies for an integer `cv` (modeling.py:1075-1085):
StratifiedKFold for the classifier, KFold for the regressor, both without shuffling
(scikit-learn 0.22 semantics, the version the reference pins)."""
import numpy as np


def stratified_kfold(y, n_splits):
    """Test-fold id per sample.  Each class's members, taken in sample order, are dealt to the
    folds so that fold f receives as many of them as positions f, f+k, f+2k, ... of the sorted
    label vector hold that class (scikit-learn's allocation since 0.22); classes are numbered
    by first appearance."""
    y = np.asarray(y)
    n = len(y)
    first_seen = {}
    for v in y.tolist():
        first_seen.setdefault(v, len(first_seen))
    enc = np.array([first_seen[v] for v in y.tolist()], dtype=np.int64)
    n_classes = len(first_seen)
    ordered = np.sort(enc)
    folds = np.zeros(n, dtype=np.int32)
    for c in range(n_classes):
        per_fold = [int(np.count_nonzero(ordered[f::n_splits] == c)) for f in range(n_splits)]
        ids = np.repeat(np.arange(n_splits, dtype=np.int32), per_fold)
        folds[enc == c] = ids
    return folds


def kfold(n, n_splits):
    """Contiguous test blocks; the first n % k folds get one extra sample."""
    base, extra = divmod(n, n_splits)
    sizes = [base + (1 if f < extra else 0) for f in range(n_splits)]
    return np.repeat(np.arange(n_splits, dtype=np.int32), sizes)
