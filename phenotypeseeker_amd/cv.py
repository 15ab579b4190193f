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


def _approximate_mode(class_counts, n_draws, rng):
    """How many of n_draws go to each class: floor of the proportional share, the remainder handed
    out by largest fractional part with random tie-breaking (scikit-learn's rule)."""
    class_counts = np.asarray(class_counts, dtype=np.float64)
    continuous = class_counts / class_counts.sum() * n_draws
    floored = np.floor(continuous)
    need = int(n_draws - floored.sum())
    if need > 0:
        remainder = continuous - floored
        for value in np.sort(np.unique(remainder))[::-1]:
            (inds,) = np.where(remainder == value)
            add_now = min(len(inds), need)
            inds = rng.choice(inds, size=add_now, replace=False)
            floored[inds] += 1
            need -= add_now
            if need == 0:
                break
    return floored.astype(int)


def train_test_split_indices(n, test_size, stratify=None, random_state=55):
    """(train, test) index arrays of sklearn.model_selection.train_test_split(..., test_size=float,
    random_state=55, stratify=y | None) -- the hold-out split of modeling.py:924-934: n_test =
    ceil(test_size * n); ShuffleSplit / StratifiedShuffleSplit drawn from numpy's legacy
    RandomState(random_state)."""
    n_test = int(np.ceil(test_size * n))
    n_train = n - n_test
    if n_train <= 0 or n_test <= 0:
        raise ValueError("With n_samples=%d and test_size=%r the resulting train or test set would be empty." % (n, test_size))
    rng = np.random.RandomState(random_state)
    if stratify is None:
        perm = rng.permutation(n)
        return perm[n_test:n_test + n_train], perm[:n_test]
    y = np.asarray(stratify)
    classes, y_idx = np.unique(y, return_inverse=True)
    counts = np.bincount(y_idx)
    if counts.min() < 2:
        raise ValueError("The least populated class in y has only 1 member, which is too few.")
    if n_train < len(classes) or n_test < len(classes):
        raise ValueError("The train / test size should be greater or equal to the number of classes")
    class_indices = np.split(np.argsort(y_idx, kind="mergesort"), np.cumsum(counts)[:-1])
    n_i = _approximate_mode(counts, n_train, rng)
    t_i = _approximate_mode(counts - n_i, n_test, rng)
    train, test = [], []
    for c in range(len(classes)):
        perm = rng.permutation(counts[c])
        idx = class_indices[c].take(perm, mode="clip")
        train.extend(idx[: n_i[c]])
        test.extend(idx[n_i[c]: n_i[c] + t_i[c]])
    return rng.permutation(train), rng.permutation(test)
