"""Performance measures written into summary_of_*_analysis_*.txt (modeling.py:1255-1412), numpy
only.  Definitions follow scikit-learn's for binary labels {0, 1}."""
import numpy as np

from . import _stats


def confusion(labels, predictions):
    y = np.asarray(labels).astype(int)
    p = np.asarray(predictions).astype(int)
    cm = np.zeros((2, 2), dtype=np.int64)
    for a in (0, 1):
        for b in (0, 1):
            cm[a, b] = int(np.sum((y == a) & (p == b)))
    return cm


def _div(a, b):
    return float(a) / float(b) if b else 0.0


def recall(labels, predictions, positive=1):
    y, p = np.asarray(labels), np.asarray(predictions)
    return np.float64(_div(np.sum((y == positive) & (p == positive)), np.sum(y == positive)))


def precision(labels, predictions, positive=1):
    y, p = np.asarray(labels), np.asarray(predictions)
    return np.float64(_div(np.sum((y == positive) & (p == positive)), np.sum(p == positive)))


def f1(labels, predictions, positive=1):
    pr, rc = precision(labels, predictions, positive), recall(labels, predictions, positive)
    return np.float64(_div(2 * pr * rc, pr + rc))


def accuracy(labels, predictions):
    return np.float64(np.mean(np.asarray(labels) == np.asarray(predictions)))


def roc_auc(labels, scores):
    """area under the ROC curve (ties share rank); on hard predictions = (TPR + TNR) / 2"""
    y = np.asarray(labels).astype(int)
    s = np.asarray(scores, dtype=np.float64)
    npos, nneg = int(y.sum()), int((1 - y).sum())
    if npos == 0 or nneg == 0:
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    r = _stats.rankdata(s)
    return np.float64((r[y == 1].sum() - npos * (npos + 1) / 2.0) / (npos * nneg))


def average_precision(labels, scores):
    y = np.asarray(labels).astype(int)
    s = np.asarray(scores, dtype=np.float64)
    order = np.argsort(-s, kind="mergesort")
    y, s = y[order], s[order]
    tp = np.cumsum(y)
    distinct = np.r_[np.nonzero(np.diff(s))[0], len(s) - 1]
    tps = tp[distinct]
    prec = tps / (distinct + 1.0)
    rec = tps / max(int(y.sum()), 1)
    prev = np.r_[0.0, rec[:-1]]
    return np.float64(np.sum((rec - prev) * prec))


def matthews(labels, predictions):
    cm = confusion(labels, predictions)
    tn, fp, fn, tp = cm[0, 0], cm[0, 1], cm[1, 0], cm[1, 1]
    den = np.sqrt(float((tp + fp) * (tp + fn) * (tn + fp) * (tn + fn)))
    return np.float64(_div(tp * tn - fp * fn, den))


def cohen_kappa(labels, predictions):
    cm = confusion(labels, predictions).astype(np.float64)
    n = cm.sum()
    po = np.trace(cm) / n
    pe = float((cm.sum(axis=0) * cm.sum(axis=1)).sum()) / (n * n)
    return np.float64(_div(po - pe, 1.0 - pe)) if pe != 1.0 else np.float64(np.nan)


def very_major_error(labels, predictions):
    """resistant (1) predicted sensitive (0) -- modeling.py:1467-1475"""
    y, p = np.asarray(labels), np.asarray(predictions)
    return round(float(np.sum((y == 1) & (p == 0))) / len(y), 2)


def major_error(labels, predictions):
    """sensitive (0) predicted resistant (1) -- modeling.py:1477-1485"""
    y, p = np.asarray(labels), np.asarray(predictions)
    return round(float(np.sum((y == 0) & (p == 1))) / len(y), 2)


def within_1_tier_accuracy(labels, predictions):
    """modeling.py:1487-1496"""
    y, p = np.asarray(labels, dtype=np.float64), np.asarray(predictions, dtype=np.float64)
    return round(float(np.sum(np.abs(y - p) <= 1)) / len(y), 2)


def mean_squared_error(labels, predictions):
    y, p = np.asarray(labels, dtype=np.float64), np.asarray(predictions, dtype=np.float64)
    return np.float64(np.mean((y - p) ** 2))


def classification_report(labels, predictions, target_names=("sensitive", "resistant"), digits=2):
    """Text table in scikit-learn's layout (modeling.py:1371-1374)."""
    y = np.asarray(labels).astype(int)
    rows = []
    for cls, name in enumerate(target_names):
        rows.append((name, precision(labels, predictions, cls), recall(labels, predictions, cls),
                     f1(labels, predictions, cls), int(np.sum(y == cls))))
    width = max(len(n) for n in list(target_names) + ["weighted avg"])
    head = "{:>{w}s} ".format("", w=width) + "".join(" {:>9}".format(h) for h in ("precision", "recall", "f1-score", "support"))
    out = head + "\n\n"
    for name, p_, r_, f_, s_ in rows:
        out += "{:>{w}s} ".format(name, w=width) + "".join(" {:>9.{d}f}".format(v, d=digits) for v in (p_, r_, f_)) + \
            " {:>9}\n".format(s_)
    out += "\n"
    total = int(sum(r[4] for r in rows))
    out += "{:>{w}s} ".format("accuracy", w=width) + " {:>9}".format("") * 2 + \
        " {:>9.{d}f}".format(accuracy(labels, predictions), d=digits) + " {:>9}\n".format(total)
    macro = [float(np.mean([r[i] for r in rows])) for i in (1, 2, 3)]
    wts = np.array([r[4] for r in rows], dtype=np.float64)
    weighted = [float(np.sum(np.array([r[i] for r in rows]) * wts) / max(wts.sum(), 1.0)) for i in (1, 2, 3)]
    for name, vals in (("macro avg", macro), ("weighted avg", weighted)):
        out += "{:>{w}s} ".format(name, w=width) + "".join(" {:>9.{d}f}".format(v, d=digits) for v in vals) + \
            " {:>9}\n".format(total)
    return out
