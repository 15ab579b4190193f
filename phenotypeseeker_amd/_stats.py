"""Small numeric helpers for the reporting stage (host side, pure Python/numpy): Student-t tail,
rank transform, Pearson / Spearman with p-values (what the reference takes from scipy.stats at
modeling.py:1268-1280)."""
import math

import numpy as np


def _betacf(a, b, x):
    tiny, eps = 1e-300, 1e-16
    qab, qap, qam = a + b, a + 1.0, a - 1.0
    c, d = 1.0, 1.0 - qab * x / qap
    d = 1.0 / (d if abs(d) > tiny else tiny)
    h = d
    for m in range(1, 10001):
        m2 = 2 * m
        aa = m * (b - m) * x / ((qam + m2) * (a + m2))
        d = 1.0 + aa * d
        d = 1.0 / (d if abs(d) > tiny else tiny)
        c = 1.0 + aa / c
        c = c if abs(c) > tiny else tiny
        h *= d * c
        aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2))
        d = 1.0 + aa * d
        d = 1.0 / (d if abs(d) > tiny else tiny)
        c = 1.0 + aa / c
        c = c if abs(c) > tiny else tiny
        de = d * c
        h *= de
        if abs(de - 1.0) < eps:
            break
    return h


def betainc(a, b, x):
    if x <= 0.0:
        return 0.0
    if x >= 1.0:
        return 1.0
    bt = math.exp(math.lgamma(a + b) - math.lgamma(a) - math.lgamma(b) + a * math.log(x) + b * math.log1p(-x))
    if x < (a + 1.0) / (a + b + 2.0):
        return bt * _betacf(a, b, x) / a
    return 1.0 - bt * _betacf(b, a, 1.0 - x) / b


def t_two_sided_p(t, df):
    if math.isnan(t) or not df > 0:
        return float("nan")
    if math.isinf(t):
        return 0.0
    return betainc(0.5 * df, 0.5, df / (df + t * t))


def rankdata(a):
    """average ranks (ties share the mean rank), 1-based"""
    a = np.asarray(a, dtype=np.float64)
    order = np.argsort(a, kind="mergesort")
    ranks = np.empty(len(a))
    i = 0
    while i < len(a):
        j = i
        while j + 1 < len(a) and a[order[j + 1]] == a[order[i]]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    return ranks


def pearsonr(x, y):
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    n = len(x)
    xm, ym = x - x.mean(), y - y.mean()
    den = math.sqrt(float((xm * xm).sum()) * float((ym * ym).sum()))
    if den == 0.0 or n < 3:
        return float("nan"), float("nan")
    r = max(-1.0, min(1.0, float((xm * ym).sum()) / den))
    if abs(r) == 1.0:
        return r, 0.0
    t = r * math.sqrt((n - 2) / ((1.0 - r) * (1.0 + r)))
    return r, t_two_sided_p(t, n - 2)


def spearmanr(x, y):
    return pearsonr(rankdata(x), rankdata(y))
