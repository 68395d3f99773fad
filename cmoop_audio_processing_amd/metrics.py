"""Pareto-front quality metrics, host only, numpy.

Restated from the reference's analysis notebooks (objectives in minimisation space
(-Accuracy, Size_MB, FPR)):
  true_front ............ union of fronts minus dominated points   compare.ipynb cell 0, step 6
  generational_distance / inverted_gd                                step 7
  spread_metric                                                      step 8
  coverage_metric (C-metric)                                         step 9
  tchebycheff_rank ...... "Tchebycheff s_rank.ipynb": max_j w_j |f_j - z*_j| on (1-acc, size, fpr),
                          equal weights, rank(method='min')
Hypervolume lives in nsga.hypervolume (pygmo is absent; exact sweep).
Pinned by tests/golden/metrics_golden.json (the notebook's functions executed here).
"""
from __future__ import annotations

from typing import Sequence

import numpy as np


def _cdist(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.sqrt(((a[:, None, :] - b[None, :, :]) ** 2).sum(-1))


def dominates_min(a, b) -> bool:
    a, b = np.asarray(a), np.asarray(b)
    return bool(np.all(a <= b) and np.any(a < b))


def true_front(points) -> np.ndarray:
    pts = np.asarray(points, float)
    keep = [i for i in range(len(pts)) if not any(i != j and dominates_min(pts[j], pts[i]) for j in range(len(pts)))]
    return pts[keep]


def generational_distance(obtained, reference) -> float:
    return float(np.sqrt(np.mean(_cdist(obtained, reference).min(axis=1) ** 2)))


def inverted_gd(obtained, reference) -> float:
    return float(np.sqrt(np.mean(_cdist(reference, obtained).min(axis=1) ** 2)))


def spread_metric(front, reference) -> float:
    front, reference = np.asarray(front, float), np.asarray(reference, float)
    if len(front) < 2:
        return float("nan")
    d = _cdist(front, reference).min(axis=1)
    d_mean = d.mean()
    df = _cdist(front, reference.min(axis=0).reshape(1, -1)).min()
    dl = _cdist(front, reference.max(axis=0).reshape(1, -1)).min()
    den = df + dl + (len(front) - 1) * d_mean
    return float((df + dl + np.abs(d - d_mean).sum()) / den) if den != 0 else float("nan")


def coverage_metric(A, B) -> float:
    B = np.asarray(B, float)
    if len(B) == 0:
        return 0
    return sum(1 for b in B if any(dominates_min(a, b) for a in np.asarray(A, float))) / len(B)


def tchebycheff_rank(accuracy: Sequence[float], size_mb: Sequence[float], fpr: Sequence[float]):
    """-> (scores, ranks): equal-weight Tchebycheff scalarisation against the ideal point; rank 1 = best,
    ties share the smallest rank (pandas rank(method='min'))."""
    F = np.column_stack([1.0 - np.asarray(accuracy, float), np.asarray(size_mb, float), np.asarray(fpr, float)])
    w = np.ones(3) / 3.0
    scores = (w * np.abs(F - F.min(axis=0))).max(axis=1)
    ranks = np.array([1 + int((scores < s).sum()) for s in scores])
    return scores, ranks
