"""Dataset adaptors on either side of the hot path, host only.

  load_npy_splits ............ load_data, nsga_penalty.py:57-83 (six .npy files, labels get a trailing axis)
  stratified_50_25_25 ........ sa_nsga_penalty.py:71-85 (two stratified train_test_split calls, random_state=42)
  load_mel_h5 ................ load_from_hdf5, sa_nsga_penalty.py:42-63 (needs h5py, absent in this image)
  read_two_stage_population .. initialize_population of the "psi"/2-stage scripts, psi_sa_nsga_local.py:255-269
                               (Final.xlsx; openpyxl is absent here, so CSV with the same columns is accepted too)
  records_to_csv ............. the per-generation / Pareto CSV dumps, nsga_penalty.py:738-763
"""
from __future__ import annotations

import csv
import os
from typing import Dict, List, Sequence

import numpy as np

from . import genes as G


def load_npy_splits(data_path: str):
    """-> X_train, X_test, X_validation, y_train, y_test, y_validation (reference return order)."""
    def ld(name):
        p = os.path.join(data_path, name)
        if not os.path.exists(p):
            raise FileNotFoundError(f"{p} (expected X_train/X_test/X_val/y_train/y_test/y_val .npy as in nsga_penalty.py:64-71)")
        return np.load(p, allow_pickle=False)
    X_train, X_test, X_val = ld("X_train.npy"), ld("X_test.npy"), ld("X_val.npy")
    y_train, y_test, y_val = (ld(n)[..., np.newaxis] for n in ("y_train.npy", "y_test.npy", "y_val.npy"))
    return X_train, X_test, X_val, y_train, y_test, y_val


def stratified_50_25_25(X, y, random_state: int = 42):
    """train 50 % / validation 25 % / test 25 %, stratified, exactly the two calls of sa_nsga_penalty.py:71-85."""
    from sklearn.model_selection import train_test_split
    X_train, X_temp, y_train, y_temp = train_test_split(X, y, test_size=0.5, random_state=random_state, stratify=y)
    X_val, X_test, y_val, y_test = train_test_split(X_temp, y_temp, test_size=0.5, random_state=random_state, stratify=y_temp)
    return X_train, y_train, X_val, y_val, X_test, y_test


def load_mel_h5(path: str) -> Dict:
    try:
        import h5py
    except ImportError as e:   # the reference needs it too (sa_nsga_penalty.py:39)
        raise ImportError("load_mel_h5 needs h5py, which is not installed in this image") from e
    with h5py.File(path, "r") as hf:
        return {name: hf[name][:] for name in hf.keys()}


def _truthy(v) -> bool:
    if isinstance(v, str):
        return v.strip().lower() in ("1", "true", "yes")
    return bool(int(v)) if not isinstance(v, bool) else v


def read_two_stage_population(path: str, min_accuracy: float, max_model_size: float, max_fpr: float) -> List[Dict]:
    """Initial population WITH stored objectives (no true evaluations), columns
    Accuracy, Size_MB, FPR + the six genes.  ``.xlsx`` needs pandas+openpyxl; ``.csv`` needs nothing."""
    if path.lower().endswith((".xlsx", ".xls")):
        import pandas as pd
        rows = pd.read_excel(path).to_dict("records")
    else:
        with open(path, newline="") as f:
            rows = list(csv.DictReader(f))
    pop = []
    for r in rows:
        hp = {"filters": int(float(r["filters"])), "kernel_size": int(float(r["kernel_size"])), "use_bn": _truthy(r["use_bn"]),
              "residual_blocks": int(float(r["residual_blocks"])), "fc_layers": int(float(r["fc_layers"])),
              "use_dropout": _truthy(r["use_dropout"])}
        G.validate_gene(G.normalize_hparams(hp))
        acc, size, fpr = float(r["Accuracy"]), float(r["Size_MB"]), float(r["FPR"])
        cv = max(0, min_accuracy - acc) + max(0, size - max_model_size) + max(0, fpr - max_fpr)
        pop.append({"hparams": hp, "objs": [-acc, size, fpr], "CV": cv})
    return pop


def records_to_csv(path: str, records: Sequence[Dict]) -> None:
    if not records:
        open(path, "w").close()
        return
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(records[0].keys()))
        w.writeheader()
        w.writerows(records)
