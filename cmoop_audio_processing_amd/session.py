"""Single-candidate session over the C ABI (cmoop_net_*): used by the parity
tests, ``__graft_entry__.smoke()`` and small interactive checks."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib, genes as G
from .evaluator import EvalConfig


class NetSession:
    def __init__(self, gene, config: EvalConfig, T: int, F: int, seed: int):
        self.gene = tuple(int(v) for v in gene)
        G.validate_gene(self.gene)
        self.config = config
        self._h = C.c_void_p()
        g = (C.c_int32 * 6)(*self.gene)
        cfg = config.to_struct()
        _lib.check(_lib.lib().cmoop_net_create(g, C.byref(cfg), C.c_int32(T), C.c_int32(F), C.c_uint32(seed & 0xFFFFFFFF),
                                               C.byref(self._h)))
        n = C.c_int64()
        _lib.check(_lib.lib().cmoop_net_total_params(self._h, C.byref(n)))
        self.n_params = int(n.value)

    def close(self):
        if self._h:
            _lib.check(_lib.lib().cmoop_net_destroy(self._h))
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def get_params(self) -> np.ndarray:
        out = np.empty(self.n_params, np.float32)
        _lib.check(_lib.lib().cmoop_net_get_params(self._h, _lib.ptr(out)))
        return out

    def set_params(self, flat) -> None:
        flat = np.ascontiguousarray(flat, np.float32)
        assert flat.size == self.n_params
        _lib.check(_lib.lib().cmoop_net_set_params(self._h, _lib.ptr(flat)))

    def get_grads(self) -> np.ndarray:
        out = np.empty(self.n_params, np.float32)
        _lib.check(_lib.lib().cmoop_net_get_grads(self._h, _lib.ptr(out)))
        return out

    def train_step(self, X, y, idx=None, row0: int = 0, B: int = None) -> None:
        """X: CUDA float32 [N,T,F]; y: CUDA int32 [N]; idx: CUDA int32 permutation or None."""
        import torch
        torch.cuda.synchronize()
        if B is None:
            B = int(len(idx) if idx is not None else len(X)) - row0
        _lib.check(_lib.lib().cmoop_net_train_step(self._h, _lib.ptr(X), _lib.ptr(y), _lib.ptr(idx), C.c_int64(row0), C.c_int32(B)))

    def evaluate(self, X, y):
        """-> (mean loss, accuracy, int32 CUDA predictions)."""
        import torch
        torch.cuda.synchronize()
        n = int(len(X))
        preds = torch.empty(n, dtype=torch.int32, device=X.device)
        ls, corr = C.c_double(), C.c_int64()
        _lib.check(_lib.lib().cmoop_net_evaluate(self._h, _lib.ptr(X), _lib.ptr(y), C.c_int64(n), C.byref(ls), C.byref(corr), _lib.ptr(preds)))
        return ls.value / max(n, 1), corr.value / max(n, 1), preds

    # -- full training state / epoch-granular driving (re-synchronised parity tests) -----------------------------
    def get_state(self):
        """-> dict(params, m, v, iterations, steps): parameters in canonical order incl. BatchNorm moving statistics,
        Adam's moments in the same layout, optimizer.iterations, global train-step count (dropout counter)."""
        p, m, v = (np.empty(self.n_params, np.float32) for _ in range(3))
        it, st = C.c_int64(), C.c_int64()
        _lib.check(_lib.lib().cmoop_net_get_state(self._h, _lib.ptr(p), _lib.ptr(m), _lib.ptr(v), C.byref(it), C.byref(st)))
        return {"params": p, "m": m, "v": v, "iterations": int(it.value), "steps": int(st.value)}

    def set_state(self, state) -> None:
        p, m, v = (np.ascontiguousarray(state[k], np.float32) for k in ("params", "m", "v"))
        assert p.size == m.size == v.size == self.n_params
        _lib.check(_lib.lib().cmoop_net_set_state(self._h, _lib.ptr(p), _lib.ptr(m), _lib.ptr(v),
                                                  C.c_int64(int(state["iterations"])), C.c_int64(int(state["steps"]))))

    def set_gather_rows(self, n_rows: int) -> None:
        _lib.check(_lib.lib().cmoop_net_set_gather_rows(self._h, C.c_int64(int(n_rows))))

    def run_epoch(self, X, y, epoch: int) -> None:
        """One epoch of Model.fit on the trainer's own path (device permutation, device step state)."""
        import torch
        torch.cuda.synchronize()
        _lib.check(_lib.lib().cmoop_net_run_epoch(self._h, _lib.ptr(X), _lib.ptr(y), C.c_int64(len(X)), C.c_int32(int(epoch))))

    def fit(self, X_train, y_train, X_val, y_val):
        """evaluate_individual's fit + read-outs on this net -> dict(acc, fpr, val_loss, epochs_run, best_epoch,
        val_loss_history, val_accuracy_history)."""
        import torch
        torch.cuda.synchronize()
        d = _lib.DatasetStruct()
        d.x_train, d.y_train, d.n_train = X_train.data_ptr(), y_train.data_ptr(), len(X_train)
        d.x_val, d.y_val, d.n_val = X_val.data_ptr(), y_val.data_ptr(), len(X_val)
        d.T, d.F = int(X_train.shape[1]), int(X_train.shape[2])
        cap = max(1, int(self.config.epochs))
        hl, ha = np.full(cap, np.nan), np.full(cap, np.nan)
        ep, be = C.c_int32(), C.c_int32()
        acc, fpr, vl = C.c_double(), C.c_double(), C.c_double()
        _lib.check(_lib.lib().cmoop_net_fit(self._h, C.byref(d), C.c_int32(cap), _lib.ptr(hl), _lib.ptr(ha), C.byref(ep), C.byref(be),
                                            C.byref(acc), C.byref(fpr), C.byref(vl)))
        n = int(ep.value)
        return {"acc": acc.value, "fpr": fpr.value, "val_loss": vl.value, "epochs_run": n, "best_epoch": int(be.value),
                "val_loss_history": hl[:n].copy(), "val_accuracy_history": ha[:n].copy()}

    def train_metrics(self, reset=True):
        ls, corr = C.c_double(), C.c_int64()
        _lib.check(_lib.lib().cmoop_net_train_metrics(self._h, C.byref(ls), C.byref(corr), C.c_int32(int(reset))))
        return ls.value, int(corr.value)


def epoch_permutation(seed: int, epoch: int, n: int) -> np.ndarray:
    out = np.empty(n, np.int32)
    _lib.check(_lib.lib().cmoop_epoch_permutation(C.c_uint32(seed & 0xFFFFFFFF), C.c_uint32(epoch), C.c_int64(n), _lib.ptr(out)))
    return out
