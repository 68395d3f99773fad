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

    def train_metrics(self, reset=True):
        ls, corr = C.c_double(), C.c_int64()
        _lib.check(_lib.lib().cmoop_net_train_metrics(self._h, C.byref(ls), C.byref(corr), C.c_int32(int(reset))))
        return ls.value, int(corr.value)


def epoch_permutation(seed: int, epoch: int, n: int) -> np.ndarray:
    out = np.empty(n, np.int32)
    _lib.check(_lib.lib().cmoop_epoch_permutation(C.c_uint32(seed & 0xFFFFFFFF), C.c_uint32(epoch), C.c_int64(n), _lib.ptr(out)))
    return out
