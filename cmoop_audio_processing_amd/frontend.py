"""Audio front end + per-bin standardisation on the GPU (libcmoop_hip.so).

The reference loads pre-extracted log-mel features (nsga_penalty.py:64-71,
sa_nsga_penalty.py:58); the north-star adds the extraction beneath that loader:
1 s @ 16 kHz clips -> framing (n_fft 512, Hann 400, hop 160, centre-padded) ->
|STFT|^2 -> 40 Slaney mel bands 20-7600 Hz -> log(mel + 1e-6)  => [N,101,40].
``prepare_dataset`` mirrors nsga_penalty.py:85-155 (StandardScaler per mel bin
over the N*T rows) including the per-script quirks Q1/Q2 of SURVEY §8a.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

HOP, N_MELS = 160, 40


def log_mel(wav):
    """wav: CUDA float32 [N, L] -> CUDA float32 [N, 1 + L//160, 40]."""
    import torch
    if not (isinstance(wav, torch.Tensor) and wav.is_cuda and wav.dtype == torch.float32 and wav.dim() == 2):
        raise ValueError("log_mel expects a CUDA float32 tensor [n_clips, n_samples]")
    wav = wav.contiguous()
    n, L = int(wav.shape[0]), int(wav.shape[1])
    out = torch.empty((n, 1 + L // HOP, N_MELS), dtype=torch.float32, device=wav.device)
    torch.cuda.synchronize()
    _lib.check(_lib.lib().cmoop_logmel(_lib.ptr(wav), C.c_int64(n), C.c_int32(L), _lib.ptr(out)))
    return out


def mfcc(wav, n_mfcc: int = N_MELS):
    """wav: CUDA float32 [N, L] -> CUDA float32 [N, 1 + L//160, n_mfcc]: DCT-II (ortho) of the log-mel frames."""
    import torch
    lm = log_mel(wav)
    n, T = int(lm.shape[0]), int(lm.shape[1])
    if not 1 <= n_mfcc <= N_MELS:
        raise ValueError("1 <= n_mfcc <= 40")
    out = torch.empty((n, T, n_mfcc), dtype=torch.float32, device=lm.device)
    torch.cuda.synchronize()
    _lib.check(_lib.lib().cmoop_mfcc(_lib.ptr(lm), C.c_int64(n * T), C.c_int32(N_MELS), C.c_int32(n_mfcc), _lib.ptr(out)))
    return out


def standardize_fit(x):
    """StandardScaler.fit over x.reshape(-1, F): (mean, scale) float64 numpy arrays."""
    import torch
    x = x.contiguous()
    cols = int(x.shape[-1])
    rows = int(x.numel() // cols)
    mean, scale = np.zeros(cols, np.float64), np.zeros(cols, np.float64)
    torch.cuda.synchronize()
    _lib.check(_lib.lib().cmoop_standardize_fit(_lib.ptr(x), C.c_int64(rows), C.c_int32(cols), _lib.ptr(mean), _lib.ptr(scale)))
    return mean, scale


def standardize_apply(x, mean, scale):
    """In place (x - mean) / scale per last-axis column; returns x."""
    import torch
    assert x.is_contiguous()
    cols = int(x.shape[-1])
    rows = int(x.numel() // cols)
    mean = np.ascontiguousarray(mean, np.float64)
    scale = np.ascontiguousarray(scale, np.float64)
    torch.cuda.synchronize()
    _lib.check(_lib.lib().cmoop_standardize_apply(_lib.ptr(x), C.c_int64(rows), C.c_int32(cols), _lib.ptr(mean), _lib.ptr(scale)))
    return x


def prepare_dataset(X_train, X_validation, X_test=None, mode="refit"):
    """Standardise the splits in place on the GPU.

    mode 'refit'      -- nsga_penalty.py:111,124,137: the scaler is RE-FIT on val and test (quirk Q1)
    mode 'train_only' -- mobo_penalty.py:69-79 and the ablations: fit on train, transform the rest
    mode 'none'       -- sa_nsga_penalty.py:61-85: no scaling (quirk Q2)
    """
    if mode == "none":
        return X_train, X_validation, X_test
    if mode not in ("refit", "train_only"):
        raise ValueError(mode)
    m, s = standardize_fit(X_train)
    standardize_apply(X_train, m, s)
    for X in (X_validation, X_test):
        if X is None:
            continue
        if mode == "refit":
            m2, s2 = standardize_fit(X)
            standardize_apply(X, m2, s2)
        else:
            standardize_apply(X, m, s)
    return X_train, X_validation, X_test
