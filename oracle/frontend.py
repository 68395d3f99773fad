"""Oracle (TEST INFRASTRUCTURE): audio front end + per-bin standardisation, numpy float64.

The reference has NO front end: its scripts
load pre-extracted log-mel features (nsga_penalty.py:64-71, sa_nsga_penalty.py:58).
``librosa==0.11.0`` is pinned (requirements.txt:80) but never called and is not
installed here, so this file restates librosa's *published* algorithm
(``librosa.feature.melspectrogram`` -> ``stft(center=True, pad_mode='constant')``,
periodic Hann of win_length zero-padded to n_fft, ``|.|**2``, Slaney mel basis
with ``norm='slaney'``) followed by ``log(mel + eps)``.  PARITY UNPINNED against
the reference; the build-defined configuration is SURVEY.md §8d:
sr 16 kHz, n_fft 512, win 400, hop 160, 40 mels, 20-7600 Hz -> [N,101,40].

Standardisation follows ``prepare_dataset`` (nsga_penalty.py:103-141):
sklearn ``StandardScaler`` over the N*T rows of each mel bin (ddof=0, zero
variance -> scale 1).
"""
import numpy as np


def hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def mel_filterbank(sr=16000, n_fft=512, n_mels=40, fmin=20.0, fmax=7600.0):
    """librosa.filters.mel(htk=False, norm='slaney') -> [n_mels, 1+n_fft//2] float64."""
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return w * enorm[:, None]


def hann_padded(win_length=400, n_fft=512):
    """scipy.signal.get_window('hann', win_length, fftbins=True) centred in n_fft."""
    n = np.arange(win_length)
    w = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)
    lpad = (n_fft - win_length) // 2
    out = np.zeros(n_fft)
    out[lpad:lpad + win_length] = w
    return out


def log_mel(wav, sr=16000, n_fft=512, win_length=400, hop=160, n_mels=40, fmin=20.0, fmax=7600.0, eps=1e-6):
    """wav [N, L] -> log-mel [N, T, n_mels] float64, T = 1 + L // hop."""
    wav = np.asarray(wav, dtype=np.float64)
    N, L = wav.shape
    T = 1 + L // hop
    pad = n_fft // 2
    x = np.pad(wav, ((0, 0), (pad, pad)))
    win = hann_padded(win_length, n_fft)
    fb = mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
    idx = np.arange(T)[:, None] * hop + np.arange(n_fft)[None, :]
    out = np.empty((N, T, n_mels))
    for i in range(N):
        frames = x[i][idx] * win[None, :]
        p = np.abs(np.fft.rfft(frames, axis=1)) ** 2
        out[i] = np.log(p @ fb.T + eps)
    return out


def mfcc(wav, n_mfcc=40, **kw):
    """DCT-II (orthonormal) of the log-mel frames along the mel axis, first n_mfcc coefficients (scipy.fft.dct)."""
    from scipy.fft import dct
    return dct(log_mel(wav, **kw), type=2, norm="ortho", axis=-1)[..., :n_mfcc]


def scaler_fit(X):
    """StandardScaler.fit over rows of X.reshape(-1, F): (mean, scale) float64."""
    flat = np.asarray(X, dtype=np.float64).reshape(-1, X.shape[-1])
    mean = flat.mean(axis=0)
    var = flat.var(axis=0)
    scale = np.sqrt(var)
    scale[scale == 0.0] = 1.0
    return mean, scale


def scaler_transform(X, mean, scale):
    return ((np.asarray(X, dtype=np.float64) - mean) / scale).astype(np.float32)
