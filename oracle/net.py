"""Oracle (TEST INFRASTRUCTURE): torch-CPU fp32 restatement of the reference's
``build_model`` + ``evaluate_individual`` with Keras-3.6 / TF-2.18 semantics.

Follows, line by line:
* topology A  /root/reference/nsga_penalty.py:255-330
* topology B  /root/reference/sa_nsga_penalty.py:151-175
* protocol    nsga_penalty.py:375-395 (last-epoch accuracy, no weight restore)
              sa_nsga_penalty.py:211-229 (restore_best_weights=True, evaluate())
and the Keras defaults those call sites rely on (tag [K3]):
glorot_uniform kernels / zero biases; TF "SAME" padding (extra pad bottom/right);
MaxPool SAME pads with -inf; BatchNormalization eps=1e-3, momentum=0.99, biased
batch variance for both normalisation and the moving average; inverted dropout;
sparse_categorical_crossentropy on probabilities (clip to [1e-7, 1-1e-7], log,
softmax-CE of the logs); Adam(1e-3, .9, .999, eps=1e-7) in the Keras form
``w -= lr*sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v)+eps)``; fit keeps the last partial
batch and reports sample-weighted means; EarlyStopping(val_loss, patience).

Gradients come from torch autograd, so this file shares no backward code with
the HIP kernels it checks.  PARITY UNPINNED for training dynamics: the
reference holds no fixtures and Keras cannot run in this image (see
oracle/__init__.py).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import rng as orng
from .metrics import FPR_V1, calculate_fpr

FC_LADDER = {1: [64], 2: [128, 64], 3: [256, 128, 64], 4: [512, 256, 128, 64]}  # nsga_penalty.py:311-316
VARIANT_A, VARIANT_B = 0, 1


@dataclass
class OracleConfig:
    variant: int = VARIANT_A
    classes: int = 10            # nsga_penalty.py:176
    epochs: int = 300            # :177
    batch: int = 64              # :178
    patience: int = 5            # :179
    early_stop: bool = True
    restore_best: bool = False   # absent in nsga_penalty.py:382, True in sa_nsga_penalty.py:215
    acc_readout: str = "last"    # history['val_accuracy'][-1] (:384) vs model.evaluate (sa_:219)
    fpr_variant: int = FPR_V1
    lr: float = 1e-3             # optimizer='adam' string default (:377); LEARNING_RATE (:162) is unused
    beta1: float = 0.9
    beta2: float = 0.999
    adam_eps: float = 1e-7
    bn_eps: float = 1e-3
    bn_momentum: float = 0.99
    dropout: float = 0.3         # code says 0.3 (:323); the docstring's 0.2 is stale
    shuffle: bool = True
    # "fp32": the reference's arithmetic.  "bf16": the build's own opt-in bf16-train mode (BASELINE configs[4]; no
    # reference counterpart): conv/dense GEMM operands of every layer between the first conv and the classifier
    # are rounded to bf16 (round-to-nearest-even) in forward, dgrad and wgrad, products accumulate in fp32.
    compute: str = "fp32"


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


class _Bf16Conv(torch.autograd.Function):
    """y = conv_same(q(x), q(w)) + b;  dx = dgrad(q(dy), q(w));  dw = wgrad(q(x), q(dy));  db = sum(dy)."""

    @staticmethod
    def forward(ctx, x, w, b, stride):
        ctx.save_for_backward(x, w)
        ctx.stride = stride
        return conv_same(bf16_round(x), bf16_round(w), b, stride)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        with torch.enable_grad():
            xq = bf16_round(x.detach()).requires_grad_(True)
            wq = bf16_round(w.detach()).requires_grad_(True)
            y = conv_same(xq, wq, None, ctx.stride)
            dx, dw = torch.autograd.grad(y, (xq, wq), bf16_round(g))
        return dx, dw, g.sum(dim=(0, 2, 3)), None


def _same_pad(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


def conv_same(x, w_ohwi, b, stride):
    """Conv2D(padding='same'); x NCHW, kernel in the canonical [O][kh][kw][I] layout."""
    w = w_ohwi.permute(0, 3, 1, 2)
    k = w.shape[2]
    t, bt = _same_pad(x.shape[2], k, stride)
    l, r = _same_pad(x.shape[3], k, stride)
    if t or bt or l or r:
        x = F.pad(x, (l, r, t, bt))
    return F.conv2d(x, w, b, stride=stride)


def maxpool_same(x):
    """MaxPooling2D((2,2), strides 2, padding='same'): -inf pad bottom/right when odd."""
    ph, pw = x.shape[2] % 2, x.shape[3] % 2
    if ph or pw:
        x = F.pad(x, (0, pw, 0, ph), value=float("-inf"))
    return F.max_pool2d(x, 2, 2)


class OracleNet:
    """One candidate CNN.  Parameters are kept in the canonical layouts/order
    documented in cmoop_audio_processing_amd/genes.py so weights can be swapped
    with the HIP library for parity tests."""

    def __init__(self, gene: Sequence[int], cfg: OracleConfig, seed: int, dtype: torch.dtype = torch.float32):
        """dtype=torch.float64 gives the same net evaluated in double precision from the SAME float32 initial weights:
        the ground truth that tells fp32 summation-order noise (of this oracle and of the HIP kernels alike) from real
        differences (tests only; the protocol itself is float32, as the reference's)."""
        self.gene = tuple(int(v) for v in gene)
        self.cfg = cfg
        self.dtype = dtype
        self.seed = int(seed) & 0xFFFFFFFF
        f, k, bn, R, fc, dr = self.gene
        self.use_bn, self.use_dropout = bool(bn), bool(dr)
        self.names: List[str] = []
        self.T: Dict[str, torch.Tensor] = {}
        self.trainable: List[str] = []

        def add(name, shape, role, fans=None):
            idx = len(self.names)
            if role == "kernel":
                arr = orng.glorot_uniform(self.seed, idx, shape, fans[0], fans[1])
            elif role in ("gamma", "moving_var"):
                arr = np.ones(shape, np.float32)
            else:
                arr = np.zeros(shape, np.float32)
            t = torch.from_numpy(arr.copy()).to(dtype)
            if role in ("kernel", "bias", "gamma", "beta"):
                t.requires_grad_(True)
                self.trainable.append(name)
            self.names.append(name)
            self.T[name] = t

        def conv(name, cin, cout, ks):
            add(name + "/kernel", (cout, ks, ks, cin), "kernel", (ks * ks * cin, ks * ks * cout))
            add(name + "/bias", (cout,), "bias")

        def bnl(name, c):
            add(name + "/gamma", (c,), "gamma")
            add(name + "/beta", (c,), "beta")
            add(name + "/moving_mean", (c,), "moving_mean")
            add(name + "/moving_var", (c,), "moving_var")

        conv("conv1", 1, f, k)
        if bn:
            bnl("bn1", f)
        if cfg.variant == VARIANT_A:
            conv("conv2", f, f, k)
            if bn:
                bnl("bn2", f)
        c = f
        for r in range(R):
            conv(f"res{r}_skip", c, 2 * c, 1)
            conv(f"res{r}_conv1", c, 2 * c, k)
            if bn:
                bnl(f"res{r}_bn1", 2 * c)
            if cfg.variant == VARIANT_A:
                conv(f"res{r}_conv2", 2 * c, 2 * c, k)
                if bn:
                    bnl(f"res{r}_bn2", 2 * c)
            c *= 2
        prev = c
        self.fc_names = []
        for i, units in enumerate(FC_LADDER[fc]):
            add(f"fc{i + 1}/kernel", (units, prev), "kernel", (prev, units))
            add(f"fc{i + 1}/bias", (units,), "bias")
            self.fc_names.append(f"fc{i + 1}")
            prev = units
        add("output_layer/kernel", (cfg.classes, prev), "kernel", (prev, cfg.classes))
        add("output_layer/bias", (cfg.classes,), "bias")
        # Adam state
        self.m = {n: torch.zeros_like(self.T[n]) for n in self.trainable}
        self.v = {n: torch.zeros_like(self.T[n]) for n in self.trainable}
        self.iterations = 0   # optimizer.iterations
        self.step = 0         # global train step (dropout counter)

    # ---- parameter exchange --------------------------------------------------
    def count_params(self) -> int:
        return sum(int(t.numel()) for t in self.T.values())

    def get_flat(self) -> np.ndarray:
        return np.concatenate([self.T[n].detach().numpy().ravel() for n in self.names]).astype(np.float32 if self.dtype == torch.float32 else np.float64)

    def set_flat(self, flat: np.ndarray) -> None:
        off = 0
        with torch.no_grad():
            for n in self.names:
                t = self.T[n]
                t.copy_(torch.from_numpy(np.asarray(flat[off:off + t.numel()], np.float32).reshape(t.shape)).to(t.dtype))
                off += t.numel()
        assert off == len(flat)

    def get_state(self) -> dict:
        """Full training state in the layout of the C ABI's cmoop_net_get_state (Adam moments zero in the non-trainable slots)."""
        def flat(d):
            return np.concatenate([(d[n] if n in d else torch.zeros_like(self.T[n])).detach().numpy().ravel() for n in self.names])
        return {"params": self.get_flat(), "m": flat(self.m), "v": flat(self.v), "iterations": self.iterations, "steps": self.step}

    def set_state(self, state: dict) -> None:
        """Load parameters (incl. BatchNorm moving statistics), Adam m / v, optimizer.iterations and the dropout step
        counter -- e.g. the GPU net's state at an epoch boundary (tests re-synchronise there)."""
        self.set_flat(state["params"])
        off = 0
        with torch.no_grad():
            for n in self.names:
                k = self.T[n].numel()
                if n in self.m:
                    for dst, key in ((self.m[n], "m"), (self.v[n], "v")):
                        dst.copy_(torch.from_numpy(np.asarray(state[key][off:off + k], np.float32).reshape(dst.shape)).to(dst.dtype))
                off += k
        self.iterations, self.step = int(state["iterations"]), int(state["steps"])

    def grads_flat(self) -> np.ndarray:
        out = []
        for n in self.names:
            t = self.T[n]
            g = t.grad if (t.requires_grad and t.grad is not None) else torch.zeros_like(t)
            out.append(g.detach().numpy().ravel())
        return np.concatenate(out).astype(np.float32 if self.dtype == torch.float32 else np.float64)

    # ---- layers --------------------------------------------------------------
    def _bn(self, x, name, train):
        cfg = self.cfg
        g, b = self.T[name + "/gamma"], self.T[name + "/beta"]
        mm, mv = self.T[name + "/moving_mean"], self.T[name + "/moving_var"]
        if train:
            mean = x.mean(dim=(0, 2, 3))
            var = ((x - mean[None, :, None, None]) ** 2).mean(dim=(0, 2, 3))
            with torch.no_grad():
                mm.mul_(cfg.bn_momentum).add_(mean.detach() * (1.0 - cfg.bn_momentum))
                mv.mul_(cfg.bn_momentum).add_(var.detach() * (1.0 - cfg.bn_momentum))
        else:
            mean, var = mm, mv
        inv = torch.rsqrt(var + cfg.bn_eps) * g
        return x * inv[None, :, None, None] + (b - mean * inv)[None, :, None, None]

    def _conv(self, x, name, stride=1):
        w, b = self.T[name + "/kernel"], self.T[name + "/bias"]
        if self.cfg.compute == "bf16" and w.shape[3] > 1:       # the C_in = 1 first conv stays fp32
            return _Bf16Conv.apply(x, w, b, stride)
        return conv_same(x, w, b, stride)

    def _dense(self, x, name):
        """hidden Dense layers (the classifier layer stays fp32 in every mode)"""
        w, b = self.T[name + "/kernel"], self.T[name + "/bias"]
        if self.cfg.compute == "bf16":
            return _Bf16Conv.apply(x[:, :, None, None], w[:, None, None, :], b, 1)[:, :, 0, 0]
        return x @ w.t() + b

    def forward(self, x: torch.Tensor, train: bool) -> torch.Tensor:
        """x [B,T,F] float32 -> softmax probabilities [B,classes]."""
        f, k, bn, R, fc, dr = self.gene
        A = self.cfg.variant == VARIANT_A
        x = x[:, None, :, :]
        if A:
            x = self._conv(x, "conv1")
            if bn:
                x = self._bn(x, "bn1", train)
            x = F.relu(x)
            x = self._conv(x, "conv2")
            if bn:
                x = self._bn(x, "bn2", train)
            x = F.relu(x)
            x = maxpool_same(x)
        else:
            x = F.relu(self._conv(x, "conv1"))
            if bn:
                x = self._bn(x, "bn1", train)
            x = maxpool_same(x)
        for r in range(R):
            skip = self._conv(x, f"res{r}_skip", stride=2)
            if A:
                y = self._conv(x, f"res{r}_conv1")
                if bn:
                    y = self._bn(y, f"res{r}_bn1", train)
                y = F.relu(y)
                y = self._conv(y, f"res{r}_conv2")
                if bn:
                    y = self._bn(y, f"res{r}_bn2", train)
            else:
                y = F.relu(self._conv(x, f"res{r}_conv1"))
                if bn:
                    y = self._bn(y, f"res{r}_bn1", train)
            y = maxpool_same(y)
            x = F.relu(y + skip)
        x = x.mean(dim=(2, 3))
        for li, name in enumerate(self.fc_names):
            x = F.relu(self._dense(x, name))
            if dr and train:
                keep = orng.dropout_keep(self.seed, li, self.step, x.shape[0], x.shape[1], self.cfg.dropout)
                scale = np.float32(1.0 / (1.0 - self.cfg.dropout))
                x = x * torch.from_numpy(keep.astype(np.float32)).to(x.dtype) * float(scale)
        z = x @ self.T["output_layer/kernel"].t() + self.T["output_layer/bias"]
        return torch.softmax(z, dim=1)

    @staticmethod
    def loss_per_sample(p: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        """Keras-3 TF backend sparse_categorical_crossentropy(from_logits=False)."""
        lo, hi = np.float32(1e-7), np.float32(1.0) - np.float32(1e-7)
        logp = torch.log(torch.clamp(p, float(lo), float(hi)))
        return -(logp.gather(1, y[:, None].long())[:, 0] - torch.logsumexp(logp, dim=1))

    # ---- one optimiser step --------------------------------------------------
    def train_step(self, xb: np.ndarray, yb: np.ndarray) -> Tuple[float, int]:
        """fwd + bwd + Adam on one batch; returns (sum of per-sample loss, n correct)."""
        cfg = self.cfg
        for n in self.trainable:
            self.T[n].grad = None
        p = self.forward(torch.from_numpy(np.ascontiguousarray(xb, np.float32)).to(self.dtype), True)
        y = torch.from_numpy(np.asarray(yb).astype(np.int64).ravel())
        lps = self.loss_per_sample(p, y)
        lps.mean().backward()
        correct = int((p.argmax(dim=1) == y).sum())
        self.iterations += 1
        t = self.iterations
        alpha = np.float32(cfg.lr * math.sqrt(1.0 - cfg.beta2 ** t) / (1.0 - cfg.beta1 ** t))
        with torch.no_grad():
            for n in self.trainable:
                g = self.T[n].grad
                self.m[n].add_((g - self.m[n]) * (1.0 - cfg.beta1))
                self.v[n].add_((g * g - self.v[n]) * (1.0 - cfg.beta2))
                self.T[n].sub_(self.m[n] * float(alpha) / (torch.sqrt(self.v[n]) + cfg.adam_eps))
        self.step += 1
        return float(lps.detach().sum()), correct

    # ---- inference -----------------------------------------------------------
    @torch.no_grad()
    def evaluate(self, X: np.ndarray, y: np.ndarray, chunk: int = 256):
        """(mean loss, accuracy, int64 predictions) with BN moving stats, no dropout."""
        n = len(X)
        tot, corr, preds = 0.0, 0, []
        yv = np.asarray(y).astype(np.int64).ravel()
        for s in range(0, n, chunk):
            p = self.forward(torch.from_numpy(np.ascontiguousarray(X[s:s + chunk], np.float32)).to(self.dtype), False)
            yy = torch.from_numpy(yv[s:s + chunk])
            tot += float(self.loss_per_sample(p, yy).double().sum())
            pr = p.argmax(dim=1)
            corr += int((pr == yy).sum())
            preds.append(pr.numpy())
        return tot / max(n, 1), corr / max(n, 1), np.concatenate(preds) if preds else np.zeros(0, np.int64)


def run_epoch(net: OracleNet, Xtr, ytr, epoch: int) -> Tuple[float, int]:
    """One epoch of Model.fit: the seeded permutation of (seed, epoch), batches of cfg.batch, last partial batch kept.
    Returns (sum of per-sample training losses, correct)."""
    n = len(Xtr)
    ytr = np.asarray(ytr).ravel()
    perm = orng.epoch_permutation(net.seed, epoch, n) if net.cfg.shuffle else np.arange(n, dtype=np.int32)
    ls, cs = 0.0, 0
    for s in range(0, n, net.cfg.batch):
        idx = perm[s:s + net.cfg.batch]
        l, c = net.train_step(Xtr[idx], ytr[idx])
        ls += l
        cs += c
    return ls, cs


class EarlyStopping:
    """keras/src/callbacks/early_stopping.py (3.6) with monitor='val_loss', min_delta=0, as the reference configures it
    (nsga_penalty.py:382): feed one validation loss per epoch; ``update`` returns True when training stops."""

    def __init__(self, patience: int):
        self.patience, self.best, self.wait, self.best_epoch, self.epoch = patience, float("inf"), 0, -1, -1

    def update(self, val_loss: float) -> bool:
        self.epoch += 1
        self.wait += 1
        if val_loss < self.best:
            self.best, self.best_epoch, self.wait = val_loss, self.epoch, 0
            return False
        return self.wait >= self.patience and self.epoch > 0


def fit(net: OracleNet, Xtr, ytr, Xval, yval, max_steps: Optional[int] = None):
    """Model.fit + EarlyStopping (keras/src/callbacks/early_stopping.py, 3.6).

    Returns history dict with per-epoch val_loss / val_accuracy, epochs_run and
    (when cfg.restore_best) leaves the best weights loaded, as on_train_end does.
    """
    cfg = net.cfg
    n = len(Xtr)
    best, wait, best_flat = float("inf"), 0, None
    hist = {"loss": [], "accuracy": [], "val_loss": [], "val_accuracy": []}
    steps_done = 0
    ytr = np.asarray(ytr).ravel()
    for epoch in range(cfg.epochs):
        perm = orng.epoch_permutation(net.seed, epoch, n) if cfg.shuffle else np.arange(n, dtype=np.int32)
        ls, cs = 0.0, 0
        for s in range(0, n, cfg.batch):
            idx = perm[s:s + cfg.batch]
            l, c = net.train_step(Xtr[idx], ytr[idx])
            ls += l
            cs += c
            steps_done += 1
            if max_steps is not None and steps_done >= max_steps:
                break
        vl, va, _ = net.evaluate(Xval, yval)
        hist["loss"].append(ls / n)
        hist["accuracy"].append(cs / n)
        hist["val_loss"].append(vl)
        hist["val_accuracy"].append(va)
        if max_steps is not None and steps_done >= max_steps:
            break
        if not cfg.early_stop:
            continue
        if cfg.restore_best and best_flat is None:
            best_flat = net.get_flat()
        wait += 1
        if vl < best:
            best = vl
            if cfg.restore_best:
                best_flat = net.get_flat()
            wait = 0
            continue
        if wait >= cfg.patience and epoch > 0:
            break
    if cfg.early_stop and cfg.restore_best and best_flat is not None:
        net.set_flat(best_flat)
    hist["epochs_run"] = len(hist["val_loss"])
    return hist


def evaluate_individual(gene, cfg: OracleConfig, Xtr, ytr, Xval, yval, seed: int, dtype: torch.dtype = torch.float32):
    """Oracle twin of evaluate_individual -> (accuracy, size_mb, fpr, epochs_run).  dtype=float64: the same protocol
    from the same float32 initial weights in double precision (tests: a third sample of the summation-order spread)."""
    net = OracleNet(gene, cfg, seed, dtype=dtype)
    hist = fit(net, Xtr, ytr, Xval, yval)
    _, acc_eval, preds = net.evaluate(Xval, yval)
    acc = hist["val_accuracy"][-1] if cfg.acc_readout == "last" else acc_eval
    fpr = calculate_fpr(np.asarray(yval).ravel(), preds, cfg.classes, cfg.fpr_variant)
    size_mb = net.count_params() * 4 / (1024 ** 2)
    return float(acc), float(size_mb), float(fpr), int(hist["epochs_run"])
