"""GPU parity: whole-candidate training / inference through the C ABI vs the oracle.

The oracle (oracle/net.py) takes gradients from torch autograd, so these tests
check every hand-written backward kernel.  fp32 everywhere; tolerances are
relative to each tensor's max magnitude and stated at the assert.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from cmoop_audio_processing_amd import EvalConfig, PopulationEvaluator, genes as G
from cmoop_audio_processing_amd.session import NetSession
from oracle import metrics as OM
from oracle import net as ON

pytestmark = pytest.mark.gpu


# CMOOP_GEMM_MODE (read once by the library) switches the arithmetic of the MFMA GEMMs for a whole process; the
# child-process tests at the bottom re-run this file under it.  bf16x3 is fp32-accurate (oracle unchanged); bf16
# rounds GEMM operands, which the oracle restates (OracleConfig.compute).
ENV_MODE = os.environ.get("CMOOP_GEMM_MODE", "")


def ocfg(cfg: EvalConfig) -> ON.OracleConfig:
    compute = "bf16" if (cfg.compute == "bf16" or (cfg.compute == "fp32" and ENV_MODE == "bf16")) else "fp32"
    return ON.OracleConfig(compute=compute, variant=G.VARIANT_NAMES[cfg.variant], classes=cfg.classes, epochs=cfg.epochs, batch=cfg.batch,
                           patience=cfg.patience, early_stop=cfg.early_stop, restore_best=cfg.restore_best,
                           acc_readout="last" if cfg.acc_readout == "last" else "evaluate",
                           fpr_variant=OM.FPR_V1 if cfg.fpr_variant == "v1" else (OM.FPR_V1_QUIRK if cfg.fpr_variant == "v1_quirk" else OM.FPR_V3),
                           lr=cfg.lr, dropout=cfg.dropout, shuffle=cfg.shuffle)


def make_data(n, T, F, classes, seed):
    rs = np.random.RandomState(seed)
    y = rs.randint(0, classes, size=n).astype(np.int32)
    proto = rs.randn(classes, T, F).astype(np.float32)
    X = (0.8 * proto[y] + rs.randn(n, T, F)).astype(np.float32)
    return X, y


def make_split(n_train, n_val, T, F, classes, seed, noise=1.0, label_noise=0.0):
    """A task these GAP-headed CNNs LEARN: class c is a stripe texture of class-specific spatial frequency / phase plus a
    class-specific level (both survive global average pooling), in N(0, noise^2) noise; train and validation are drawn
    from the same prototypes.  ``label_noise`` re-draws that fraction of the labels at random: accuracy then tops out
    below 1 and the validation loss has a real minimum (early stopping triggers) while the predictions stay CONFIDENT,
    i.e. far from the decision boundaries where any two fp32 implementations flip samples.
    (the protocol tests used different random prototypes for train and validation, so validation accuracy
    was chance and early stopping / read-outs were compared on noise predictions.)"""
    rs = np.random.RandomState(seed)
    f, t = np.arange(F)[None, :], np.arange(T)[:, None]
    proto = np.stack([np.sin(2 * np.pi * (1 + c % 5) * f / F + 0.7 * c) * np.cos(2 * np.pi * (1 + c // 5) * t / T)
                      + (c - classes / 2) / classes for c in range(classes)]).astype(np.float32)
    y = rs.randint(0, classes, size=n_train + n_val).astype(np.int32)
    X = (proto[y] + noise * rs.randn(n_train + n_val, T, F)).astype(np.float32)
    if label_noise > 0:
        flip = rs.rand(n_train + n_val) < label_noise
        y = np.where(flip, rs.randint(0, classes, size=n_train + n_val), y).astype(np.int32)
    return X[:n_train], y[:n_train], X[n_train:], y[n_train:]


def oracle_pair(gene, cfg, Xtr, ytr, Xva, yva, seed):
    """The oracle twice: with torch's default CPU conv algorithm (mkldnn) and with the native one.  Same arithmetic, a
    different fp32 summation order -- the spread between the two is the oracle's OWN sensitivity, the floor under any
    comparison of trained nets (BatchNorm nets amplify it: tests/test_oracle_golden.py pins that on the CPU)."""
    a = ON.evaluate_individual(gene, ocfg(cfg), Xtr, ytr, Xva, yva, seed=seed)
    with torch.backends.mkldnn.flags(enabled=False):
        b = ON.evaluate_individual(gene, ocfg(cfg), Xtr, ytr, Xva, yva, seed=seed)
    return a, b


def oracle_band(gene, cfg, Xtr, ytr, Xva, yva, seed):
    """oracle_pair plus the same protocol in float64 (from the same float32 initial weights): three equivalent CPU
    evaluations of one algorithm; their range is the band a correct fp32 implementation lands in."""
    a, b = oracle_pair(gene, cfg, Xtr, ytr, Xva, yva, seed)
    c = ON.evaluate_individual(gene, ocfg(cfg), Xtr, ytr, Xva, yva, seed=seed, dtype=torch.float64)
    return a, b, c


def gate(gpu, *oracle_values, tol=1e-3):
    """north-star gate |gpu - oracle| <= 1e-3 against the nearest of the equivalent oracle evaluations -- widened ONLY
    to their own range when that is larger (0 for nets without BatchNorm on a learnable task: then it IS 1e-3)"""
    width = max(oracle_values) - min(oracle_values)
    return min(abs(gpu - v) for v in oracle_values) <= max(tol, width) + 1e-12


def per_tensor_err(gene, variant, classes, a, b):
    """max-abs error of each canonical tensor relative to that tensor's max magnitude.

    A conv bias directly in front of a train-mode BatchNorm has an analytically ZERO gradient
    (BN's dx sums to 0 over the batch): both sides hold rounding noise of a long fp32 sum, so
    for those tensors the check is |g| <= 2e-5 * (largest gradient entry of the net) on BOTH
    sides, reported on the same scale as the other tensors' 5e-4 gate."""
    out, off = {}, 0
    gmax = float(np.abs(b).max())
    tensors = G.param_tensors(gene, variant, classes)
    for i, (name, shape, role) in enumerate(tensors):
        n = int(np.prod(shape))
        ra, rb = a[off:off + n].astype(np.float64), b[off:off + n].astype(np.float64)
        zero_grad = role == "bias" and i + 1 < len(tensors) and tensors[i + 1][2] == "gamma" and variant == 0
        if zero_grad:
            out[name] = float(max(np.abs(ra).max(), np.abs(rb).max()) / (2e-5 * gmax + 1e-30)) * 5e-4 * 0.999
        else:
            out[name] = float(np.abs(ra - rb).max() / max(np.abs(rb).max(), 1e-4 * gmax, 1e-30))
        off += n
    return out


GENES = [
    # gene (filters, kernel, bn, res_blocks, fc_layers, dropout), variant
    ((16, 3, 0, 1, 1, 0), "A"),
    ((16, 3, 1, 1, 2, 1), "A"),
    ((32, 5, 1, 2, 3, 0), "A"),
    ((16, 5, 0, 3, 4, 1), "A"),
    ((16, 3, 0, 1, 1, 0), "B"),
    ((16, 3, 1, 2, 2, 1), "B"),
    ((32, 5, 1, 3, 4, 0), "B"),
    ((64, 3, 0, 2, 1, 1), "B"),
]


@pytest.mark.parametrize("gene,variant", GENES)
def test_init_step_grads_and_eval_parity(gene, variant):
    T, F, classes, B, seed = 21, 12, 10, 24, 1234
    cfg = EvalConfig(variant=variant, classes=classes, batch=32, eval_batch=16)
    X, y = make_data(80, T, F, classes, 1)
    Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
    v = G.VARIANT_NAMES[variant]
    onet = ON.OracleNet(gene, ocfg(cfg), seed)
    with NetSession(gene, cfg, T, F, seed) as net:
        assert net.n_params == G.param_count(gene, v, classes) == onet.count_params()
        # 1. seeded glorot init: bit-exact
        assert np.array_equal(net.get_params(), onet.get_flat())
        # 2. inference from identical weights: loss 1e-5 rel, identical predictions
        l_o, a_o, p_o = onet.evaluate(X, y)
        l_g, a_g, p_g = net.evaluate(Xd, yd)
        assert abs(l_g - l_o) < 1e-5 * max(1.0, abs(l_o)), (l_g, l_o)
        assert np.array_equal(p_g.cpu().numpy(), p_o) and a_g == a_o
        # 3. one training step on rows idx[4:4+B]: gradients, updated weights, BN moving stats
        idx = np.random.RandomState(2).permutation(80).astype(np.int32)
        idxd = torch.from_numpy(idx).cuda()
        net.train_step(Xd, yd, idxd, row0=4, B=B)
        lo, co = onet.train_step(X[idx[4:4 + B]], y[idx[4:4 + B]])
        lg, cg = net.train_metrics()
        assert abs(lg - lo) < 2e-5 * max(1.0, abs(lo)) and cg == co
        gerr = per_tensor_err(gene, v, classes, net.get_grads(), onet.grads_flat())
        worst = max(gerr, key=gerr.get)
        print(f"{variant}{gene} worst grad err {worst}: {gerr[worst]:.2e}")
        # fp32, different summation order; BN nets amplify (cancellation in dx): 5e-4 of the tensor's max
        assert gerr[worst] < 5e-4, gerr
        perr = per_tensor_err(gene, v, classes, net.get_params(), onet.get_flat())
        worstp = max(perr, key=perr.get)
        # Adam's first step is +-lr for every weight regardless of |g|: sign flips of ~0 gradients
        # move a weight by 2*lr, so compare at 2.5*lr absolute instead of relative
        d = np.abs(net.get_params() - onet.get_flat())
        print(f"   worst param err {worstp}: {perr[worstp]:.2e}; max abs diff {d.max():.2e}")
        assert d.max() <= 2.5 * cfg.lr
        # 4. four more steps, then inference with the moving statistics
        for s in range(4):
            r0 = 4 + (s + 1) * 8
            net.train_step(Xd, yd, idxd, row0=r0, B=B)
            onet.train_step(X[idx[r0:r0 + B]], y[idx[r0:r0 + B]])
        l_o, a_o, p_o = onet.evaluate(X, y)
        l_g, a_g, p_g = net.evaluate(Xd, yd)
        print(f"   after 5 steps: loss gpu {l_g:.6f} oracle {l_o:.6f}; preds differing {(p_g.cpu().numpy() != p_o).sum()}")
        assert abs(l_g - l_o) < 5e-3 * max(1.0, abs(l_o))


def test_partial_batch_and_full_feature_size():
    """T x F = 101 x 40 (BASELINE feature size), batch 5 < configured 64 (Keras keeps the last partial batch)."""
    gene, variant, classes, seed = (16, 3, 1, 1, 1, 0), "A", 10, 7
    cfg = EvalConfig(variant=variant, classes=classes, batch=64, eval_batch=8)
    X, y = make_data(12, 101, 40, classes, 5)
    Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
    onet = ON.OracleNet(gene, ocfg(cfg), seed)
    with NetSession(gene, cfg, 101, 40, seed) as net:
        net.train_step(Xd, yd, None, row0=3, B=5)
        onet.train_step(X[3:8], y[3:8])
        gerr = per_tensor_err(gene, 0, classes, net.get_grads(), onet.grads_flat())
        assert max(gerr.values()) < 5e-4, gerr
        l_o, a_o, p_o = onet.evaluate(X, y)
        l_g, a_g, p_g = net.evaluate(Xd, yd)
        assert abs(l_g - l_o) < 1e-4 * max(1.0, abs(l_o))


def test_classes_35_and_11():
    for classes in (11, 35):
        gene, seed = (16, 3, 0, 1, 2, 0), 3
        cfg = EvalConfig(variant="B", classes=classes, batch=16, eval_batch=16)
        X, y = make_data(32, 21, 12, classes, classes)
        Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
        onet = ON.OracleNet(gene, ocfg(cfg), seed)
        with NetSession(gene, cfg, 21, 12, seed) as net:
            net.train_step(Xd, yd, None, row0=0, B=16)
            onet.train_step(X[:16], y[:16])
            gerr = per_tensor_err(gene, 1, classes, net.get_grads(), onet.grads_flat())
            assert max(gerr.values()) < 5e-4, gerr


def test_birdclef_shaped_path_config3():
    """BASELINE configs[3]: the sa_nsga_penalty.py data path -- 11 classes, log-mel patches used UNSCALED (quirk Q2:
    no StandardScaler, sa_nsga_penalty.py:61-85), stratified 50/25/25 split with random_state 42 (:71-85), topology B.
    The true BirdCLEF patch shape is not in the reference; 128x128 is SURVEY §8d's example.  One training step and an
    inference pass at that shape against the oracle, then the script's protocol end to end on the split."""
    from cmoop_audio_processing_amd import datasets
    classes, T, F, seed = 11, 128, 128, 42
    gene = (16, 3, 1, 2, 2, 1)
    rs = np.random.RandomState(7)
    y_all = np.repeat(np.arange(classes), 8).astype(np.int32)
    proto = rs.randn(classes, T, F).astype(np.float32)
    X_all = (3.0 + 2.0 * (0.8 * proto[y_all] + rs.randn(len(y_all), T, F))).astype(np.float32)   # unscaled, mean 3
    Xtr, ytr, Xva, yva, Xte, yte = datasets.stratified_50_25_25(X_all, y_all, random_state=42)
    assert len(Xtr) == 44 and len(Xva) == 22 and len(Xte) == 22
    cfg = EvalConfig.preset("sa_nsga_penalty", classes=classes, epochs=3, patience=2, batch=16, eval_batch=8, seed=seed, n_slots=1)
    onet = ON.OracleNet(gene, ocfg(cfg), seed)
    Xd, yd = torch.from_numpy(Xtr).cuda(), torch.from_numpy(ytr.astype(np.int32)).cuda()
    with NetSession(gene, cfg, T, F, seed) as net:
        net.train_step(Xd, yd, None, row0=0, B=16)
        onet.train_step(Xtr[:16], ytr[:16])
        gerr = per_tensor_err(gene, 1, classes, net.get_grads(), onet.grads_flat())
        # unscaled inputs (mean 3, no StandardScaler) put a large common offset into every first-layer sum, so the
        # fp32 summation-order noise is ~4x that of standardised features: observed 9e-4, gate 2e-3
        assert max(gerr.values()) < 2e-3, gerr
        l_o, a_o, p_o = onet.evaluate(Xva, yva)
        l_g, a_g, p_g = net.evaluate(torch.from_numpy(Xva).cuda(), torch.from_numpy(yva.astype(np.int32)).cuda())
        assert abs(l_g - l_o) < 1e-4 * max(1.0, abs(l_o)) and a_g == a_o
    ev = PopulationEvaluator(Xtr, ytr, Xva, yva, cfg)
    acc, size_mb, fpr = ev.evaluate_individual(G.gene_to_hparams(gene))
    (o_acc, o_size, o_fpr, o_epochs), (b_acc, _, b_fpr, b_epochs) = oracle_pair(gene, cfg, Xtr, ytr, Xva, yva, seed)
    assert size_mb == o_size == G.model_size_mb(gene, 1, classes)
    assert gate(acc, o_acc, b_acc) and gate(fpr, o_fpr, b_fpr) and ev.last_epochs_run[0] in (o_epochs, b_epochs)


def test_birdclef_shaped_population_config3():
    """BASELINE configs[3] beyond one gene: 8 random genes (one forced to 64 filters) of the DS-CNN-style
    search space on BirdCLEF-shaped data -- 128x128 patches, 11 classes, topology B, features UNSCALED (prepare mode
    'none', sa_nsga_penalty.py:61-85), stratified 50/25/25 split (:71-85), batch 64 (:106), the script's own protocol.
    Properties at that size: size_mb bit-exact for all 8, results bit-identical across the number of candidates in
    flight; oracle parity (every epoch re-synchronised) on the cheapest candidate."""
    import random
    from cmoop_audio_processing_amd import datasets, frontend
    classes, T, F = 11, 128, 128
    rs = np.random.RandomState(17)
    y_all = np.repeat(np.arange(classes), 24).astype(np.int32)
    proto = rs.randn(classes, T, F).astype(np.float32)
    X_all = (3.0 + 2.0 * (0.8 * proto[y_all] + rs.randn(len(y_all), T, F))).astype(np.float32)
    Xtr, ytr, Xva, yva, Xte, yte = datasets.stratified_50_25_25(X_all, y_all, random_state=42)
    assert len(Xtr) == 132 and len(Xva) == 66
    Xtr_d, Xva_d = torch.from_numpy(Xtr).cuda(), torch.from_numpy(Xva).cuda()
    frontend.prepare_dataset(Xtr_d, Xva_d, None, mode="none")                       # quirk Q2: no scaling
    assert np.array_equal(Xtr_d.cpu().numpy(), Xtr)
    rng = random.Random(3)
    pop = [G.random_hparams(rng) for _ in range(8)]
    pop[0] = dict(pop[0], filters=64, residual_blocks=2)
    genes = [G.normalize_hparams(hp) for hp in pop]
    base = dict(classes=classes, epochs=2, patience=2, batch=64, eval_batch=64, seed=6)
    res = {}
    for slots in (4, 2):
        ev = PopulationEvaluator(Xtr_d, ytr, Xva_d, yva, EvalConfig.preset("sa_nsga_penalty", n_slots=slots, **base))
        res[slots] = ev.compute_objectives_and_constraints(pop)
    assert [r["objs"] for r in res[4]] == [r["objs"] for r in res[2]] and [r["CV"] for r in res[4]] == [r["CV"] for r in res[2]]
    for g, r in zip(genes, res[4]):
        acc, size, fpr = -r["objs"][0], r["objs"][1], r["objs"][2]
        assert size == G.model_size_mb(g, 1, classes) and 0.0 <= acc <= 1.0 and 0.0 <= fpr <= 1.0
        assert r["CV"] == max(0.0, 0.75 - acc) + max(0.0, size - 2.5) + max(0.0, fpr - 0.09)
    i = min(range(8), key=lambda k: G.fwd_flops_per_sample(genes[k], 1, classes, T, F))
    cfg = EvalConfig.preset("sa_nsga_penalty", n_slots=1, **base)
    # unscaled inputs (mean 3) put a common offset into every first-layer sum: fp32 summation-order noise is ~4x that of
    # standardised features (see test_birdclef_shaped_path_config3), hence 4e-4 on the per-epoch validation loss
    resynchronised_fit_check(genes[i], cfg, Xtr, ytr, Xva, yva, cfg.seed + i,
                             expected=(-res[4][i]["objs"][0], res[4][i]["objs"][2], ev.last_epochs_run[i]), loss_tol=4e-4, tag="configs[3]")


PROTOCOLS = [
    ("nsga_penalty", (16, 3, 0, 1, 1, 0)),       # A, last-epoch accuracy, no restore, y_true quirk
    ("sa_nsga_local", (16, 5, 0, 2, 1, 0)),      # B, V3
    ("mobo_penalty", (16, 5, 0, 1, 2, 0)),       # A, restore_best + LAST-epoch accuracy + FPR of the restored weights (mobo_penalty.py:227,236,239: Q6)
]
CHAOTIC = ("sa_nsga_penalty", (16, 3, 1, 1, 2, 1))   # B, restore_best, evaluate(), V1 -- BatchNorm + dropout


def protocol_case(preset, gene, epochs, patience, seed=11):
    classes = 10 if preset != "sa_nsga_penalty" else 11
    cfg = EvalConfig.preset(preset, classes=classes, epochs=epochs, patience=patience, batch=32, eval_batch=64, seed=seed, n_slots=1)
    Xtr, ytr, Xva, yva = make_split(192, 128, 21, 12, classes, 21, noise=0.3, label_noise=0.25)
    ev = PopulationEvaluator(Xtr, ytr, Xva, yva, cfg)
    acc, size_mb, fpr = ev.evaluate_individual(G.gene_to_hparams(gene))
    a, b = oracle_pair(gene, cfg, Xtr, ytr, Xva, yva, seed)
    print(preset, f"seed {seed} epochs {epochs}: gpu", (acc, fpr, ev.last_epochs_run), "oracle", (a[0], a[2], a[3]), "oracle, native conv", (b[0], b[2], b[3]))
    assert size_mb == a[1] == G.model_size_mb(gene, G.VARIANT_NAMES[cfg.variant], classes)
    return (acc, fpr, ev.last_epochs_run[0]), a, b


@pytest.mark.parametrize("preset,gene", PROTOCOLS)
def test_evaluate_individual_protocol_parity(preset, gene):
    """End-to-end parity at the north-star gate on a task the nets LEARN: |d accuracy| <= 1e-3, |d FPR| <= 1e-3,
    epochs_run equal, size_mb bit-exact, with early stopping (patience 2; it triggers at epochs 21 / 16 of 25 on the
    oracle), best-weight restore and the per-script read-outs active; ~100-130 optimiser steps.  Train and validation
    share class prototypes, 25 % of the labels are re-drawn at random (validation accuracy 0.45-0.8, confident
    predictions).  For these nets (no BatchNorm) the oracle's two CPU conv algorithms agree exactly, so the gate IS 1e-3."""
    (acc, fpr, ep), a, b = protocol_case(preset, gene, 25, 2)
    assert 0.4 <= a[0] <= 0.97 and a[3] < 25, f"the parity task must be learnable, unsaturated and early-stopped: {a[0]}, {a[3]}"
    assert gate(acc, a[0], b[0]) and gate(fpr, a[2], b[2])
    if a[3] == b[3]:
        assert ep == a[3]
    else:
        # The oracle's own two conv algorithms stop at different epochs on this case (mobo_penalty gene, patience 2: 12 vs 23 epochs,
        # accuracies 0.48 vs 0.77 -- observed on the GPU box's CPU): the stopping epoch of an un-synchronised 25-epoch run is then a
        # property of the summation order, not of the implementation.  The GPU's accuracy / FPR must still equal one twin's (gate
        # above: it reproduces the native-conv twin to the last digit); its early-stopping DECISIONS are checked exactly, epoch by
        # epoch, in test_full_protocol_resynchronised_every_epoch[mobo_penalty-...].
        print(f"oracle twins disagree on the stopping epoch ({a[3]} vs {b[3]}); gpu ran {ep}")
        assert min(a[3], b[3]) <= ep <= 25


def test_protocol_parity_batchnorm_dropout_short_horizon():
    """The BatchNorm + dropout candidate (restore_best, evaluate() read-out, FPR V1) over a horizon on which rounding
    differences have not yet been amplified: 5 epochs (30 optimiser steps), no early stop.  Strict gate."""
    preset, gene = CHAOTIC
    (acc, fpr, ep), a, b = protocol_case(preset, gene, 5, 5)
    assert ep == a[3] == b[3] == 5
    assert gate(acc, a[0], b[0]) and gate(fpr, a[2], b[2])


def _es_replay(cfg, val_loss_history):
    """epochs_run / best_epoch that Keras' EarlyStopping (oracle.net.EarlyStopping, the restatement oracle.net.fit uses)
    produces from a validation-loss history"""
    if not cfg.early_stop:
        return len(val_loss_history), -1
    es = ON.EarlyStopping(cfg.patience)
    for e, vl in enumerate(val_loss_history):
        if es.update(float(vl)):
            return e + 1, es.best_epoch
    return len(val_loss_history), es.best_epoch


def _bn_stat_deviation(tensors, pa, pb, conv_bias_gauge=False, steps=0, lr=1e-3):
    """largest |moving statistic a - b| relative to that tensor's largest entry (floor 1e-3).

    conv_bias_gauge (topology A, where BatchNorm directly follows a conv): that conv's bias has an analytically ZERO
    gradient (train-mode BatchNorm removes it in the forward AND the backward pass), so under Adam it performs an
    implementation-specific random walk of up to ~lr per step driven by rounding noise -- in Keras as much as here -- and
    moving_mean, an average of batch means that CONTAIN the bias, carries that walk with a lag.  Those moving_mean tensors
    are therefore not comparable between any two implementations beyond the walk's reach; they are checked against that
    reach only (0.01 momentum weight x sum over the epoch's steps of 2 x 7.3 lr t) and, functionally, through the inference pass
    they feed (validation loss / predictions, gated tightly by the caller).  moving_var is shift-invariant and every other
    moving_mean (topology B: BatchNorm after the ReLU) is compared in full."""
    off, worst, bias = 0, 0.0, None
    for name, shape, role in tensors:
        n = int(np.prod(shape))
        if role == "bias":
            bias = (off, n)
        if role in ("moving_mean", "moving_var"):
            d = float(np.abs(pa[off:off + n] - pb[off:off + n]).max())
            if role == "moving_mean" and conv_bias_gauge and bias is not None and bias[1] == n:
                # Adam's largest possible step: |m| / sqrt(v) <= sqrt((1 - beta1)^2 / (1 - beta2) * sum_i (beta1^2 / beta2)^i) = 7.3 (Cauchy-
                # Schwarz over the gradient history; a single gradient after zeros gives 3.16, which round 3 first used here and a
                # walk of 3.22 lr per step then exceeded by 1.5 %), times lr (the bias-corrected step size never exceeds lr)
                # What exceeds that reach is not the bias walk: it is compared like every other moving statistic (an epoch in which
                # the two fp32 trajectories part -- a tie flip moves many weights by 2 lr -- then goes to the twin band / the
                # step-by-step replay of the caller instead of failing here).
                reach = 0.01 * 2.0 * 7.3 * lr * steps * (steps + 1) / 2
                worst = max(worst, max(0.0, d - reach) / max(float(np.abs(pb[off:off + n]).max()), 1e-3))
            else:
                worst = max(worst, d / max(float(np.abs(pb[off:off + n]).max()), 1e-3))
        off += n
    return worst


def _decisive_differences(pred_a, pred_b, probs_b, margin=2e-3):
    """predictions that differ on clips where the reference's top-1 / top-2 probabilities are more than `margin` apart (an
    untrained net's softmax is near-uniform: its argmax is a coin toss for ANY two implementations)"""
    top = np.sort(probs_b, axis=1)
    decisive = (top[:, -1] - top[:, -2]) > margin
    return int(((np.asarray(pred_a) != np.asarray(pred_b)) & decisive).sum())


def _stepwise_epoch_check(gene, cfg, seed, epoch, state, Xtr, ytr, Xtr_d, ytr_d, tensors, v, tag):
    """One epoch replayed from `state` on a fresh GPU net with the fp32 oracle re-synchronised before EVERY optimiser step
    (the finest horizon there is: nothing can be amplified).  Per step: training loss within 1e-5 (relative, floor 1) and
    BatchNorm moving statistics within 1e-5 -- the forward pass has no discrete decisions that matter -- and per-tensor
    gradients within 1e-4 of the tensor's max on the steps without a tie flip.  A max-pool / ReLU tie that rounds the
    other way in the backward pass (seen on either fp32 side about once in five steps: profiles/
    r03_adam_first_step_tie_flips.txt) re-routes one gradient term in a late layer and perturbs EVERY upstream tensor by
    1e-4 ... a few 1e-2: such steps are allowed up to 1e-1, but at least half of the epoch's steps must be clean."""
    from cmoop_audio_processing_amd.session import epoch_permutation
    T, F = int(Xtr.shape[1]), int(Xtr.shape[2])
    perm = epoch_permutation(seed, epoch, len(Xtr)) if cfg.shuffle else np.arange(len(Xtr), dtype=np.int32)
    idx = torch.from_numpy(np.ascontiguousarray(perm)).cuda()
    with NetSession(gene, cfg, T, F, seed) as net:
        net.set_state(state)
        net.train_metrics()
        onet = ON.OracleNet(gene, ocfg(cfg), seed)
        n_steps = n_clean = 0
        for s0 in range(0, len(Xtr), cfg.batch):
            b = min(cfg.batch, len(Xtr) - s0)
            onet.set_state(net.get_state())
            net.train_step(Xtr_d, ytr_d, idx, row0=s0, B=b)
            rows = perm[s0:s0 + b]
            l_o, _ = onet.train_step(Xtr[rows], ytr[rows])
            l_g, _ = net.train_metrics()
            where = f"{tag} {gene} epoch {epoch} step {s0 // cfg.batch}"
            assert abs(l_g - l_o) <= 1e-5 * max(b, abs(l_o)), (where, l_g, l_o)
            d_stat = _bn_stat_deviation(tensors, net.get_params(), onet.get_flat(), conv_bias_gauge=(v == 0), steps=1, lr=cfg.lr)
            assert d_stat <= 1e-5, (where, d_stat)
            err = per_tensor_err(gene, v, cfg.classes, net.get_grads(), onet.grads_flat())
            worst = max(err.values())
            assert worst <= 1e-1, (where, max(err, key=err.get), worst)
            n_steps += 1
            n_clean += worst <= 1e-4
    assert 2 * n_clean >= n_steps, f"{tag} {gene} epoch {epoch}: only {n_clean} of {n_steps} steps free of tie flips"


def resynchronised_fit_check(gene, cfg, Xtr, ytr, Xva, yva, seed, expected=None, loss_tol=1e-4, pred_tol=1, stat_tol=5e-5, tag=""):
    """Tight end-to-end parity of the FULL early-stopped protocol without comparing two long chaotic fp32 trajectories at
    their ends: the GPU and the oracle are RE-SYNCHRONISED at every epoch boundary.

    1. The product's fit + read-outs run on a session net (cmoop_net_fit = the body of cmoop_eval_population's per-candidate
       work).  Its control flow is checked exactly: Keras' EarlyStopping fed the GPU's own validation-loss history must stop
       at the same epoch and pick the same best epoch; the accuracy read-out is the history's last entry ('last') or comes
       from the final weights ('evaluate'); `expected` (accuracy, fpr, epochs_run from PopulationEvaluator) must be
       reproduced bit for bit.
    2. The oracle loads the GPU's FINAL state (best weights restored when the script asks for it) and recomputes the
       read-outs: decisive predictions differ in at most pred_tol validation clips, accuracy / FPR accordingly.
    3. A second GPU net replays the run epoch by epoch (cmoop_net_run_epoch, the trainer's own device-state path); BEFORE
       each epoch the oracle loads the GPU's full state (weights, BatchNorm moving statistics, Adam m / v, iteration and
       dropout counters), both run the epoch on the same permutation and masks, and after it: the GPU's validation loss
       equals the history of run 1 BIT FOR BIT (it is the same trajectory), |val loss GPU - oracle| <= loss_tol (relative,
       floor 1), decisive predictions differ in <= pred_tol clips, BatchNorm moving statistics agree to stat_tol
       (epoch 0, which starts from Adam's zero state: 2e-3 / 5e-2, see the comment at the gate).
       Where an epoch misses one of these TIGHT gates, the oracle's own reproducibility over that very epoch is measured --
       the same epoch from the same state with torch's other CPU conv algorithm and in float64 -- and the GPU must be within
       5x of it, or else (the twins' deviations are heavy-tailed too) that epoch is replayed with a re-synchronisation
       before EVERY step and gated per step (_stepwise_epoch_check).  That happens where training is locally unstable for every implementation: the first epoch (Adam turns
       rounding-noise gradients into +-lr steps at t = 1), the take-off phase of a many-class run, and conv biases in front
       of a BatchNorm (zero true gradient: a random walk whose lag shows in moving_mean).
       The horizon of every comparison is one epoch, so rounding differences cannot be amplified into different runs, yet
       every epoch of the protocol as the product runs it is checked.  Returns the fit record + 'band_epochs'."""
    import torch as _t
    T, F = int(Xtr.shape[1]), int(Xtr.shape[2])
    Xtr_d, ytr_d = _t.from_numpy(Xtr).cuda(), _t.from_numpy(ytr.astype(np.int32)).cuda()
    Xva_d, yva_d = _t.from_numpy(Xva).cuda(), _t.from_numpy(yva.astype(np.int32)).cuda()
    v = G.VARIANT_NAMES[cfg.variant]
    oc = ocfg(cfg)

    def probs(onet):
        with _t.no_grad():
            return onet.forward(_t.from_numpy(np.ascontiguousarray(Xva, np.float32)).to(onet.dtype), False).double().numpy()
    with NetSession(gene, cfg, T, F, seed) as net:
        fit = net.fit(Xtr_d, ytr_d, Xva_d, yva_d)
        final = net.get_state()
        _, _, p_final = net.evaluate(Xva_d, yva_d)
    hist_l, hist_a, E = fit["val_loss_history"], fit["val_accuracy_history"], fit["epochs_run"]
    assert len(hist_l) == E >= 1 and np.isfinite(hist_l).all()
    stop, best = _es_replay(cfg, hist_l)
    assert (stop, best) == (E, fit["best_epoch"]), f"early stopping differs from Keras' on the same history: {(stop, best)} vs {(E, fit['best_epoch'])}"
    if expected is not None:
        assert (fit["acc"], fit["fpr"], E) == tuple(expected), (fit["acc"], fit["fpr"], E, expected)
    onet = ON.OracleNet(gene, oc, seed)
    onet.set_state(final)
    l_o, a_o, p_o = onet.evaluate(Xva, yva)
    fpr_o = OM.calculate_fpr(np.asarray(yva).ravel(), p_o, cfg.classes, oc.fpr_variant)
    restored = cfg.early_stop and cfg.restore_best and best != E - 1
    if cfg.acc_readout == "last":
        assert fit["acc"] == hist_a[-1]
    assert _decisive_differences(p_final.cpu().numpy(), p_o, probs(onet)) <= pred_tol
    n_diff = int((p_final.cpu().numpy() != p_o).sum())
    if n_diff == 0:          # identical predictions from identical weights: the read-outs must then be EQUAL
        assert abs(fit["fpr"] - fpr_o) <= 1e-12 and (cfg.acc_readout == "last" or abs(fit["acc"] - a_o) <= 1e-12), (fit, a_o, fpr_o)
    worst = dict(loss=0.0, preds=0, stats=0.0)
    band_epochs = []
    tensors = G.param_tensors(gene, v, cfg.classes)
    with NetSession(gene, cfg, T, F, seed) as net:
        onet = ON.OracleNet(gene, oc, seed)
        for e in range(E):
            st = net.get_state()
            onet.set_state(st)
            net.run_epoch(Xtr_d, ytr_d, e)
            ON.run_epoch(onet, Xtr, ytr, e)
            l_g, a_g, p_g = net.evaluate(Xva_d, yva_d)
            l_o, a_o, p_o = onet.evaluate(Xva, yva)
            assert l_g == hist_l[e] and a_g == hist_a[e], f"epoch {e}: the replayed GPU run left the product's trajectory ({l_g} vs {hist_l[e]})"
            d_loss = abs(l_g - l_o) / max(1.0, abs(l_o))
            d_pred = _decisive_differences(p_g.cpu().numpy(), p_o, probs(onet))
            d_stat = _bn_stat_deviation(tensors, net.get_params(), onet.get_flat(), conv_bias_gauge=(v == 0), steps=-(-len(Xtr) // cfg.batch), lr=cfg.lr)
            # epoch 0 starts from Adam's zero state: at t = 1 the update is lr * sign(g) for EVERY weight, so one max-pool /
            # ReLU tie that rounds differently (it happens about once in five steps on either side: step-by-step record
            # profiles/r03_adam_first_step_tie_flips.txt) flips the sign of every sub-noise gradient and moves those
            # weights by 2 lr; from epoch 1 on Adam's moments damp that
            l_tol, s_tol = (max(loss_tol, 2e-3), max(stat_tol, 5e-2)) if e == 0 else (loss_tol, stat_tol)
            tight = d_loss <= l_tol and d_pred <= pred_tol and d_stat <= s_tol
            if not tight:
                # the oracle's own reproducibility over THIS epoch from THIS state: other conv algorithm, and float64
                twins = []
                with _t.backends.mkldnn.flags(enabled=False):
                    tw = ON.OracleNet(gene, oc, seed)
                    tw.set_state(st)
                    ON.run_epoch(tw, Xtr, ytr, e)
                    twins.append((tw,) + tuple(tw.evaluate(Xva, yva)))
                if oc.compute != "bf16":            # (the bf16 restatement rounds through float32 tensors: no float64 form)
                    tw = ON.OracleNet(gene, oc, seed, dtype=_t.float64)
                    tw.set_state(st)
                    ON.run_epoch(tw, Xtr, ytr, e)
                    twins.append((tw,) + tuple(tw.evaluate(Xva, yva)))
                s_loss = max(abs(l_t - l_o) / max(1.0, abs(l_o)) for _, l_t, _, _ in twins)
                s_pred = max(_decisive_differences(p_t, p_o, probs(onet)) for _, _, _, p_t in twins)
                s_stat = max(_bn_stat_deviation(tensors, t_.get_flat().astype(np.float32), onet.get_flat(), conv_bias_gauge=(v == 0),
                                                 steps=-(-len(Xtr) // cfg.batch), lr=cfg.lr) for t_, _, _, _ in twins)
                band_epochs.append((e, f"loss {d_loss:.1e}/{s_loss:.1e} preds {d_pred}/{s_pred} stats {d_stat:.1e}/{s_stat:.1e}"))
                in_band = d_loss <= max(l_tol, 5.0 * s_loss) and d_pred <= max(pred_tol, 2 * s_pred + 1) and d_stat <= max(s_tol, 5.0 * s_stat)
                if not in_band:
                    # the twins' deviations are heavy-tailed too (a tie flipped early in the epoch or not): settle it at the
                    # finest grain -- replay THIS epoch with a re-synchronisation before EVERY STEP
                    _stepwise_epoch_check(gene, cfg, seed, e, st, Xtr, ytr, Xtr_d, ytr_d, tensors, v, tag)
                    band_epochs[-1] = (e, band_epochs[-1][1] + " -> verified step by step")
            else:
                worst = dict(loss=max(worst["loss"], d_loss), preds=max(worst["preds"], d_pred), stats=max(worst["stats"], d_stat))
    print(f"{tag} {gene} seed {seed}: {E} epochs (best {best}, restored {restored}) acc {fit['acc']:.4f} fpr {fit['fpr']:.4f}; "
          f"{E - len(band_epochs)} epochs at the tight gates (worst val loss {worst['loss']:.1e}, decisive predictions {worst['preds']}, moving stats "
          f"{worst['stats']:.1e}); {len(band_epochs)} epochs gated by the oracle's own one-epoch reproducibility: {band_epochs[:4]}")
    fit["band_epochs"] = [e for e, _ in band_epochs]
    return fit


RESYNC_PROTOCOLS = [
    ("sa_nsga_penalty", (16, 3, 1, 1, 2, 1)),    # the BatchNorm + dropout candidate whose end-to-end run is chaotic (B, restore_best, evaluate(), V1)
    ("mobo_penalty", (16, 3, 1, 1, 2, 1)),       # A, restore_best (mobo_penalty.py:227) + LAST-epoch accuracy (:236) + FPR of the restored weights (:239): quirk Q6
    ("nsga_penalty", (32, 5, 1, 2, 3, 1)),       # A, BatchNorm + dropout, no restore, y_true quirk
    ("sa_nsga_local", (16, 5, 0, 2, 1, 0)),      # B, V3
    ("init_sa_nsga_local", (64, 3, 1, 3, 4, 1)), # B, 64 filters, R = 3: the K = 512 head
]


@pytest.mark.parametrize("preset,gene", RESYNC_PROTOCOLS)
def test_full_protocol_resynchronised_every_epoch(preset, gene):
    """The full early-stopped protocol of each reference script (25 epochs, patience 2, ~100-150 optimiser steps) gated
    TIGHTLY at every epoch -- val loss 1e-4, <= 1 decisive prediction, moving statistics 5e-5, early-stopping decisions
    exact -- by re-synchronising the oracle with the GPU at each epoch boundary (resynchronised_fit_check); an epoch that
    misses a tight gate must lie within 5x of the oracle's own one-epoch reproducibility from the same state, and at most
    a third of a run's epochs may need that.  This replaces round 2's end-of-run statistical band for the BatchNorm +
    dropout candidate (three seeds here, as there)."""
    classes = 11 if preset == "sa_nsga_penalty" else 10
    Xtr, ytr, Xva, yva = make_split(192, 128, 21, 12, classes, 21, noise=0.3, label_noise=0.25)
    for seed in ((11, 12, 13) if preset == "sa_nsga_penalty" else (11,)):
        cfg = EvalConfig.preset(preset, classes=classes, epochs=25, patience=2, batch=32, eval_batch=64, seed=seed, n_slots=1)
        ev = PopulationEvaluator(Xtr, ytr, Xva, yva, cfg)
        acc, size_mb, fpr = ev.evaluate_individual(G.gene_to_hparams(gene))
        assert size_mb == G.model_size_mb(gene, G.VARIANT_NAMES[cfg.variant], classes)
        fit = resynchronised_fit_check(gene, cfg, Xtr, ytr, Xva, yva, seed, expected=(acc, fpr, ev.last_epochs_run[0]), tag=preset)
        # the band is the exception, not the rule: most epochs of a run meet the tight gates outright
        assert len(fit["band_epochs"]) <= max(2, fit["epochs_run"] // 3), fit["band_epochs"]
        if preset == "mobo_penalty" and fit["best_epoch"] != fit["epochs_run"] - 1:
            # Q6: accuracy is the LAST epoch's, FPR belongs to the RESTORED weights
            assert fit["acc"] == fit["val_accuracy_history"][-1]


def test_reference_input_shapes_at_the_boundary():
    """The hand-over the reference actually does: features [N,T,F,1] (channel axis added by prepare_dataset,
    nsga_penalty.py:151-153) and labels (N,1) (load_data, :74-76), as float64 / int64 numpy arrays.  Same result as
    the squeezed float32 / int32 inputs, bit for bit."""
    classes = 10
    cfg = EvalConfig.preset("nsga_penalty", epochs=2, batch=32, eval_batch=64, seed=3, n_slots=1, early_stop=False)
    Xtr, ytr, Xva, yva = make_split(96, 64, 21, 12, classes, 5)
    hp = G.gene_to_hparams((16, 3, 1, 1, 1, 0))
    a = PopulationEvaluator(Xtr, ytr, Xva, yva, cfg).compute_objectives_and_constraints([hp])[0]
    b = PopulationEvaluator(Xtr[..., None].astype(np.float64), ytr.reshape(-1, 1).astype(np.int64),
                            Xva[..., None].astype(np.float64), yva.reshape(-1, 1).astype(np.int64),
                            cfg).compute_objectives_and_constraints([hp])[0]
    assert a["objs"] == b["objs"] and a["CV"] == b["CV"]
    with pytest.raises(ValueError):
        PopulationEvaluator(Xtr[:, :, :, None, None], ytr, Xva, yva, cfg)
    with pytest.raises(ValueError):
        PopulationEvaluator(Xtr, ytr[:-1], Xva, yva, cfg)


@pytest.mark.parametrize("B", [28, 40, 51])
def test_wgrad_workspace_partial_batches_advice_r1(B):
    """gene (32,3,*,1,*,*) topology A at 101x40, batch 64 -- the block's second conv (51x20, 64->64 k3) asks for 98
    wgrad slices at the full batch but 109 for B in 28..51, which overflowed a workspace sized from the full batch
    (corrupting gradients or faulting).  One train step at such a B, then a full-batch step on the same net.
    At this size (8 M activations per layer) a handful of ReLU / max-pool decisions sit within fp32 rounding of a tie,
    so the fp32 oracle itself is 1e-4..1e-3 away from the same net evaluated in float64 (measured: tools/debug/
    grad_vs_fp64.py); the gate is therefore against the float64 oracle: HIP no further from it than 5e-3 or 5x the
    fp32 oracle's own distance.  A corrupted slab is an O(1) error."""
    gene, classes, seed = (32, 3, 1, 1, 1, 0), 10, 9
    cfg = EvalConfig(variant="A", classes=classes, batch=64, eval_batch=64)
    X, y = make_data(64, 101, 40, classes, 77)
    Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
    o32, o64 = ON.OracleNet(gene, ocfg(cfg), seed), ON.OracleNet(gene, ocfg(cfg), seed, dtype=torch.float64)
    with NetSession(gene, cfg, 101, 40, seed) as net:
        for step, b in enumerate((B, 64)):
            net.train_step(Xd, yd, None, row0=0, B=b)
            o32.train_step(X[:b], y[:b])
            o64.train_step(X[:b], y[:b])
            if step == 0:
                e_hip = per_tensor_err(gene, 0, classes, net.get_grads(), o64.grads_flat())
                e_o32 = per_tensor_err(gene, 0, classes, o32.grads_flat(), o64.grads_flat())
                worst = max(e_hip, key=e_hip.get)
                print(f"B={b}: worst HIP-vs-fp64 {worst} {e_hip[worst]:.2e} (fp32 oracle vs fp64 there: {e_o32[worst]:.2e})")
                # A flipped pool / ReLU decision moves a tensor of this BatchNorm gene by up to a few 1e-2 of its max on a few
                # elements (either fp32 side draws them independently: tests/test_gpu_production_shapes.py, DESIGN section 2;
                # observed here: 1.1e-2 on res0_conv1/kernel with the oracle at 2.0e-3) -- so the max-abs floor is the BatchNorm
                # one (5e-2) and the corruption check is the relative L2 error per tensor: a slab written past the workspace or
                # summed from stale memory is broad, >= 1e-1 and usually O(1).  The layer's kernels at exactly this batch are
                # compared at 5e-5 in tests/_production_shapes.py (51, 51, 20, ...).
                for name in e_hip:
                    assert e_hip[name] <= max(5e-2, 5.0 * e_o32[name]), (name, e_hip[name], e_o32[name])
                g_hip, g_64, g_32 = net.get_grads().astype(np.float64), o64.grads_flat().astype(np.float64), o32.grads_flat().astype(np.float64)
                off = 0
                for (name, shape, role) in G.param_tensors(gene, 0, classes):
                    n = int(np.prod(shape))
                    if role == "kernel":
                        ref = max(np.linalg.norm(g_64[off:off + n]), 1e-30)
                        l_hip, l_o32 = np.linalg.norm(g_hip[off:off + n] - g_64[off:off + n]) / ref, np.linalg.norm(g_32[off:off + n] - g_64[off:off + n]) / ref
                        assert l_hip <= max(1e-2, 5.0 * l_o32), (name, l_hip, l_o32)
                    off += n
        lg, _ = net.train_metrics()          # summed loss of both steps: the second step ran on sane weights
        assert np.isfinite(lg)
        l_o, a_o, _ = o32.evaluate(X, y)
        l_g, a_g, _ = net.evaluate(Xd, yd)
        assert abs(l_g - l_o) < 1e-3 * max(1.0, abs(l_o)), (l_g, l_o)

def test_optimiser_launch_sums_the_weight_gradient_slabs_bit_exactly(monkeypatch):
    """r2: the per-layer reduce_slices launches are folded into ONE optimiser launch (slab segments summed in the same fixed
    order, then Adam).  Four steps (full, partial, full, full batch -- the slice counts change with B) of a BatchNorm +
    dropout candidate must leave bit-identical parameters and gradients under the round-1 launch sequence
    (CMOOP_ADAM_UNFUSED, read per step).  Adam's arithmetic is pinned (no FMA contraction): left to the compiler the two
    kernels rounded m differently and this comparison failed by an ulp at step 4."""
    gene, classes, seed = (16, 3, 1, 2, 2, 1), 11, 11
    cfg = EvalConfig.preset("sa_nsga_penalty", classes=classes, batch=32, eval_batch=64, seed=seed, n_slots=1)
    X, y = make_data(128, 41, 20, classes, 5)
    Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
    got = {}
    for mode in ("fused", "unfused"):
        if mode == "unfused":
            monkeypatch.setenv("CMOOP_ADAM_UNFUSED", "1")
        else:
            monkeypatch.delenv("CMOOP_ADAM_UNFUSED", raising=False)
        with NetSession(gene, cfg, 41, 20, seed) as net:
            for step, b in enumerate((32, 19, 32, 32)):
                net.train_step(Xd, yd, None, row0=32 * step, B=b)
            got[mode] = (np.array(net.get_params()), np.array(net.get_grads()))
    monkeypatch.delenv("CMOOP_ADAM_UNFUSED", raising=False)
    for a, b in zip(got["fused"], got["unfused"]):
        assert np.isfinite(a).all() and np.array_equal(a.view(np.uint32), b.view(np.uint32))



def test_merged_dense_backward_launch_equals_the_two_launches_bit_for_bit(monkeypatch):
    """r3: the weight gradient and the data gradient of a hidden dense layer share ONE launch (dense_bwd_kernel: the same two
    bodies on disjoint workgroup ranges).  Four steps (full, partial, full, full) of a four-layer MLP head with dropout must
    leave bit-identical parameters and gradients under the round-2 launch pair (CMOOP_DENSE_UNFUSED, read per step)."""
    gene, classes, seed = (16, 3, 0, 1, 4, 1), 11, 13
    cfg = EvalConfig.preset("sa_nsga_penalty", classes=classes, batch=32, eval_batch=64, seed=seed, n_slots=1)
    X, y = make_data(128, 41, 20, classes, 6)
    Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
    got = {}
    for mode in ("merged", "pair"):
        if mode == "pair":
            monkeypatch.setenv("CMOOP_DENSE_UNFUSED", "1")
        else:
            monkeypatch.delenv("CMOOP_DENSE_UNFUSED", raising=False)
        with NetSession(gene, cfg, 41, 20, seed) as net:
            for step, b in enumerate((32, 19, 32, 32)):
                net.train_step(Xd, yd, None, row0=32 * step, B=b)
            got[mode] = (np.array(net.get_params()), np.array(net.get_grads()))
    monkeypatch.delenv("CMOOP_DENSE_UNFUSED", raising=False)
    for a, b in zip(got["merged"], got["pair"]):
        assert np.isfinite(a).all() and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_population_40_at_baseline_feature_size_config1():
    """BASELINE configs[1] at full population and feature size: the 40 genes of random.Random(0) (the bench's
    population), 101x40 features, a few hundred clips, one epoch.  Size-independent properties for all 40 --
    size_mb bit-exact, results independent of the number of candidates in flight, the FPR-quirk bound -- plus oracle
    parity (north-star gate 1e-3) on the three cheapest genes."""
    import random
    classes = 10
    rng = random.Random(0)
    pop = [G.random_hparams(rng) for _ in range(40)]
    genes = [G.normalize_hparams(hp) for hp in pop]
    Xtr, ytr, Xva, yva = make_split(256, 128, 101, 40, classes, 123, noise=0.3, label_noise=0.2)
    base = dict(epochs=2, batch=64, eval_batch=128, seed=0, early_stop=False)
    ev8 = PopulationEvaluator(Xtr, ytr, Xva, yva, EvalConfig.preset("nsga_penalty", n_slots=8, **base))
    res8 = ev8.compute_objectives_and_constraints(pop)
    ev3 = PopulationEvaluator(Xtr, ytr, Xva, yva, EvalConfig.preset("nsga_penalty", n_slots=3, **base))
    res3 = ev3.compute_objectives_and_constraints(pop)
    assert [r["objs"] for r in res8] == [r["objs"] for r in res3]          # deterministic, slot-count independent
    for g, r in zip(genes, res8):
        acc, size, fpr = -r["objs"][0], r["objs"][1], r["objs"][2]
        assert size == G.model_size_mb(g, 0, classes)                      # bit-exact (== in float64)
        assert 0.0 <= acc <= 1.0 and 0.0 <= fpr <= 1.0 / classes + 1e-12   # nsga_penalty.py:387 quirk: FPR <= 1/C
        assert r["CV"] == max(0.0, 0.9 - acc) + max(0.0, size - 2.5) + max(0.0, fpr - 0.1)
    assert ev8.last_epochs_run == [2] * 40
    cheapest = sorted(range(40), key=lambda i: G.fwd_flops_per_sample(genes[i], 0, classes, 101, 40))[:3]
    for i in cheapest:
        (a_acc, a_size, a_fpr, _), (b_acc, _, b_fpr, _) = oracle_pair(genes[i], ev8.config, Xtr, ytr, Xva, yva, i)   # seed = cfg.seed + index
        acc, size, fpr = -res8[i]["objs"][0], res8[i]["objs"][1], res8[i]["objs"][2]
        print(genes[i], "gpu", (acc, fpr), "oracle", (a_acc, a_fpr), "oracle, native conv", (b_acc, b_fpr))
        assert size == a_size and gate(acc, a_acc, b_acc) and gate(fpr, a_fpr, b_fpr)


def _oracle_population_evaluator(cfg, Xtr, ytr, Xva, yva, dtype=torch.float32):
    """compute_objectives_and_constraints on the oracle with the evaluator's seed convention (cfg.seed + running index)."""
    counter = {"n": 0}

    def oracle_eval(pop):
        out = []
        for hp in pop:
            g = G.normalize_hparams(hp)
            acc, size, fpr, _ = ON.evaluate_individual(g, ocfg(cfg), Xtr, ytr, Xva, yva, seed=cfg.seed + counter["n"], dtype=dtype)
            counter["n"] += 1
            out.append(OM.assemble(hp, acc, size, fpr, cfg.min_accuracy, cfg.max_model_size, cfg.max_fpr))
        return out
    return oracle_eval


def _recording(ev):
    """compute_objectives_and_constraints of a PopulationEvaluator, recording per call
    [(gene, seed, accuracy, size_mb, fpr, epochs_run)] -- what a replay of the same candidates needs."""
    calls = []

    def wrapped(pop):
        first = ev.evals_done
        out = ev.compute_objectives_and_constraints(pop)
        calls.append([(G.normalize_hparams(r["hparams"]), ev.config.seed + first + i, -r["objs"][0], r["objs"][1], r["objs"][2],
                       ev.last_epochs_run[i]) for i, r in enumerate(out)])
        return out
    return wrapped, calls


def _replay_every_candidate(calls, cfg, Xtr, ytr, Xva, yva, tag, **tol):
    """The GPU drove the search; every candidate it truly evaluated is now checked epoch by epoch against the oracle
    (same gene, same seed; resynchronised_fit_check), its recorded objectives reproduced bit for bit."""
    v = G.VARIANT_NAMES[cfg.variant]
    n = 0
    for call in calls:
        for gene, seed, acc, size, fpr, epochs in call:
            assert size == G.model_size_mb(gene, v, cfg.classes)                       # bit-exact (== in float64)
            resynchronised_fit_check(gene, cfg, Xtr, ytr, Xva, yva, seed, expected=(acc, fpr, epochs), tag=tag, **tol)
            n += 1
    return n


def test_sa_nsga2_35_classes_gpu_search_replayed_on_the_oracle_config2():
    """BASELINE configs[2] at reduced size: the surrogate-assisted loop of sa_nsga_penalty.py:522-637 (topology B,
    restore_best + evaluate(), infill 0.2, Kriging surrogate on the host) driven by the GPU evaluator on a 35-class task,
    pop 8 / gen 2 -> 8 + 2 * 1 true evaluations.  Round 2 ran a second, independent search on the oracle and compared
    hypervolumes after the two had diverged (one flipped prediction re-routes the Kriging infill): a 0.5-2.0x gate.  Now the
    GPU drives and the oracle REPLAYS: each of the 10 candidates the search evaluated is checked at every epoch of its
    early-stopped run (val loss 1e-4, <= 1 decisive prediction, moving statistics 5e-5, early-stopping decisions exact), and its
    objectives as the search saw them are reproduced bit for bit -- so the search consumed correct numbers, whatever
    route it took.  (The host loop itself is pinned by tests/golden/surrogate_golden.json.)"""
    from cmoop_audio_processing_amd import nsga, surrogate as S
    classes = 35
    cfg = EvalConfig.preset("sa_nsga_penalty", classes=classes, epochs=6, patience=2, batch=32, eval_batch=64, seed=5, n_slots=4)
    Xtr, ytr, Xva, yva = make_split(420, 140, 21, 12, classes, 61, noise=0.3, label_noise=0.1)
    ev = PopulationEvaluator(Xtr, ytr, Xva, yva, cfg)
    f_gpu, calls = _recording(ev)
    _, hist, n_true = S.sa_nsga2(f_gpu, 8, 2, infill_percent=0.2, seed=3)
    assert n_true == 8 + 2 * 1 == ev.evals_done and [len(c) for c in calls] == [8, 1, 1]
    fronts = [[[-r["Accuracy"], r["Size_MB"], r["FPR"]] for r in h] for h in hist]
    ref = nsga.shared_reference_point(fronts)
    assert all(nsga.hypervolume(f, ref) > 0 for f in fronts)
    assert _replay_every_candidate(calls, cfg, Xtr, ytr, Xva, yva, "configs[2]") == 10


def test_memetic_sa_nsga2_bf16_gpu_search_replayed_on_the_oracle_config4():
    """BASELINE configs[4] at reduced size: the full memetic method of init_sa_nsga_local.py:388-470 (LHS init, Kriging
    surrogate, Lamarckian LCB local search, infill 0.334) with the opt-in bf16-train arithmetic (compute='bf16'), pop 8 /
    gen 2 on the GPU evaluator; every truly evaluated candidate replayed epoch by epoch on the bf16 oracle.  bf16 nets are
    not comparable at fp32 tolerances even over one epoch (the bf16 oracle differs from itself by 1e-2..1e-1 on
    ONE step's gradients when only torch's conv algorithm changes), so the per-epoch gates are: val loss 2e-2, <= 4 of 96
    predictions, moving statistics 2e-3; control flow and read-outs exact as everywhere."""
    from cmoop_audio_processing_amd import surrogate as S
    classes = 10
    cfg = EvalConfig.preset("init_sa_nsga_local", classes=classes, epochs=4, patience=2, batch=32, eval_batch=64, seed=8, n_slots=4,
                            compute="bf16")
    Xtr, ytr, Xva, yva = make_split(192, 96, 21, 12, classes, 71, noise=1.0)
    ev = PopulationEvaluator(Xtr, ytr, Xva, yva, cfg)
    f_gpu, calls = _recording(ev)
    _, hist, n_true = S.sa_nsga2(f_gpu, 8, 2, infill_percent=0.334, seed=4, init="lhs", local_search=True)
    assert n_true == 8 + 2 * 2 == ev.evals_done and len(hist) == 2 and all(len(h) == 8 for h in hist)
    for h in hist:
        for rec in h:
            assert 0.0 <= rec["Accuracy"] <= 1.0 and rec["Size_MB"] > 0
    assert _replay_every_candidate(calls, cfg, Xtr, ytr, Xva, yva, "configs[4] bf16", loss_tol=2e-2, pred_tol=4, stat_tol=2e-3) == 12


def test_population_schema_determinism_and_problem_shim():
    classes = 10
    cfg = EvalConfig.preset("nsga_penalty", epochs=2, batch=32, eval_batch=64, seed=5, n_slots=3, early_stop=False)
    Xtr, ytr, Xva, yva = make_split(96, 64, 21, 12, classes, 31)
    import random
    rng = random.Random(0)
    pop = [G.random_hparams(rng) for _ in range(5)]
    ev = PopulationEvaluator(Xtr, ytr, Xva, yva, cfg)
    res = ev.compute_objectives_and_constraints(pop)
    assert [set(r) for r in res] == [{"hparams", "objs", "CV"}] * 5
    for hp, r in zip(pop, res):
        assert r["hparams"] is hp                                   # reference, not a copy (nsga_penalty.py:438)
        assert r["objs"][1] == G.model_size_mb(G.normalize_hparams(hp), 0, classes)
        acc, size, fpr = -r["objs"][0], r["objs"][1], r["objs"][2]
        assert r["CV"] == max(0.0, 0.9 - acc) + max(0.0, size - 2.5) + max(0.0, fpr - 0.1)
        assert 0.0 <= acc <= 1.0 and 0.0 <= fpr <= 0.1 + 1e-12     # quirk: FPR <= 1/C
    assert ev.last_epochs_run == [2] * 5
    # bit-reproducible: same seeds, different slot count -> identical objective vectors
    ev2 = PopulationEvaluator(Xtr, ytr, Xva, yva, EvalConfig.preset("nsga_penalty", epochs=2, batch=32, eval_batch=64, seed=5, n_slots=1, early_stop=False))
    res2 = ev2.compute_objectives_and_constraints(pop)
    assert [r["objs"] for r in res] == [r["objs"] for r in res2]
    # empty population and pymoo-style shim
    assert ev.compute_objectives_and_constraints([]) == []
    from cmoop_audio_processing_amd import AudioNASProblem
    out = {}
    AudioNASProblem(ev2)._evaluate(np.array([[0.0, 0.0, 1.0, 0.0, 0.0, 1.0], [0.5, 1.0, 0.0, 0.5, 1.0, 0.0]]), out)
    assert out["F"].shape == (2, 3) and out["G"].shape == (2, 3)
    assert np.allclose(out["G"][:, 1], out["F"][:, 1] - 2.5)


def test_pull_queue_entry_point_matches_the_plain_population_call():
    """cmoop_eval_population_pull (the C ABI the cross-rank queue binds): the library's worker threads call back into
    Python for candidate indices.  Draining a local longest-first queue through it must give bit-identical results to
    cmoop_eval_population, train every candidate exactly once, and hand out the expensive candidates first; a callback
    that raises ends the workers and re-raises in the caller."""
    import itertools
    import random
    import threading
    classes = 10
    cfg = EvalConfig.preset("nsga_penalty", epochs=2, batch=32, eval_batch=64, seed=9, n_slots=3, early_stop=False)
    Xtr, ytr, Xva, yva = make_split(96, 64, 21, 12, classes, 77)
    rng = random.Random(4)
    genes = [G.normalize_hparams(G.random_hparams(rng)) for _ in range(7)]
    seeds = [cfg.seed + i for i in range(7)]
    ev = PopulationEvaluator(Xtr, ytr, Xva, yva, cfg)
    plain = ev.evaluate_genes(genes, seeds)
    costs = [G.fwd_flops_per_sample(g, 0, classes, 21, 12) for g in genes]
    order = sorted(range(7), key=lambda i: (-costs[i], i))
    ctr, lock, handed = itertools.count(), threading.Lock(), []

    def pull():
        with lock:
            j = next(ctr)
            if j < 7:
                handed.append(order[j])
        return order[j] if j < 7 else -1
    got = ev.evaluate_genes_pull(genes, seeds, pull)
    assert sorted(got) == list(range(7)) and handed == order
    for i in range(7):
        assert np.array_equal(got[i][:4], plain[i][:4]), (i, got[i], plain[i])      # acc, size, fpr, epochs: bit-identical

    def bad_pull():
        raise RuntimeError("queue broke")
    with pytest.raises(RuntimeError, match="queue broke"):
        ev.evaluate_genes_pull(genes, seeds, bad_pull)


def test_bad_inputs_fail_loudly():
    from cmoop_audio_processing_amd import _lib
    cfg = EvalConfig(epochs=1)
    Xtr, ytr = make_data(8, 21, 12, 10, 1)
    ev = PopulationEvaluator(Xtr, ytr, Xtr, ytr, cfg)
    with pytest.raises(ValueError):
        ev.evaluate_individual({"filters": 48, "kernel_size": 3, "use_bn": True, "residual_blocks": 1, "fc_layers": 1, "use_dropout": False})
    with pytest.raises(KeyError):
        ev.evaluate_individual({"filters": 16})
    with pytest.raises(_lib.CmoopError):
        NetSession((16, 3, 0, 1, 1, 0), EvalConfig(classes=1), 21, 12, 0)


def test_nsga2_pop4_gen2_on_gpu_config0():
    """BASELINE configs[0] plumbing: the constrained NSGA-II loop (host) driving the GPU evaluator,
    pop=4, gen=2 -> 4*(1+2) = 12 true evaluations; sharded_map path is the single-rank one here."""
    from cmoop_audio_processing_amd import nsga
    cfg = EvalConfig.preset("nsga_penalty", epochs=3, patience=1, batch=32, eval_batch=64, seed=1, n_slots=4)
    Xtr, ytr, Xva, yva = make_split(128, 64, 21, 12, 10, 41)
    ev = PopulationEvaluator(Xtr, ytr, Xva, yva, cfg)
    pareto, hist = nsga.nsga2(ev.compute_objectives_and_constraints, 4, 2, seed=0)
    assert ev.evals_done == 12 and len(hist) == 2 and all(len(h) == 4 for h in hist)
    for rec in hist[-1]:
        assert rec["Size_MB"] == G.model_size_mb(G.normalize_hparams(rec), 0, 10)
    fronts = [[ind["objs"] for ind in fake] for fake in ([{"objs": [-r["Accuracy"], r["Size_MB"], r["FPR"]]} for r in h] for h in hist)]
    ref = nsga.shared_reference_point(fronts)
    assert all(nsga.hypervolume(f, ref) > 0 for f in fronts)


def test_hypervolume_parity_gpu_vs_oracle_search():
    """North-star gate 'hypervolume within 1 % of reference at equal generation count', at a scale the
    CPU oracle finishes in about a minute: the same seeded NSGA-II search (pop 4, 2 generations, 3 epochs,
    early stopping on) driven once by the GPU evaluator and once by the oracle; one shared reference point."""
    from cmoop_audio_processing_amd import nsga
    cfg = EvalConfig.preset("nsga_penalty", epochs=3, patience=2, batch=32, eval_batch=64, seed=7, n_slots=4)
    Xtr, ytr, Xva, yva = make_split(96, 64, 21, 12, 10, 51, noise=0.7)
    ev = PopulationEvaluator(Xtr, ytr, Xva, yva, cfg)
    _, hist_gpu = nsga.nsga2(ev.compute_objectives_and_constraints, 4, 2, seed=2)

    counter = {"n": 0}

    def oracle_eval(pop):
        out = []
        for hp in pop:
            g = G.normalize_hparams(hp)
            acc, size, fpr, _ = ON.evaluate_individual(g, ocfg(cfg), Xtr, ytr, Xva, yva, seed=cfg.seed + counter["n"])
            counter["n"] += 1
            out.append(OM.assemble(hp, acc, size, fpr, cfg.min_accuracy, cfg.max_model_size, cfg.max_fpr))
        return out
    _, hist_cpu = nsga.nsga2(oracle_eval, 4, 2, seed=2)
    f_gpu = [[[-r["Accuracy"], r["Size_MB"], r["FPR"]] for r in h] for h in hist_gpu]
    f_cpu = [[[-r["Accuracy"], r["Size_MB"], r["FPR"]] for r in h] for h in hist_cpu]
    ref = nsga.shared_reference_point(f_gpu + f_cpu)
    for g in range(2):
        # same genes survive in both searches (size is a pure function of the genes)
        assert sorted(r["Size_MB"] for r in hist_gpu[g]) == sorted(r["Size_MB"] for r in hist_cpu[g])
        hv_g, hv_c = nsga.hypervolume(f_gpu[g], ref), nsga.hypervolume(f_cpu[g], ref)
        print(f"gen {g}: HV gpu {hv_g:.6f} oracle {hv_c:.6f}")
        assert abs(hv_g - hv_c) <= 0.01 * max(hv_c, 1e-12)


def test_hard_synthetic_set_search_replayed_epoch_by_epoch():
    """on the section-8d synthetic set every candidate scores accuracy 1.0, so 'HV within 1 %' could not fail.
    A search (pop 4, 2 generations, early stopping on, BatchNorm and dropout genes as drawn) on the HARD variant of the set
    -- low SNR, neighbouring classes share two of three partials (bench.synth_waveforms(hard=True)) -- through the real
    pipeline: HIP front end -> StandardScaler -> GPU evaluator.  Accuracies spread.

    Round 2 showed that end-of-run comparisons are a lottery on this task for ANY two implementations (the fp32 and float64
    oracles end 5-14 accuracy points apart on single candidates; a 25 % hypervolume gate was all that held).  Instead the
    GPU drives the search and each of its 12 true evaluations is replayed on the oracle with a re-synchronisation at every
    epoch boundary (resynchronised_fit_check): val loss within 1e-4, <= 1 decisive prediction, moving statistics 5e-5 per epoch,
    early-stopping decisions and read-outs exact, the search's objectives reproduced bit for bit."""
    import bench
    from cmoop_audio_processing_amd import frontend, nsga
    wav, y = bench.synth_waveforms(480, 10, 7, torch.device("cuda"), n_samples=4000, chunk=160, hard=True, hard_snr_db=-6.0)
    feats = frontend.log_mel(wav)                                   # [480, 26, 40]
    Xtr_d, Xva_d = feats[:320].contiguous(), feats[320:].contiguous()
    frontend.prepare_dataset(Xtr_d, Xva_d, None, mode="refit")
    Xtr, Xva = Xtr_d.cpu().numpy(), Xva_d.cpu().numpy()
    ytr, yva = y[:320].cpu().numpy(), y[320:].cpu().numpy()
    cfg = EvalConfig.preset("nsga_penalty", epochs=12, patience=2, batch=32, eval_batch=64, seed=3, n_slots=4, fpr_variant="v1")
    ev = PopulationEvaluator(Xtr_d, y[:320], Xva_d, y[320:], cfg)
    f_gpu, calls = _recording(ev)
    nsga.nsga2(f_gpu, 4, 2, seed=5)
    assert len(calls) == 3 and all(len(c) == 4 for c in calls)
    accs = sorted(r[2] for call in calls for r in call)
    print("hard set accuracies:", [round(a, 3) for a in accs])
    assert accs[-1] - accs[0] >= 0.15, "the hard set must spread the accuracies"
    assert _replay_every_candidate(calls, cfg, Xtr, ytr, Xva, yva, "hard set") == 12


def _cos(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))


def test_bf16_modes_selected_by_config_field():
    """EvalConfig(compute=...) (cmoop_config.gemm_mode, no environment variable), one training step.

    bf16x3 is fp32-accurate: same 5e-4 gate as the exact path (observed 2e-6 .. 4e-6).

    bf16 rounds the GEMM operands of every layer between the first conv and the classifier (forward, dgrad, wgrad;
    fp32 accumulation), which the oracle restates (OracleConfig.compute='bf16').  The KERNELS match that definition
    at the fp32 tolerance (tests/test_gpu_kernels.py under CMOOP_GEMM_MODE=bf16: products of bf16 values are exact
    in fp32).  A whole net cannot be gated that tightly: a 1e-7 difference in an fp32 sum flips bf16 roundings
    downstream, and the bf16 oracle differs FROM ITSELF by 1.6e-2 .. 1.4e-1 (same metric) when only torch's CPU conv
    algorithm changes (mkldnn vs native; measured).  So the gates are: loss within 1e-3, the full
    gradient vector within cos >= 0.995 of the bf16 oracle's, and closer to it than to the fp32 oracle's."""
    classes, seed, T, F, B = 10, 99, 21, 12, 24
    X, y = make_data(64, T, F, classes, 3)
    Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
    for gene in [(32, 5, 1, 2, 3, 1), (32, 5, 0, 2, 3, 1), (16, 3, 1, 1, 2, 1)]:
        o32 = ON.OracleNet(gene, ocfg(EvalConfig(variant="A", classes=classes, batch=32, compute="bf16x3")), seed)
        o32.train_step(X[8:8 + B], y[8:8 + B])
        for compute in ("bf16x3", "bf16"):
            cfg = EvalConfig(variant="A", classes=classes, batch=32, eval_batch=16, compute=compute)
            onet = ON.OracleNet(gene, ocfg(cfg), seed)
            assert onet.cfg.compute == ("bf16" if compute == "bf16" else "fp32")
            with NetSession(gene, cfg, T, F, seed) as net:
                net.train_step(Xd, yd, None, row0=8, B=B)
                lo, co = onet.train_step(X[8:8 + B], y[8:8 + B])
                lg, cg = net.train_metrics()
                g, go, g32 = net.get_grads(), onet.grads_flat(), o32.grads_flat()
                gerr = per_tensor_err(gene, 0, classes, g, go)
                print(f"{gene} compute={compute}: loss {lg:.6f} / {lo:.6f}; worst tensor err {max(gerr.values()):.2e}; "
                      f"cos own oracle {_cos(g, go):.6f}, cos fp32 oracle {_cos(g, g32):.6f}")
                if compute == "bf16x3":
                    assert abs(lg - lo) < 2e-5 * max(1.0, abs(lo)) and max(gerr.values()) < 5e-4, gerr
                else:
                    assert abs(lg - lo) < 1e-3 * max(1.0, abs(lo))
                    assert _cos(g, go) >= 0.995 and _cos(g, go) > _cos(g, g32)
                l_o, a_o, p_o = onet.evaluate(X, y)
                l_g, a_g, p_g = net.evaluate(Xd, yd)
                assert abs(l_g - l_o) < 2e-3 * max(1.0, abs(l_o))


@pytest.mark.skipif(ENV_MODE != "", reason="already inside a mode-forced child run")
def test_net_parity_under_forced_bf16x3():
    """Re-run the gradient / inference / protocol parity tests of this file in a child process with
    CMOOP_GEMM_MODE=bf16x3 (every MFMA GEMM of every net on the split-precision bodies): the exact path's gates."""
    env = dict(os.environ, CMOOP_GEMM_MODE="bf16x3")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k",
                        "test_init_step_grads_and_eval_parity or test_partial_batch or test_classes_35 or "
                        "test_evaluate_individual_protocol_parity"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.skipif(ENV_MODE != "", reason="already inside a mode-forced child run")
def test_rccl_executes_the_generation_exchange_on_a_one_rank_group():
    """'one RCCL all_gather of objective vectors per generation'.  No multi-GPU node has been in reach, so until
    round 3 RCCL had never executed a single call of this repo.  torch.distributed backend 'nccl' IS RCCL on ROCm and accepts
    a ONE-rank group on one GPU: with CMOOP_FORCE_COLLECTIVES=1 the evaluator takes its N-rank path there -- the candidates
    go through the shared c10d-store counter (dynamic schedule) or the LPT buckets (static), and the [n, width] float64
    objective matrix through `all_gather_into_tensor` on device tensors -- and must return what the plain path returns, bit
    for bit.  (Degenerate in the number of peers, real in every call: process-group creation with a device id, the store
    fetch-add, the collective on HBM-resident float64 tensors, the ownership check.)"""
    code = r'''
import os, sys, json
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["CMOOP_ROOT"]); sys.path.insert(0, os.path.join(os.environ["CMOOP_ROOT"], "tests"))
from cmoop_audio_processing_amd import EvalConfig, PopulationEvaluator, genes as G
from test_gpu_net import make_split
import random
torch.cuda.set_device(0)
Xtr, ytr, Xva, yva = make_split(96, 64, 21, 12, 10, 31)
rng = random.Random(0)
pop = [G.random_hparams(rng) for _ in range(5)]
base = dict(epochs=2, batch=32, eval_batch=64, seed=5, n_slots=3, early_stop=False)
plain = PopulationEvaluator(Xtr, ytr, Xva, yva, EvalConfig.preset("nsga_penalty", **base)).compute_objectives_and_constraints(pop)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % int(os.environ["CMOOP_PORT"]), rank=0, world_size=1,
                        device_id=torch.device("cuda", 0))
os.environ["CMOOP_FORCE_COLLECTIVES"] = "1"
out = {}
for schedule in ("dynamic", "static"):
    ev = PopulationEvaluator(Xtr, ytr, Xva, yva, EvalConfig.preset("nsga_penalty", schedule=schedule, **base))
    res = ev.compute_objectives_and_constraints(pop)
    assert [r["objs"] for r in res] == [r["objs"] for r in plain] and [r["CV"] for r in res] == [r["CV"] for r in plain], schedule
    assert ev.last_rank_of == [0] * 5
    out[schedule] = dict(ev.last_queue_stats)
    assert ev.compute_objectives_and_constraints([]) == []
assert out["dynamic"]["all_gather_ms"] > 0 and out["dynamic"]["store_adds"] >= 1, out
t = torch.ones(4, dtype=torch.float64, device="cuda"); dist.all_reduce(t); assert float(t.sum()) == 4.0
print("RCCL_ONE_RANK_OK backend=%s" % dist.get_backend(), json.dumps(out["dynamic"]))
dist.destroy_process_group()
'''
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CMOOP_ROOT=root, CMOOP_PORT=str(port), MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    print(r.stdout[-1500:])
    assert r.returncode == 0 and "RCCL_ONE_RANK_OK backend=nccl" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
