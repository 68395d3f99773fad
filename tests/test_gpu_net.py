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


PROTOCOLS = [
    ("nsga_penalty", (16, 3, 0, 1, 1, 0)),       # A, last-epoch accuracy, no restore, y_true quirk
    ("sa_nsga_local", (16, 5, 0, 2, 1, 0)),      # B, V3
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
    assert ep in (a[3], b[3])


def test_protocol_parity_batchnorm_dropout_short_horizon():
    """The BatchNorm + dropout candidate (restore_best, evaluate() read-out, FPR V1) over a horizon on which rounding
    differences have not yet been amplified: 5 epochs (30 optimiser steps), no early stop.  Strict gate."""
    preset, gene = CHAOTIC
    (acc, fpr, ep), a, b = protocol_case(preset, gene, 5, 5)
    assert ep == a[3] == b[3] == 5
    assert gate(acc, a[0], b[0]) and gate(fpr, a[2], b[2])


def test_protocol_parity_batchnorm_dropout_full_protocol_is_statistical():
    """The same candidate through the full early-stopped protocol (25 epochs, patience 2: ~140 steps).  Here training is
    CHAOTIC in the numerical sense: the oracle's own two CPU conv algorithms (same arithmetic, another fp32 summation
    order) end 4 of 128 predictions apart, and on the GPU a one-ulp change in Adam's rounding of m (FMA contraction chosen
    differently by the compiler in two kernels, since pinned) moved the result from 0.461 / 23 epochs to 0.508 / 25 epochs.
    A single run therefore cannot be gated at 1e-3 against anything.  What is pinned instead, over 3 seeds:
    every GPU result lies within the oracle's range widened by three times the oracle's mean own spread, the stopping
    epoch within the patience of an oracle run, and the seed-averaged accuracy / FPR within 0.04 / 0.01 of the oracle's."""
    preset, gene = CHAOTIC
    runs = [protocol_case(preset, gene, 25, 2, seed=s) for s in (11, 12, 13)]
    for k, name, mean_gate in ((0, "accuracy", 0.04), (1, "fpr", 0.01)):
        ok = 0 if k == 0 else 2                      # column of the oracle tuple
        spread = max(1.0 / 128 if k == 0 else 1e-3, float(np.mean([abs(a[ok] - b[ok]) for _, a, b in runs])))
        for (gpu, a, b) in runs:
            lo, hi = min(a[ok], b[ok]) - 3 * spread, max(a[ok], b[ok]) + 3 * spread
            assert lo <= gpu[k] <= hi, (name, gpu[k], a[ok], b[ok], spread)
        g_mean = float(np.mean([gpu[k] for gpu, _, _ in runs]))
        o_mean = float(np.mean([0.5 * (a[ok] + b[ok]) for _, a, b in runs]))
        print(f"mean {name} over seeds: gpu {g_mean:.4f} oracle {o_mean:.4f} (mean own spread {spread:.4f})")
        assert abs(g_mean - o_mean) <= mean_gate, (name, g_mean, o_mean)
    for (gpu, a, b) in runs:
        assert min(abs(gpu[2] - a[3]), abs(gpu[2] - b[3])) <= 2, (gpu[2], a[3], b[3])
        assert 0.3 <= a[0] <= 0.97



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
                for name in e_hip:
                    # 5e-3: a flipped pool / ReLU decision moves a tensor by ~1e-3 of its max (it happens on either side:
                    # HIP 1.2e-3 with the fp32 oracle at 6e-6, and the reverse, were both observed); a slab written
                    # past the workspace or summed from stale memory is >= 1e-2 and usually O(1)
                    assert e_hip[name] <= max(5e-3, 5.0 * e_o32[name]), (name, e_hip[name], e_o32[name])
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


def _recording(evaluate):
    calls = []

    def wrapped(pop):
        out = evaluate(pop)
        calls.append([(G.normalize_hparams(r["hparams"]), -r["objs"][0], r["objs"][1], r["objs"][2]) for r in out])
        return out
    return wrapped, calls


def test_sa_nsga2_35_classes_on_gpu_vs_oracle_config2():
    """BASELINE configs[2] at reduced size: the surrogate-assisted loop of sa_nsga_penalty.py:522-637 (topology B,
    restore_best + evaluate(), infill 0.2, Kriging surrogate on the host) driven by the GPU evaluator on a 35-class
    task, pop 8 / gen 2 -> 8 + 2*1 true evaluations, against the same seeded loop on the oracle.
    * the initial population (the same 8 genes by construction): size bit-exact per candidate; the population's MEAN
      accuracy and FPR inside the band of three equivalent oracle evaluations (fp32 mkldnn conv, fp32 native conv,
      float64).  Per-candidate gates are meaningless here: in the take-off phase of a 35-class run the oracle differs
      from ITSELF by up to 0.24 accuracy between its two conv algorithms (gene (64,3,0,3,1,1): 0.564 vs 0.321 after 6
      epochs, measured), so no implementation can reproduce single candidates; the 1e-3 gate proper is
      test_evaluate_individual_protocol_parity;
    * the loop: same number of true evaluations; final-generation hypervolumes of the same order on a shared
      reference point (one flipped validation prediction legitimately steers the Kriging infill choice to another gene)."""
    from cmoop_audio_processing_amd import nsga, surrogate as S
    classes = 35
    cfg = EvalConfig.preset("sa_nsga_penalty", classes=classes, epochs=6, patience=2, batch=32, eval_batch=64, seed=5, n_slots=4)
    Xtr, ytr, Xva, yva = make_split(420, 140, 21, 12, classes, 61, noise=0.3, label_noise=0.1)
    ev = PopulationEvaluator(Xtr, ytr, Xva, yva, cfg)
    f_gpu, calls_gpu = _recording(ev.compute_objectives_and_constraints)
    _, hist_gpu, n_gpu = S.sa_nsga2(f_gpu, 8, 2, infill_percent=0.2, seed=3)
    f_cpu, calls_cpu = _recording(_oracle_population_evaluator(cfg, Xtr, ytr, Xva, yva))
    _, hist_cpu, n_cpu = S.sa_nsga2(f_cpu, 8, 2, infill_percent=0.2, seed=3)
    assert n_gpu == n_cpu == 8 + 2 * 1 and ev.evals_done == n_gpu
    assert [c[0] for c in calls_gpu[0]] == [c[0] for c in calls_cpu[0]]
    bands = []
    for i, ((g_g, acc_g, size_g, fpr_g), (g_c, acc_c, size_c, fpr_c)) in enumerate(zip(calls_gpu[0], calls_cpu[0])):
        band = oracle_band(g_c, cfg, Xtr, ytr, Xva, yva, cfg.seed + i)
        assert band[0][0] == acc_c and band[0][2] == fpr_c           # the loop's oracle call is the band's first member
        print(g_g, "gpu", (acc_g, fpr_g), "oracle band acc", [b[0] for b in band], "fpr", [b[2] for b in band])
        assert size_g == size_c and 0.0 <= acc_g <= 1.0 and 0.0 <= fpr_g <= 1.0
        bands.append(band)
    for col, name in ((0, "accuracy"), (2, "FPR")):
        means = [float(np.mean([band[v][col] for band in bands])) for v in range(3)]
        gpu_mean = float(np.mean([c[1 if col == 0 else 3] for c in calls_gpu[0]]))
        print(f"population mean {name}: gpu {gpu_mean:.4f}, oracle variants {means}")
        assert gate(gpu_mean, *means, tol=0.02), (name, gpu_mean, means)
    fr_g = [[-r["Accuracy"], r["Size_MB"], r["FPR"]] for r in hist_gpu[-1]]
    fr_c = [[-r["Accuracy"], r["Size_MB"], r["FPR"]] for r in hist_cpu[-1]]
    ref = nsga.shared_reference_point([fr_g, fr_c])
    hv_g, hv_c = nsga.hypervolume(fr_g, ref), nsga.hypervolume(fr_c, ref)
    print(f"SA-NSGA-II 35 classes: HV gpu {hv_g:.6f} oracle {hv_c:.6f}")
    # the two searches evaluate different genes after the first infill (see above), so their fronts differ as two
    # runs of the oracle with different conv algorithms do; same order of magnitude is all that can be asked here --
    # the 1 % hypervolume gate at equal genes is test_hypervolume_parity_gpu_vs_oracle_search
    assert hv_g > 0 and hv_c > 0 and 0.5 <= hv_g / hv_c <= 2.0


def test_memetic_sa_nsga2_bf16_on_gpu_vs_oracle_config4():
    """BASELINE configs[4] at reduced size: the full memetic method of init_sa_nsga_local.py:388-470 (LHS init,
    Kriging surrogate, Lamarckian LCB local search, infill 0.334) with the opt-in bf16-train arithmetic
    (compute='bf16'), pop 8 / gen 2 on the GPU evaluator vs the same loop on the bf16 oracle.  bf16 nets are not
    bit-comparable (a bf16 oracle differs from itself by 1e-2..1e-1 on one step's gradients), so the gate
    is: same number of true evaluations, valid records, and hypervolume within 25 % -- the loop runs end to end on
    the GPU path with the bf16 kernels."""
    from cmoop_audio_processing_amd import nsga, surrogate as S
    classes = 10
    cfg = EvalConfig.preset("init_sa_nsga_local", classes=classes, epochs=4, patience=2, batch=32, eval_batch=64, seed=8, n_slots=4,
                            compute="bf16")
    Xtr, ytr, Xva, yva = make_split(192, 96, 21, 12, classes, 71, noise=1.0)
    ev = PopulationEvaluator(Xtr, ytr, Xva, yva, cfg)
    kw = dict(infill_percent=0.334, seed=4, init="lhs", local_search=True)
    _, hist_gpu, n_gpu = S.sa_nsga2(ev.compute_objectives_and_constraints, 8, 2, **kw)
    assert n_gpu == 8 + 2 * 2 == ev.evals_done and len(hist_gpu) == 2 and all(len(h) == 8 for h in hist_gpu)
    for h in hist_gpu:
        for rec in h:
            assert 0.0 <= rec["Accuracy"] <= 1.0 and rec["Size_MB"] > 0
    _, hist_cpu, n_cpu = S.sa_nsga2(_oracle_population_evaluator(cfg, Xtr, ytr, Xva, yva), 8, 2, **kw)
    assert n_cpu == n_gpu
    f_gpu = [[[-r["Accuracy"], r["Size_MB"], r["FPR"]] for r in h] for h in hist_gpu]
    f_cpu = [[[-r["Accuracy"], r["Size_MB"], r["FPR"]] for r in h] for h in hist_cpu]
    ref = nsga.shared_reference_point(f_gpu + f_cpu)
    hv_g, hv_c = nsga.hypervolume(f_gpu[-1], ref), nsga.hypervolume(f_cpu[-1], ref)
    print(f"memetic bf16: HV gpu {hv_g:.6f} oracle {hv_c:.6f}")
    assert abs(hv_g - hv_c) <= 0.25 * max(hv_c, 1e-12)


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


def test_hard_synthetic_set_gpu_deviates_no_more_than_the_oracle_does_from_itself():
    """on the §8d synthetic set every candidate scores accuracy 1.0, so 'HV within 1 %' could not fail.
    A search (pop 4, 2 generations, early stopping on) on the HARD variant of the set -- low SNR, neighbouring classes
    share two of three partials (bench.synth_waveforms(hard=True)) -- through the real pipeline: HIP front end ->
    StandardScaler -> GPU evaluator.  Accuracies spread from chance to 1.0.

    The GPU drives the search; every population it evaluated (initial population, offspring of generations 0 and 1: 12
    evaluations) is replayed, same genes and same seeds, through the oracle in fp32 AND in float64 (same float32 initial
    weights) -- two equivalent CPU evaluations of one algorithm whose rounding differs about as much as a GPU's does.
    What round 2 measured on this task, and why the gate below is statistical: training here is numerically UNSTABLE for
    every implementation.  The fp32 and float64 oracles end 5, 6 and 14 accuracy points apart on single candidates (0.869 vs
    0.919; 0.788 vs 0.844; 0.938 vs 0.800) and their hypervolumes of one evaluated set differ by 16 % (2.74 vs 3.19); the
    GPU lands ON the float64 oracle for one set (HV 4.459015 both), on the fp32 oracle for another and 23 % off both for the
    third (one candidate early-stops at 0.86 where both oracles reach 0.97).  Candidates near chance take off late, the
    validation set has 160 clips and patience is 2: one rounding difference moves the stopping epoch.  A 1 % hypervolume
    gate on such a task is a lottery for ANY pair of implementations (it stays asserted where training is reproducible:
    test_hypervolume_parity_gpu_vs_oracle_search, and the 1e-3 protocol gates).  Pinned here instead:
      * the same genes reach both sides, sizes bit-exact, accuracies spread (the set has teeth);
      * the GPU's mean |accuracy - fp32 oracle| and mean |FPR - fp32 oracle| over the 12 evaluations are no larger than
        twice the float64 oracle's own (floors 0.01 / 0.002): measured 0.022 vs 0.026 for accuracy;
      * summed hypervolume of the three evaluated sets within 25 % of the fp32 oracle's (measured -8.5 %; float64: +4.4 %).
    BatchNorm and dropout are off for every candidate (with them on, the oracle's own two conv algorithms gave HV 0.096
    and 0.144 on the same genes); the search itself runs only on the GPU side because independent searches diverge after
    one flipped tournament (measured: generation-1 HV 0.061 vs 0.079)."""
    import bench
    from cmoop_audio_processing_amd import frontend, nsga

    def reproducible(evaluate):
        def f(pop):
            res = evaluate([dict(hp, use_bn=False, use_dropout=False) for hp in pop])
            for r, hp in zip(res, pop):
                r["hparams"] = hp          # the search operators keep working on the caller's dicts
            return res
        return f
    wav, y = bench.synth_waveforms(480, 10, 7, torch.device("cuda"), n_samples=4000, chunk=160, hard=True, hard_snr_db=-6.0)
    feats = frontend.log_mel(wav)                                   # [480, 26, 40]
    Xtr_d, Xva_d = feats[:320].contiguous(), feats[320:].contiguous()
    frontend.prepare_dataset(Xtr_d, Xva_d, None, mode="refit")
    Xtr, Xva = Xtr_d.cpu().numpy(), Xva_d.cpu().numpy()
    ytr, yva = y[:320].cpu().numpy(), y[320:].cpu().numpy()
    cfg = EvalConfig.preset("nsga_penalty", epochs=12, patience=2, batch=32, eval_batch=64, seed=3, n_slots=4, fpr_variant="v1")
    ev = PopulationEvaluator(Xtr_d, y[:320], Xva_d, y[320:], cfg)
    f_gpu, calls_gpu = _recording(reproducible(ev.compute_objectives_and_constraints))
    nsga.nsga2(f_gpu, 4, 2, seed=5)
    assert len(calls_gpu) == 3 and all(len(c) == 4 for c in calls_gpu)
    pops = [[G.gene_to_hparams(g) for g, *_ in call] for call in calls_gpu]

    def replay(dtype):
        f, calls = _recording(reproducible(_oracle_population_evaluator(cfg, Xtr, ytr, Xva, yva, dtype=dtype)))
        for pop in pops:
            f([dict(hp) for hp in pop])
        return calls
    calls_32, calls_64 = replay(torch.float32), replay(torch.float64)
    flat = lambda calls: [r for call in calls for r in call]
    g, o32, o64 = flat(calls_gpu), flat(calls_32), flat(calls_64)
    assert [r[0] for r in g] == [r[0] for r in o32] and [r[2] for r in g] == [r[2] for r in o32]   # same genes, sizes bit-exact
    for k in range(3):
        print(f"set {k}: accuracy gpu / oracle fp32 / oracle fp64:",
              [(round(a[1], 4), round(b[1], 4), round(c[1], 4)) for a, b, c in zip(calls_gpu[k], calls_32[k], calls_64[k])])
    accs = sorted(r[1] for r in g)
    assert accs[-1] - accs[0] >= 0.15, "the hard set must spread the accuracies"
    for col, name, floor in ((1, "accuracy", 0.01), (3, "fpr", 0.002)):
        d_gpu = float(np.mean([abs(a[col] - b[col]) for a, b in zip(g, o32)]))
        d_o64 = float(np.mean([abs(a[col] - b[col]) for a, b in zip(o64, o32)]))
        print(f"mean |{name} - fp32 oracle| over 12 evaluations: gpu {d_gpu:.4f}, float64 oracle {d_o64:.4f}")
        assert d_gpu <= max(floor, 2.0 * d_o64), (name, d_gpu, d_o64)
    fronts = lambda calls: [[[-acc, size, fpr] for _, acc, size, fpr in call] for call in calls]
    f_g, f_32, f_64 = fronts(calls_gpu), fronts(calls_32), fronts(calls_64)
    ref = nsga.shared_reference_point(f_g + f_32 + f_64)
    hv = lambda f: [nsga.hypervolume(f[k], ref) for k in range(3)]
    hv_g, hv_32, hv_64 = hv(f_g), hv(f_32), hv(f_64)
    print("HV per evaluated set: gpu", [round(v, 4) for v in hv_g], "fp32 oracle", [round(v, 4) for v in hv_32],
          "float64 oracle", [round(v, 4) for v in hv_64])
    assert abs(sum(hv_g) - sum(hv_32)) <= 0.25 * sum(hv_32)


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
