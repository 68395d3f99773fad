"""Search space, gene codecs and closed-form architecture arithmetic (host only).

Everything here is integer / float64 host arithmetic -- no device code.  It
mirrors the reference's search-space constants and the quantities its evaluator
derives from a built Keras model without ever building one:

* search space .................. nsga_penalty.py:186-196, sa_nsga_penalty.py:106-111
* fc ladder ..................... nsga_penalty.py:311-316
* topology A ("deep") ........... nsga_penalty.py:225-334
* topology B ("shallow") ........ sa_nsga_penalty.py:137-177
* size in MB .................... nsga_penalty.py:337-344 (count_params()*4/1024**2;
                                  count_params includes BN moving mean/var)
* [0,1]^6 codec ................. mobo_penalty.py:305-338
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, List, Sequence, Tuple

FILTER_OPTIONS = [16, 32, 64]
KERNEL_SIZE_OPTIONS = [3, 5]
USE_BN_OPTIONS = [True, False]
RESIDUAL_BLOCK_OPTIONS = [1, 2, 3]
FC_LAYER_OPTIONS = [1, 2, 3, 4]
USE_DROPOUT_OPTIONS = [True, False]

#: key order of the reference's hparams dict (nsga_penalty.py:405-412)
GENE_KEYS = ("filters", "kernel_size", "use_bn", "residual_blocks", "fc_layers", "use_dropout")
GENE_OPTIONS = (FILTER_OPTIONS, KERNEL_SIZE_OPTIONS, USE_BN_OPTIONS,
                RESIDUAL_BLOCK_OPTIONS, FC_LAYER_OPTIONS, USE_DROPOUT_OPTIONS)

FC_LAYER_CONFIGS = {1: [64], 2: [128, 64], 3: [256, 128, 64], 4: [512, 256, 128, 64]}

VARIANT_A = 0  # "deep":    nsga_penalty.py / mobo_penalty.py / acc_*_nsga_1.py / psi_mobo_2.py
VARIANT_B = 1  # "shallow": sa_nsga_penalty.py and every sa_/psi_/init_ ablation script
VARIANT_NAMES = {"A": VARIANT_A, "B": VARIANT_B, 0: VARIANT_A, 1: VARIANT_B}

Gene = Tuple[int, int, int, int, int, int]


def normalize_hparams(hp: Dict) -> Gene:
    """hparams dict -> (filters, kernel, bn, res_blocks, fc_layers, dropout) ints.

    Accepts python/numpy ints and bools and 0/1 ints read back from spreadsheets
    (psi_sa_nsga_local.py:255-269 casts them the same way).
    """
    missing = [k for k in GENE_KEYS if k not in hp]
    if missing:
        raise KeyError(f"hparams is missing keys {missing}; expected {GENE_KEYS}")
    g = (int(hp["filters"]), int(hp["kernel_size"]), int(bool(hp["use_bn"])),
         int(hp["residual_blocks"]), int(hp["fc_layers"]), int(bool(hp["use_dropout"])))
    validate_gene(g)
    return g


def validate_gene(g: Sequence[int]) -> None:
    f, k, bn, r, fc, dr = g
    if f not in FILTER_OPTIONS or k not in KERNEL_SIZE_OPTIONS or bn not in (0, 1) \
            or r not in RESIDUAL_BLOCK_OPTIONS or fc not in FC_LAYER_OPTIONS or dr not in (0, 1):
        raise ValueError(f"gene {tuple(g)} is outside the search space")


def gene_to_hparams(g: Sequence[int]) -> Dict:
    f, k, bn, r, fc, dr = g
    return {"filters": int(f), "kernel_size": int(k), "use_bn": bool(bn),
            "residual_blocks": int(r), "fc_layers": int(fc), "use_dropout": bool(dr)}


def all_genes() -> List[Gene]:
    """The 288 genotypes in lexicographic option order."""
    out = []
    for f in FILTER_OPTIONS:
        for k in KERNEL_SIZE_OPTIONS:
            for bn in (1, 0):
                for r in RESIDUAL_BLOCK_OPTIONS:
                    for fc in FC_LAYER_OPTIONS:
                        for dr in (1, 0):
                            out.append((f, k, bn, r, fc, dr))
    return out


def random_hparams(rng) -> Dict:
    """One individual drawn like initialize_population (nsga_penalty.py:402-415):
    ``rng.choice`` per gene in the reference's key order (rng = random.Random)."""
    return {key: rng.choice(opts) for key, opts in zip(GENE_KEYS, GENE_OPTIONS)}


# --------------------------------------------------------------------------
# [0,1]^6 codec (mobo_penalty.py:305-338) -- used by the Problem._evaluate shim
# --------------------------------------------------------------------------
def hparams_to_vector(hp: Dict) -> List[float]:
    v = []
    for key, opts in zip(GENE_KEYS, GENE_OPTIONS):
        v.append(opts.index(hp[key]) / (len(opts) - 1))
    return v


def vector_to_hparams(vec: Iterable[float]) -> Dict:
    hp = {}
    for x, key, opts in zip(vec, GENE_KEYS, GENE_OPTIONS):
        # python round() = banker's rounding, exactly what the reference calls
        hp[key] = opts[int(round(float(x) * (len(opts) - 1)))]
    return hp


# --------------------------------------------------------------------------
# SAME-padding shape arithmetic (TF semantics: out = ceil(n / stride))
# --------------------------------------------------------------------------
def half_up(n: int) -> int:
    return (n + 1) // 2


# --------------------------------------------------------------------------
# Parameter tensors in CANONICAL ORDER.  This order (and the per-tensor layouts)
# is shared by the HIP library, the oracle and the tests:
#   conv kernel  [C_out][kh][kw][C_in]   (the transpose of Keras' HWIO)
#   conv bias    [C_out]
#   bn gamma, bn beta [C]      (trainable)      bn moving mean, var [C] (state)
#   dense kernel [out][in]     dense bias [out]
# --------------------------------------------------------------------------
def layer_specs(g: Sequence[int], variant: int, classes: int) -> List[Dict]:
    """Flat description of every parameterised layer, forward order.

    Each entry: {"kind": "conv"|"bn"|"dense", "name", shapes...}.  ``variant``
    selects the reference topology (A: nsga_penalty.py:255-301,
    B: sa_nsga_penalty.py:151-165).
    """
    f, k, bn, R, fc, _ = g
    specs: List[Dict] = []

    def conv(name, cin, cout, ks, stride=1):
        specs.append({"kind": "conv", "name": name, "cin": cin, "cout": cout, "k": ks, "stride": stride})

    def bnl(name, c):
        specs.append({"kind": "bn", "name": name, "c": c})

    conv("conv1", 1, f, k)
    if bn:
        bnl("bn1", f)
    if variant == VARIANT_A:
        conv("conv2", f, f, k)
        if bn:
            bnl("bn2", f)
    c = f
    for r in range(R):
        conv(f"res{r}_skip", c, 2 * c, 1, 2)
        conv(f"res{r}_conv1", c, 2 * c, k)
        if bn:
            bnl(f"res{r}_bn1", 2 * c)
        if variant == VARIANT_A:
            conv(f"res{r}_conv2", 2 * c, 2 * c, k)
            if bn:
                bnl(f"res{r}_bn2", 2 * c)
        c *= 2
    prev = c
    for i, units in enumerate(FC_LAYER_CONFIGS[fc]):
        specs.append({"kind": "dense", "name": f"fc{i + 1}", "cin": prev, "cout": units})
        prev = units
    specs.append({"kind": "dense", "name": "output_layer", "cin": prev, "cout": classes})
    return specs


def param_tensors(g: Sequence[int], variant: int, classes: int) -> List[Tuple[str, Tuple[int, ...], str]]:
    """[(name, shape, role)] in canonical order; role in
    {"kernel","bias","gamma","beta","moving_mean","moving_var"}."""
    out = []
    for s in layer_specs(g, variant, classes):
        if s["kind"] == "conv":
            out.append((s["name"] + "/kernel", (s["cout"], s["k"], s["k"], s["cin"]), "kernel"))
            out.append((s["name"] + "/bias", (s["cout"],), "bias"))
        elif s["kind"] == "bn":
            out.append((s["name"] + "/gamma", (s["c"],), "gamma"))
            out.append((s["name"] + "/beta", (s["c"],), "beta"))
            out.append((s["name"] + "/moving_mean", (s["c"],), "moving_mean"))
            out.append((s["name"] + "/moving_var", (s["c"],), "moving_var"))
        else:
            out.append((s["name"] + "/kernel", (s["cout"], s["cin"]), "kernel"))
            out.append((s["name"] + "/bias", (s["cout"],), "bias"))
    return out


def param_count(g: Sequence[int], variant: int, classes: int) -> int:
    """Keras ``model.count_params()`` for the gene, closed form.

    Counts trainable weights AND the non-trainable BN moving statistics, as
    count_params() does (nsga_penalty.py:341).
    """
    f, k, bn, R, fc, _ = (int(v) for v in g)
    kk = k * k
    if variant == VARIANT_A:
        p = (kk * f + f) + (kk * f * f + f) + (8 * f if bn else 0)
        c = f
        for _ in range(R):
            p += c * 2 * c + 2 * c
            p += kk * c * 2 * c + 2 * c
            p += kk * (2 * c) * (2 * c) + 2 * c
            p += 16 * c if bn else 0
            c *= 2
    else:
        p = (kk * f + f) + (4 * f if bn else 0)
        c = f
        for _ in range(R):
            p += c * 2 * c + 2 * c
            p += kk * c * 2 * c + 2 * c
            p += 8 * c if bn else 0
            c *= 2
    prev = c
    for units in FC_LAYER_CONFIGS[fc]:
        p += prev * units + units
        prev = units
    p += prev * classes + classes
    return p


def model_size_mb(g: Sequence[int], variant: int, classes: int) -> float:
    """compute_model_size_mb (nsga_penalty.py:337-344), bit-exact in float64."""
    return (param_count(g, variant, classes) * 4) / (1024 ** 2)


def fwd_flops_per_sample(g: Sequence[int], variant: int, classes: int, T: int, F: int) -> int:
    """Algorithmic forward FLOPs per sample, 2*MAC for conv/dense only."""
    f, k, _, R, fc, _ = (int(v) for v in g)
    kk = k * k
    H, W = T, F
    fl = 2 * H * W * kk * 1 * f
    if variant == VARIANT_A:
        fl += 2 * H * W * kk * f * f
    h, w = half_up(H), half_up(W)
    c = f
    for _ in range(R):
        h2, w2 = half_up(h), half_up(w)
        fl += 2 * h2 * w2 * c * 2 * c
        fl += 2 * h * w * kk * c * 2 * c
        if variant == VARIANT_A:
            fl += 2 * h * w * kk * 2 * c * 2 * c
        h, w, c = h2, w2, 2 * c
    prev = c
    for units in FC_LAYER_CONFIGS[fc]:
        fl += 2 * prev * units
        prev = units
    fl += 2 * prev * classes
    return fl


def eval_flops(g, variant, classes, T, F, n_train, n_val, epochs_run, post_passes) -> int:
    """W = F*(3*N_tr*E + N_val*E + N_val*P)  (SURVEY §8d)."""
    return fwd_flops_per_sample(g, variant, classes, T, F) * (
        3 * n_train * epochs_run + n_val * epochs_run + n_val * post_passes)


def lpt_assign(costs: Sequence[float], world: int) -> List[List[int]]:
    """Longest-processing-time greedy: candidate indices per rank.

    Deterministic: ties broken by index, so every SPMD rank computes the same
    assignment without communicating.
    """
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0.0] * world
    buckets: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda q: (loads[q], q))
        buckets[r].append(i)
        loads[r] += costs[i]
    return buckets
