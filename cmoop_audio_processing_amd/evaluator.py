"""Host-side mirror of the reference's evaluator surface, backed by libcmoop_hip.so.

Same names, argument meaning and result schema as the reference:

* ``evaluate_individual(hparams) -> (accuracy, size_mb, fpr)``
      nsga_penalty.py:368-395, sa_nsga_penalty.py:205-229
* ``compute_objectives_and_constraints(population) -> [{'hparams','objs','CV'}]``
      nsga_penalty.py:418-442, sa_nsga_penalty.py:231-253
  (results hold a REFERENCE to the caller's hparams dict, like :438)
* a duck-typed pymoo ``Problem._evaluate(X, out)`` over the [0,1]^6 codec of
  mobo_penalty.py:305-338 (pymoo itself is pinned but never imported by the
  reference).

The reference closes over module globals (X_train, ..., CLASSES, EPOCHS, ...,
thresholds); here they live in a ``PopulationEvaluator``; ``install()`` binds
the module-level functions to one, so a host loop written against the
reference runs unchanged.  Per-script quirks are selected by
``EvalConfig.preset(<script name>)``.

Multi-GPU: under ``torch.distributed`` every rank runs the same seeded host
loop; ``compute_objectives_and_constraints`` shards the generation's candidates
over the ranks -- by default through ONE shared longest-first queue (a fetch-add
counter on the c10d store, ``queued_map``), optionally by static LPT buckets on
closed-form FLOPs (``sharded_map``) -- and exchanges the objective vectors with ONE
all_gather (RCCL over xGMI) per generation; no other collective.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, replace
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib, genes as G

FPR_CODES = {"v1": 0, "v1_quirk": 1, "v3": 2}
ACC_CODES = {"last": 0, "evaluate": 1}
# arithmetic of the MFMA GEMMs: "fp32" (the reference's, exact fp32 MFMA; CMOOP_GEMM_MODE may override for experiments),
# "bf16x3" (fp32 operands split exactly into 3 bf16: fp32-accurate, not bit-exact), "bf16" (BASELINE configs[4])
GEMM_CODES = {"fp32": 0, "bf16x3": 2, "bf16": 3}


@dataclass(frozen=True)
class EvalConfig:
    """The reference's module constants, frozen (defaults = nsga_penalty.py:159-208)."""
    variant: str = "A"            # build_model topology
    classes: int = 10             # CLASSES
    epochs: int = 300             # EPOCHS
    batch: int = 64               # BATCH_SIZE
    patience: int = 5             # PATIENCE
    early_stop: bool = True
    restore_best: bool = False    # EarlyStopping(restore_best_weights=...)
    acc_readout: str = "last"     # 'last' = history['val_accuracy'][-1]; 'evaluate' = model.evaluate
    fpr_variant: str = "v1_quirk"
    min_accuracy: float = 0.9     # MIN_ACCURACY
    max_model_size: float = 2.5   # MAX_MODEL_SIZE
    max_fpr: float = 0.1          # MAX_FPR
    seed: int = 0                 # the reference seeds nothing; the build makes runs reproducible
    n_slots: int = 8              # candidates in flight per GPU (each on its own HIP stream; 6-16 measure the same)
    eval_batch: int = 256
    profile_every: int = 0
    compute: str = "fp32"       # GEMM_CODES: arithmetic of the conv/dense MFMA kernels
    lr: float = 1e-3
    dropout: float = 0.3
    shuffle: bool = True
    verbose: bool = False         # print the reference's per-candidate lines
    # multi-GPU candidate placement: "dynamic" = every rank's workers drain ONE longest-first queue (a fetch-add
    # counter on the c10d store; balances early-stopped runs whose epoch counts are unknown in advance);
    # "static" = LPT buckets by closed-form FLOPs.  Results are bit-identical either way (per-candidate seeds).
    schedule: str = "dynamic"
    # result packing of compute_objectives_and_constraints: "all" = [-acc, size, fpr] (nsga_penalty.py:418-442); the
    # bi-objective ablations keep the third quantity as a tracked key and drop its constraint from CV:
    # "acc_fpr" (acc_fpr_nsga_1.py:283-310, 'size_metric'), "acc_size" (acc_size_nsga_1.py:283-311, 'fpr_metric'),
    # "size_fpr" (size_fpr_nsga_1.py:283-310, 'acc_metric')
    objectives: str = "all"

    @staticmethod
    def preset(script: str, **over) -> "EvalConfig":
        """Protocol of one reference script (file:line in SURVEY §8a)."""
        table = {
            # nsga_penalty.py: topology A, no weight restore (:382), last-epoch accuracy (:384),
            # y_true quirk (:387), thresholds :206-208
            "nsga_penalty": dict(variant="A", classes=10, restore_best=False, acc_readout="last",
                                 fpr_variant="v1_quirk", min_accuracy=0.9, max_model_size=2.5, max_fpr=0.1),
            # sa_nsga_penalty.py: topology B, restore_best (:215), evaluate() (:219), thresholds :122-124
            "sa_nsga_penalty": dict(variant="B", classes=11, restore_best=True, acc_readout="evaluate",
                                    fpr_variant="v1", min_accuracy=0.75, max_model_size=2.5, max_fpr=0.09),
            # mobo_penalty.py: topology A, restore_best (:227) but last-epoch accuracy (:236); :114-116
            "mobo_penalty": dict(variant="A", classes=10, restore_best=True, acc_readout="last",
                                 fpr_variant="v1", min_accuracy=0.90, max_model_size=2.5, max_fpr=0.09),
            # ablation_study/sa_nsga_local.py: topology B, FPR V3 (:138-141)
            "sa_nsga_local": dict(variant="B", classes=10, restore_best=True, acc_readout="evaluate",
                                  fpr_variant="v3", min_accuracy=0.90, max_model_size=2.5, max_fpr=0.09),
            # ablation_study/init_sa_nsga_local.py: topology B, vectorised V1 (:137-143)
            "init_sa_nsga_local": dict(variant="B", classes=10, restore_best=True, acc_readout="evaluate",
                                       fpr_variant="v1", min_accuracy=0.90, max_model_size=2.5, max_fpr=0.09),
            # ablation_study/{acc_fpr,acc_size,size_fpr}_nsga_1.py: topology A, restore_best (:246), evaluate() (:249),
            # FPR V1 on y.flatten() (:225-237), train-only scaler (:90-100), thresholds :148-149, two objectives
            "acc_fpr_nsga_1": dict(variant="A", classes=10, restore_best=True, acc_readout="evaluate", fpr_variant="v1",
                                   min_accuracy=0.90, max_model_size=2.5, max_fpr=0.09, objectives="acc_fpr"),
            "acc_size_nsga_1": dict(variant="A", classes=10, restore_best=True, acc_readout="evaluate", fpr_variant="v1",
                                    min_accuracy=0.90, max_model_size=2.5, max_fpr=0.09, objectives="acc_size"),
            "size_fpr_nsga_1": dict(variant="A", classes=10, restore_best=True, acc_readout="evaluate", fpr_variant="v1",
                                    min_accuracy=0.90, max_model_size=2.5, max_fpr=0.09, objectives="size_fpr"),
        }
        if script not in table:
            raise KeyError(f"unknown preset {script!r}; have {sorted(table)}")
        return replace(EvalConfig(**table[script]), **over)

    def to_struct(self) -> "_lib.Config":
        c = _lib.default_config()
        c.variant = G.VARIANT_NAMES[self.variant]
        c.classes, c.epochs, c.batch, c.patience = self.classes, self.epochs, self.batch, self.patience
        c.early_stop, c.restore_best = int(self.early_stop), int(self.restore_best)
        c.acc_readout = ACC_CODES[self.acc_readout]
        c.fpr_variant = FPR_CODES[self.fpr_variant]
        c.shuffle = int(self.shuffle)
        c.eval_batch, c.n_slots, c.profile_every = self.eval_batch, self.n_slots, self.profile_every
        c.lr, c.dropout = self.lr, self.dropout
        c.gemm_mode = GEMM_CODES[self.compute]
        return c


def _as_device_features(x):
    import torch
    t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))
    if t.dim() == 4 and t.shape[-1] == 1:      # the channel axis prepare_dataset adds (nsga_penalty.py:151-153)
        t = t[..., 0]
    if t.dim() != 3:
        raise ValueError(f"features must be [N,T,F] or [N,T,F,1], got {tuple(t.shape)}")
    return t.to(device="cuda", dtype=torch.float32).contiguous()


def _as_device_labels(y):
    import torch
    t = y if isinstance(y, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(y))
    return t.reshape(-1).to(device="cuda", dtype=torch.int32).contiguous()   # (N,1) -> (N,)  (nsga_penalty.py:74-76)


def _dist():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist
    except Exception:
        pass
    return None


def _multi(dist) -> bool:
    """More than one rank -- or CMOOP_FORCE_COLLECTIVES=1, which sends a ONE-rank process group through the same store
    counter and all_gather as N ranks (how the RCCL path is executed on a single-GPU box: tests/test_gpu_net.py)."""
    import os
    return dist is not None and (dist.get_world_size() > 1 or os.environ.get("CMOOP_FORCE_COLLECTIVES") == "1")


def sharded_map(local_fn: Callable[[List[int]], np.ndarray], costs: Sequence[float], width: int,
                device: str = "cpu") -> np.ndarray:
    """Evaluate items 0..n-1 across the ranks of the default process group.

    ``local_fn(indices) -> float64 [len(indices), width]`` runs on this rank for its
    LPT bucket; results are exchanged with ONE all_gather of a ``[slots, width]``
    tensor per rank.  Without a process group everything runs locally.
    """
    n = len(costs)
    dist = _dist()
    if not _multi(dist):
        return np.asarray(local_fn(list(range(n))), dtype=np.float64).reshape(n, width)
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    if str(dist.get_backend()).lower() == "gloo":
        device = "cpu"            # CPU rehearsals (tests, single-GPU boxes); RCCL ("nccl") exchanges device tensors
    buckets = G.lpt_assign(costs, world)
    slots = max(1, max(len(b) for b in buckets))
    mine = buckets[rank]
    local = np.full((slots, width), np.nan, dtype=np.float64)
    if mine:
        local[:len(mine)] = np.asarray(local_fn(mine), dtype=np.float64).reshape(len(mine), width)
    send = torch.from_numpy(local).to(device)
    recv = torch.empty((world * slots, width), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(recv, send)
    allr = recv.cpu().numpy().reshape(world, slots, width)
    out = np.empty((n, width), dtype=np.float64)
    for r, b in enumerate(buckets):
        for j, i in enumerate(b):
            out[i] = allr[r, j]
    return out


_queue_serial = [0]   # queue keys are unique per (evaluator, generation); SPMD ranks construct evaluators in the same order


def _default_store():
    import torch.distributed.distributed_c10d as c10d
    return c10d._get_default_store()


def queue_plan(costs: Sequence[float], world: int, slots: int) -> Tuple[List[int], int, List[List[int]]]:
    """(cost-sorted order, workers per rank W, head lists per rank) of the cross-rank queue.

    W = min(slots, ceil(n / world)): no rank STARTS more than its fair share of the generation at once.  Without that
    bound 8 ranks x 8 worker threads drain a 40-candidate queue in one burst of fetch-adds and who trains what is a race
    (the rank that hosts the store answers itself fastest and would collect the largest candidates).  The first
    world * W positions of the longest-first order -- what the workers start with -- are therefore dealt out
    deterministically, longest first to the least-loaded rank that still has a free worker (capacity-constrained LPT on
    the closed-form cost); only what is left after that goes through the shared counter.  n <= world * W: the whole
    generation is that static deal.  Same inputs on every rank -> same plan, no communication."""
    n = len(costs)
    order = sorted(range(n), key=lambda i: (-costs[i], i))      # longest first, ties by index (same on every rank)
    W = max(1, min(int(slots), -(-n // max(world, 1)))) if n else 1
    heads: List[List[int]] = [[] for _ in range(world)]
    loads = [0.0] * world
    for i in order[:min(n, world * W)]:
        r = min((q for q in range(world) if len(heads[q]) < W), key=lambda q: (loads[q], q))
        heads[r].append(i)
        loads[r] += costs[i]
    return order, W, heads


def queued_map(local_pull_fn: Callable[..., Dict[int, Sequence[float]]], costs: Sequence[float], width: int,
               key: str, device: str = "cpu", slots: Optional[int] = None, stats: Optional[Dict] = None) -> np.ndarray:
    """Evaluate items 0..n-1 across the ranks through ONE shared longest-first queue.

    ``slots`` (worker threads a rank may run): with it, ``local_pull_fn(pull, workers)`` is called with the number of
    workers to start, each worker's FIRST item comes from this rank's share of ``queue_plan``'s deterministic deal and
    the shared counter hands out the rest; without it (``local_pull_fn(pull)``) every pull goes to the counter.

    Every rank's workers call ``pull()`` -- a fetch-add on the process group's c10d store (``store.add(key, 1)``,
    no collective) -- for the position in the cost-sorted order of the next item to evaluate; ``pull() < 0`` means
    the queue is drained.  ``local_pull_fn(pull) -> {index: row[width]}`` runs this rank's workers.  The results are
    then exchanged with ONE all_gather of ``[n, width+1]`` float64 per rank (last column = "I evaluated it").
    This is the cross-rank twin of the in-GPU queue of ``eval_population`` (csrc/net.hip): with early stopping the
    epochs a candidate runs are unknown in advance (7...173 observed), which static LPT buckets cannot balance.
    Without a process group the queue is a local counter.

    ``stats`` (optional dict) receives what makes a multi-GPU run diagnosable: ``store_adds`` / ``store_add_us_mean`` /
    ``store_add_us_max`` (round trips of the shared counter from this rank), ``dealt_here`` (candidates this rank started
    from the deterministic deal), ``local_s`` (wall seconds this rank spent evaluating) and ``all_gather_ms``.
    """
    import itertools
    import threading
    import time as _time
    add_us: List[float] = []
    n = len(costs)
    dist = _dist()
    multi = _multi(dist)
    world = dist.get_world_size() if multi else 1
    order, workers, heads = queue_plan(costs, world, slots if slots is not None else max(n, 1))
    if slots is None:
        workers, head, dealt = None, [], 0          # plain queue: every position through the counter
    else:
        head, dealt = list(heads[dist.get_rank() if multi else 0]), sum(len(h) for h in heads)
    lock = threading.Lock()
    if multi:
        store = _default_store()

        def pull() -> int:
            with lock:
                if head:
                    return head.pop(0)
            t0 = _time.perf_counter()
            j = int(store.add(key, 1)) - 1 + dealt
            add_us.append((_time.perf_counter() - t0) * 1e6)
            return order[j] if j < n else -1
    else:
        ctr = itertools.count()

        def pull() -> int:
            with lock:
                if head:
                    return head.pop(0)
                j = next(ctr) + dealt
            return order[j] if j < n else -1
    n_dealt_here = len(head)
    t_local = _time.perf_counter()
    if not n:
        mine = {}
    elif workers is None:
        mine = local_pull_fn(pull)
    else:
        mine = local_pull_fn(pull, workers)
    t_local = _time.perf_counter() - t_local
    if stats is not None:
        stats.update(store_adds=len(add_us), store_add_us_mean=float(np.mean(add_us)) if add_us else 0.0,
                     store_add_us_max=float(np.max(add_us)) if add_us else 0.0, dealt_here=n_dealt_here, local_s=t_local,
                     all_gather_ms=0.0)
    local = np.full((n, width + 1), np.nan, dtype=np.float64)
    local[:, width] = 0.0
    for i, row in mine.items():
        local[i, :width] = np.asarray(row, dtype=np.float64).reshape(width)
        local[i, width] = 1.0
    if not multi:
        if n and not bool((local[:, width] == 1.0).all()):
            raise _lib.CmoopError("queued_map: the local workers left candidates unevaluated")
        return local[:, :width].copy()
    import torch
    world = dist.get_world_size()
    if str(dist.get_backend()).lower() == "gloo":
        device = "cpu"
    send = torch.from_numpy(local).to(device)
    recv = torch.empty((world * max(n, 1), width + 1), dtype=torch.float64, device=device)
    if n == 0:
        send = torch.zeros((1, width + 1), dtype=torch.float64, device=device)
    t_ag = _time.perf_counter()
    dist.all_gather_into_tensor(recv, send)
    if n == 0:
        return np.zeros((0, width), dtype=np.float64)
    allr = recv.cpu().numpy().reshape(world, n, width + 1)       # the .cpu() copy waits for the collective
    if stats is not None:
        stats["all_gather_ms"] = (_time.perf_counter() - t_ag) * 1e3
    owners = allr[:, :, width]
    if not bool((owners.sum(axis=0) == 1.0).all()):
        raise _lib.CmoopError(f"queued_map: every candidate must be evaluated by exactly one rank, got {owners.sum(axis=0)}")
    who = owners.argmax(axis=0)
    return allr[who, np.arange(n), :width].copy()


class PopulationEvaluator:
    """Holds the resident dataset + protocol and evaluates candidates on this rank's GPU."""

    def __init__(self, X_train, y_train, X_validation, y_validation, config: EvalConfig = EvalConfig()):
        import torch
        if not torch.cuda.is_available():
            raise _lib.CmoopError("PopulationEvaluator needs a GPU: the fitness path has no CPU fallback")
        _lib.lib()   # fail loudly now if the HIP library is missing
        self.config = config
        self.X_train, self.y_train = _as_device_features(X_train), _as_device_labels(y_train)
        self.X_val, self.y_val = _as_device_features(X_validation), _as_device_labels(y_validation)
        if self.X_train.shape[1:] != self.X_val.shape[1:]:
            raise ValueError("train / validation feature shapes differ")
        if len(self.X_train) != len(self.y_train) or len(self.X_val) != len(self.y_val):
            raise ValueError("features and labels differ in length")
        self.T, self.F = int(self.X_train.shape[1]), int(self.X_train.shape[2])
        self.evals_done = 0          # seeds are cfg.seed + running candidate index (same on every SPMD rank)
        _queue_serial[0] += 1
        self._queue_prefix, self._generation = f"cmoop/queue/{_queue_serial[0]}", 0
        self.last_rank_of: List[int] = []   # which rank trained each candidate of the last generation
        self.last_epochs_run: List[int] = []
        self.last_seconds: List[float] = []
        self.last_queue_stats: Dict = {}     # multi-GPU diagnostics of the last generation on THIS rank (queued_map's stats)
        torch.cuda.synchronize()

    # -- low level ------------------------------------------------------------
    def _dataset(self) -> "_lib.DatasetStruct":
        d = _lib.DatasetStruct()
        d.x_train, d.y_train, d.n_train = self.X_train.data_ptr(), self.y_train.data_ptr(), len(self.X_train)
        d.x_val, d.y_val, d.n_val = self.X_val.data_ptr(), self.y_val.data_ptr(), len(self.X_val)
        d.T, d.F = self.T, self.F
        return d

    def evaluate_genes(self, gene_list: Sequence[Sequence[int]], seeds: Sequence[int]) -> np.ndarray:
        """[n,6] genes -> float64 [n,5] = (accuracy, size_mb, fpr, epochs_run, seconds) on THIS GPU."""
        n = len(gene_list)
        out = np.zeros((n, 5), dtype=np.float64)
        if n == 0:
            return out
        for g in gene_list:
            G.validate_gene(g)
        genes = np.ascontiguousarray(np.asarray(gene_list, dtype=np.int32).reshape(n, 6))
        sd = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint64).astype(np.uint32))
        acc, size, fpr, secs = (np.zeros(n, np.float64) for _ in range(4))
        ep = np.zeros(n, np.int32)
        cfg, ds = self.config.to_struct(), self._dataset()
        _lib.check(_lib.lib().cmoop_eval_population(
            C.byref(cfg), C.byref(ds), _lib.ptr(genes), _lib.ptr(sd), C.c_int32(n), _lib.ptr(acc), _lib.ptr(size),
            _lib.ptr(fpr), _lib.ptr(ep), None, _lib.ptr(secs)))
        out[:, 0], out[:, 1], out[:, 2], out[:, 3], out[:, 4] = acc, size, fpr, ep, secs
        return out

    def evaluate_genes_pull(self, gene_list: Sequence[Sequence[int]], seeds: Sequence[int], pull: Callable[[], int],
                            workers: Optional[int] = None) -> Dict[int, np.ndarray]:
        """Train the candidates ``pull()`` hands to this GPU's worker threads (``workers`` of them, default cfg.n_slots,
        call it concurrently, from C++); returns {index: (accuracy, size_mb, fpr, epochs_run, seconds)}."""
        n = len(gene_list)
        if n == 0:
            return {}
        for g in gene_list:
            G.validate_gene(g)
        genes = np.ascontiguousarray(np.asarray(gene_list, dtype=np.int32).reshape(n, 6))
        sd = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint64).astype(np.uint32))
        acc, size, fpr, secs = (np.zeros(n, np.float64) for _ in range(4))
        ep, done = np.zeros(n, np.int32), np.zeros(n, np.int32)
        cfg, ds = self.config.to_struct(), self._dataset()
        if workers is not None:
            cfg.n_slots = max(1, int(workers))
        errors: List[BaseException] = []

        def _next(_ctx):
            try:
                return int(pull())
            except BaseException as e:      # never let an exception cross the C boundary: end this worker, re-raise below
                errors.append(e)
                return -1
        cb = _lib.NEXT_FN(_next)
        _lib.check(_lib.lib().cmoop_eval_population_pull(
            C.byref(cfg), C.byref(ds), _lib.ptr(genes), _lib.ptr(sd), C.c_int32(n), cb, None, _lib.ptr(acc), _lib.ptr(size),
            _lib.ptr(fpr), _lib.ptr(ep), None, _lib.ptr(secs), _lib.ptr(done)))
        if errors:
            raise errors[0]
        return {int(i): np.array([acc[i], size[i], fpr[i], ep[i], secs[i]], dtype=np.float64) for i in np.nonzero(done)[0]}

    # -- the reference's surface ------------------------------------------------
    def evaluate_individual(self, hparams: Dict):
        """(accuracy, size_mb, fpr) of one candidate -- nsga_penalty.py:368-395."""
        g = G.normalize_hparams(hparams)
        r = self.evaluate_genes([g], [self.config.seed + self.evals_done])[0]
        self.evals_done += 1
        self.last_epochs_run, self.last_seconds = [int(r[3])], [float(r[4])]
        self._print(r)
        return float(r[0]), float(r[1]), float(r[2])

    def compute_objectives_and_constraints(self, population: List[Dict]) -> List[Dict]:
        """nsga_penalty.py:418-442: objs = [-acc, size, fpr], CV = g1 + g2 + g3."""
        gl = [G.normalize_hparams(hp) for hp in population]
        n = len(gl)
        seeds = [self.config.seed + self.evals_done + i for i in range(n)]
        v = G.VARIANT_NAMES[self.config.variant]
        costs = [float(G.fwd_flops_per_sample(g, v, self.config.classes, self.T, self.F)) for g in gl]
        dist = _dist()
        multi = _multi(dist)
        self._generation += 1
        if not n:
            res = np.zeros((0, 6))
            if multi:   # keep the SPMD ranks in step: the empty generation still does its (empty) exchange
                queued_map(lambda pull: {}, [], 6, f"{self._queue_prefix}/{self._generation}", device="cuda")
        elif not multi:
            res = np.concatenate([self.evaluate_genes(gl, seeds), np.zeros((n, 1))], axis=1)
        elif self.config.schedule == "static":     # LPT buckets by closed-form FLOPs (fixed-epoch throughput runs)
            rank = float(dist.get_rank())
            res = sharded_map(lambda idx: np.concatenate([self.evaluate_genes([gl[i] for i in idx], [seeds[i] for i in idx]),
                                                          np.full((len(idx), 1), rank)], axis=1), costs, 6, device="cuda")
        else:                                       # one longest-first queue drained by all ranks (default)
            rank = float(dist.get_rank())
            self.last_queue_stats = {}
            res = queued_map(lambda pull, workers: {i: np.append(r, rank)
                                                    for i, r in self.evaluate_genes_pull(gl, seeds, pull, workers).items()},
                             costs, 6, f"{self._queue_prefix}/{self._generation}", device="cuda", slots=self.config.n_slots,
                             stats=self.last_queue_stats)
        self.last_rank_of = [int(r) for r in res[:, 5]]
        self.evals_done += n
        self.last_epochs_run = [int(e) for e in res[:, 3]]
        self.last_seconds = [float(s) for s in res[:, 4]]
        c = self.config
        results = []
        for ind, r in zip(population, res):
            acc, size_mb, fpr = float(r[0]), float(r[1]), float(r[2])
            self._print(r)
            results.append(pack_result(ind, acc, size_mb, fpr, c))
        return results

    def _print(self, r):
        if self.config.verbose:
            print(f"  -> True Eval: Acc={r[0]:.4f}, Size={r[1]:.2f}MB, FPR={r[2]:.4f}")


def pack_result(ind: Dict, acc: float, size_mb: float, fpr: float, c: EvalConfig) -> Dict:
    """One entry of compute_objectives_and_constraints' result list, in the schema of the script ``c.objectives`` names
    (holding a REFERENCE to the caller's hparams dict, as nsga_penalty.py:438 does)."""
    g1 = max(0.0, c.min_accuracy - acc)
    g2 = max(0.0, size_mb - c.max_model_size)
    g3 = max(0.0, fpr - c.max_fpr)
    if c.objectives == "all":
        return {"hparams": ind, "objs": [-acc, size_mb, fpr], "CV": g1 + g2 + g3}
    if c.objectives == "acc_fpr":
        return {"hparams": ind, "objs": [-acc, fpr], "size_metric": size_mb, "CV": g1 + g3}
    if c.objectives == "acc_size":
        return {"hparams": ind, "objs": [-acc, size_mb], "fpr_metric": fpr, "CV": g1 + g2}
    if c.objectives == "size_fpr":
        return {"hparams": ind, "acc_metric": acc, "objs": [size_mb, fpr], "CV": g2 + g3}
    raise ValueError(f"unknown objectives {c.objectives!r}")


class AudioNASProblem:
    """Duck-typed pymoo ``Problem`` over x in [0,1]^6 (codec: mobo_penalty.py:305-338).

    ``_evaluate(X, out)`` fills ``out['F'] = [-acc, size, fpr]`` and
    ``out['G'] = [MIN_ACC-acc, size-MAX_SIZE, fpr-MAX_FPR]`` (feasible iff <= 0).
    """
    n_var, n_obj, n_ieq_constr, n_constr = 6, 3, 3, 3

    def __init__(self, evaluator: PopulationEvaluator):
        self.evaluator = evaluator
        self.xl, self.xu = np.zeros(6), np.ones(6)

    def _evaluate(self, X, out, *args, **kwargs):
        X = np.atleast_2d(np.asarray(X, dtype=np.float64))
        pop = [G.vector_to_hparams(x) for x in X]
        res = self.evaluator.compute_objectives_and_constraints(pop)
        c = self.evaluator.config
        F = np.array([r["objs"] for r in res], dtype=np.float64).reshape(len(pop), 3)
        out["F"] = F
        out["G"] = np.stack([c.min_accuracy + F[:, 0], F[:, 1] - c.max_model_size, F[:, 2] - c.max_fpr], axis=1)

    def evaluate(self, X):
        out: Dict = {}
        self._evaluate(X, out)
        return out


# ---- module-level functions with the reference's names -------------------------
_default: Optional[PopulationEvaluator] = None


def install(evaluator: PopulationEvaluator) -> PopulationEvaluator:
    """Bind evaluate_individual / compute_objectives_and_constraints to ``evaluator``
    (the reference binds them to module globals at import time, nsga_penalty.py:167)."""
    global _default
    _default = evaluator
    return evaluator


def _need() -> PopulationEvaluator:
    if _default is None:
        raise _lib.CmoopError("no evaluator installed: call install(PopulationEvaluator(...)) first")
    return _default


def evaluate_individual(hparams: Dict):
    return _need().evaluate_individual(hparams)


def compute_objectives_and_constraints(population: List[Dict]) -> List[Dict]:
    return _need().compute_objectives_and_constraints(population)


def compute_model_size_mb(hparams: Dict, variant: str = "A", classes: int = 10) -> float:
    """compute_model_size_mb (nsga_penalty.py:337-344) without building a model."""
    return G.model_size_mb(G.normalize_hparams(hparams), G.VARIANT_NAMES[variant], classes)


def calculate_fpr(y_true, y_pred, num_classes: int, variant: str = "v1") -> float:
    """calculate_fpr (nsga_penalty.py:351-364 / sa_nsga_local.py:138-141) via the C ABI."""
    yt = np.ascontiguousarray(np.asarray(y_true).reshape(-1).astype(np.int32))
    yp = np.ascontiguousarray(np.asarray(y_pred).reshape(-1).astype(np.int32))
    if len(yt) != len(yp):
        raise ValueError("y_true and y_pred differ in length")
    out = C.c_double(0.0)
    _lib.check(_lib.lib().cmoop_calculate_fpr(_lib.ptr(yt), _lib.ptr(yp), C.c_int64(len(yt)), C.c_int32(num_classes),
                                              C.c_int32(FPR_CODES[variant]), C.byref(out)))
    return float(out.value)
