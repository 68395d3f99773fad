"""Host-side Kriging surrogate, infill selection, LHS initialisation, Lamarckian LCB local
search and the surrogate-assisted NSGA-II loop.

These decide WHICH genes reach the GPU hot path; they stay on the host (north-star) and
reuse scikit-learn's GaussianProcessRegressor as the reference does.  Restated from:
  SurrogateManager ............ sa_nsga_penalty.py:258-363 (+ return_std: sa_nsga_local.py:169-234)
  select_infill_points ........ sa_nsga_penalty.py:472-518
  SA-NSGA-II loop ............. sa_nsga_penalty.py:522-637
  perturb / LCB local search .. ablation_study/sa_nsga_local.py:351-433
  LHS initialisation .......... ablation_study/sa_nsga_init.py:369-396 (pyDOE lhs 'maximin')
  memetic loop ................ ablation_study/init_sa_nsga_local.py:388-470
Pinned by tests/golden/surrogate_golden.json (the reference's SurrogateManager and
select_infill_points executed in the build container under a fixed numpy seed).  The LHS
sampler is PARITY UNPINNED (pyDOE is not installed): it restates pyDOE's published
'maximin' criterion (best of 5 random Latin designs by minimum pairwise distance).
"""
from __future__ import annotations

import random
from copy import deepcopy
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import genes as G
from . import nsga

NUMERICAL = ("filters", "kernel_size", "residual_blocks", "fc_layers")
CATEGORICAL = ("use_bn", "use_dropout")
TARGETS = ("neg_acc", "size", "fpr", "cv")


class SurrogateManager:
    """Four Gaussian processes (Matern-1.5 x constant + white noise, 10 optimiser restarts) on
    standardised targets (-acc, size, fpr, CV); features = 4 numeric genes + one-hot of the
    two boolean genes over the categories seen at fit time (unknown category -> zeros, as
    OneHotEncoder(handle_unknown='ignore'))."""

    def __init__(self, random_state=None):
        from sklearn.gaussian_process import GaussianProcessRegressor
        from sklearn.gaussian_process.kernels import ConstantKernel as C, Matern, WhiteKernel
        self.is_fitted = False
        kernel = C(1.0) * Matern(length_scale=1.0, nu=1.5) + WhiteKernel(noise_level=0.1)
        self.models = {k: GaussianProcessRegressor(kernel=deepcopy(kernel), n_restarts_optimizer=10, random_state=random_state)
                       for k in TARGETS}
        self.table: Dict[Tuple, Tuple[Dict, List[float]]] = {}   # gene tuple -> (hparams, targets); insertion-ordered
        self.cats: Dict[str, List] = {}
        self.y_mean: Dict[str, float] = {}
        self.y_scale: Dict[str, float] = {}
        self.y_var: Dict[str, float] = {}

    @staticmethod
    def _key(hp: Dict) -> Tuple:
        return tuple(hp[k] for k in NUMERICAL + CATEGORICAL)

    def _encode(self, hparams_list: Sequence[Dict]) -> np.ndarray:
        cols = [[float(hp[k]) for k in NUMERICAL] for hp in hparams_list]
        X = np.asarray(cols, dtype=np.float64).reshape(len(hparams_list), len(NUMERICAL))
        onehot = []
        for k in CATEGORICAL:
            for c in self.cats[k]:
                onehot.append([1.0 if hp[k] == c else 0.0 for hp in hparams_list])
        if onehot:
            X = np.concatenate([X, np.asarray(onehot, dtype=np.float64).T.reshape(len(hparams_list), -1)], axis=1)
        return X

    def update(self, hparams_list: Sequence[Dict], results_list: Sequence[Dict]) -> None:
        # append, dropping earlier rows with the same genes (drop_duplicates(keep='last') keeps the LAST
        # occurrence's position)
        for hp, res in zip(hparams_list, results_list):
            key = self._key(hp)
            self.table.pop(key, None)
            self.table[key] = (hp, [res["objs"][0], res["objs"][1], res["objs"][2], res["CV"]])
        rows = list(self.table.values())
        hps = [r[0] for r in rows]
        self.cats = {k: sorted(set(hp[k] for hp in hps)) for k in CATEGORICAL}
        X = self._encode(hps)
        for j, key in enumerate(TARGETS):
            y = np.asarray([r[1][j] for r in rows], dtype=np.float64)
            mean, var = float(y.mean()), float(y.var())
            scale = np.sqrt(var) if var > 0 else 1.0           # StandardScaler: zero variance -> scale 1
            if scale < 10 * np.finfo(np.float64).eps:
                scale = 1.0
            self.y_mean[key], self.y_scale[key], self.y_var[key] = mean, float(scale), var
            self.models[key].fit(X, ((y - mean) / scale).reshape(-1, 1))
        self.is_fitted = True

    def predict(self, hparams_list: Sequence[Dict], return_std: bool = False):
        if not self.is_fitted:
            raise RuntimeError("Surrogate models must be fitted before prediction.")
        X = self._encode(hparams_list)
        preds, stds = {}, {}
        for key in TARGETS:
            if return_std:
                m, s = self.models[key].predict(X, return_std=True)
                stds[key] = (np.asarray(s).ravel() * np.sqrt(self.y_var[key])) if self.y_var[key] > 0 else np.zeros(len(X))
            else:
                m = self.models[key].predict(X)
            preds[key] = np.asarray(m).ravel() * self.y_scale[key] + self.y_mean[key]
        return (preds, stds) if return_std else preds

    def predict_and_structure(self, hparams_list: Sequence[Dict]) -> List[Dict]:
        p = self.predict(hparams_list)
        return [{"hparams": hp, "objs": [p["neg_acc"][i], p["size"][i], p["fpr"][i]], "CV": max(0, p["cv"][i])}
                for i, hp in enumerate(hparams_list)]


def select_infill_points(predicted: Sequence[Dict], num_to_select: int):
    """Feasible-first ranking: predicted-feasible by the sum of min-max normalised objectives,
    then infeasible by predicted CV; returns (indices, hparams) of the top ``num_to_select``."""
    feas = [(i, r) for i, r in enumerate(predicted) if r["CV"] < nsga.EPSILON]
    infe = [(i, r) for i, r in enumerate(predicted) if not r["CV"] < nsga.EPSILON]
    order: List[int] = []
    if feas:
        objs = np.array([r["objs"] for _, r in feas], dtype=np.float64)
        rng_ = objs.max(axis=0) - objs.min(axis=0)
        rng_[rng_ < nsga.EPSILON] = 1.0
        scores = ((objs - objs.min(axis=0)) / rng_).sum(axis=1)
        order += [i for i, _ in sorted(zip([f[0] for f in feas], scores), key=lambda p: p[1])]
    order += [i for i, _ in sorted(infe, key=lambda it: it[1]["CV"])]
    sel = order[:num_to_select]
    return sel, [predicted[i]["hparams"] for i in sel]


# ---- Latin hypercube initialisation ------------------------------------------------------
def lhs_maximin(dims: int, samples: int, rs: np.random.RandomState, iterations: int = 5) -> np.ndarray:
    best, best_d = None, -1.0
    for _ in range(iterations):
        cut = np.linspace(0, 1, samples + 1)
        u = rs.rand(samples, dims)
        pts = cut[:samples, None] + u * (cut[1:, None] - cut[:samples, None])
        H = np.empty_like(pts)
        for j in range(dims):
            H[:, j] = pts[rs.permutation(samples), j]
        if samples > 1:
            diff = H[:, None, :] - H[None, :, :]
            d = np.sqrt((diff ** 2).sum(-1))[np.triu_indices(samples, 1)].min()
        else:
            d = 0.0
        if d > best_d:
            best, best_d = H, d
    return best


def latin_hypercube_initialization(pop_size: int, seed: int = 0) -> List[Dict]:
    unit = lhs_maximin(len(G.GENE_KEYS), pop_size, np.random.RandomState(seed))
    pop = []
    for row in unit:
        pop.append({k: opts[min(int(row[i] * len(opts)), len(opts) - 1)]
                    for i, (k, opts) in enumerate(zip(G.GENE_KEYS, G.GENE_OPTIONS))})
    return pop


# ---- Lamarckian local search on the surrogate's lower confidence bound ----------------------
def perturb_hparams(hp: Dict, rng: random.Random) -> Dict:
    out = deepcopy(hp)
    key = rng.choice(list(G.GENE_KEYS))
    opts = G.GENE_OPTIONS[G.GENE_KEYS.index(key)]
    if isinstance(opts[0], bool):
        out[key] = not out[key]
    else:
        others = [v for v in opts if v != out[key]]
        if others:
            out[key] = rng.choice(others)
    return out


def lcb_dominates(a: Dict, b: Dict) -> bool:
    return all(x <= y for x, y in zip(a["lcb_objs"], b["lcb_objs"])) and any(x < y for x, y in zip(a["lcb_objs"], b["lcb_objs"]))


def perform_local_search(offspring: List[Dict], surrogate: SurrogateManager, rng: random.Random, k_lcb: float = 1.0,
                         sweeps: int = 5) -> List[Dict]:
    """5 sweeps over the LCB-non-dominated offspring: replace an elite's genes by a one-gene neighbour whenever
    the neighbour's LCB vector dominates it (entries are edited in place, as the reference does)."""
    for sol in offspring:
        sol["lcb_objs"] = (np.array(sol["objs"]) - k_lcb * np.array(sol["stds"])).tolist()
    elite = [i for i in range(len(offspring))
             if not any(j != i and lcb_dominates(offspring[j], offspring[i]) for j in range(len(offspring)))]
    for _ in range(sweeps):
        for i in elite:
            cand = perturb_hparams(offspring[i]["hparams"], rng)
            p, s = surrogate.predict([cand], return_std=True)
            lcb = [p[k][0] - k_lcb * s[k][0] for k in ("neg_acc", "size", "fpr")]
            if lcb_dominates({"lcb_objs": lcb}, offspring[i]):
                offspring[i].update(hparams=cand, lcb_objs=lcb, objs=[p[k][0] for k in ("neg_acc", "size", "fpr")],
                                    stds=[s[k][0] for k in ("neg_acc", "size", "fpr")])
    return [sol["hparams"] for sol in offspring]


# ---- surrogate-assisted (memetic) NSGA-II -----------------------------------------------------
def sa_nsga2(evaluate: Callable[[List[Dict]], List[Dict]], pop_size: int, max_gen: int, infill_percent: float = 0.2,
             seed: int = 0, init: str = "random", local_search: bool = False,
             on_generation: Optional[Callable[[int, List[Dict]], None]] = None):
    """SA-NSGA-II: only max(1, int(pop*infill_percent)) offspring per generation get a TRUE (GPU) evaluation; the
    rest keep their predicted objectives.  ``init='lhs'`` + ``local_search=True`` is the full memetic method."""
    rng = random.Random(seed)
    pop0 = latin_hypercube_initialization(pop_size, seed) if init == "lhs" else nsga.initialize_population(pop_size, rng)
    pop_data = evaluate(pop0)
    sm = SurrogateManager(random_state=seed)
    sm.update([d["hparams"] for d in pop_data], pop_data)
    history: List[List[Dict]] = []
    true_evals = len(pop0)
    for gen in range(max_gen):
        lam = nsga.get_lambda(gen, max_gen)
        parents = [pop_data[nsga.tournament_selection(pop_data, lam, rng)]["hparams"] for _ in range(pop_size)]
        offspring: List[Dict] = []
        while len(offspring) < pop_size:
            p1, p2 = rng.sample(parents, 2)
            c1, c2 = nsga.crossover(p1, p2, rng) if rng.random() < nsga.CROSSOVER_PROB else (deepcopy(p1), deepcopy(p2))
            offspring += [nsga.mutate(c1, rng), nsga.mutate(c2, rng)]
        offspring = offspring[:pop_size]
        if local_search:
            p, s = sm.predict(offspring, return_std=True)
            pred = [{"hparams": hp, "objs": [p[k][i] for k in ("neg_acc", "size", "fpr")],
                     "stds": [s[k][i] for k in ("neg_acc", "size", "fpr")], "CV": max(0, p["cv"][i])}
                    for i, hp in enumerate(offspring)]
            offspring = perform_local_search(pred, sm, rng)
        predicted = sm.predict_and_structure(offspring)
        n_infill = max(1, int(pop_size * infill_percent))
        idx, infill = select_infill_points(predicted, n_infill)
        true = evaluate(infill)
        true_evals += len(infill)
        sm.update(infill, true)
        off_data = list(predicted)
        for i, res in zip(idx, true):
            off_data[i] = res
        combined = list(pop_data) + off_data
        nxt: List[Dict] = []
        for front in nsga.fast_non_dominated_sort(combined, lam):
            if len(nxt) + len(front) <= pop_size:
                nxt += [combined[i] for i in front]
            else:
                d = nsga.crowding_distance(front, combined)
                nxt += [combined[i] for i in sorted(front, key=lambda i: d.get(i, 0), reverse=True)[:pop_size - len(nxt)]]
                break
        pop_data = nxt
        history.append(nsga.generation_records(gen, pop_data))
        if on_generation:
            on_generation(gen, pop_data)
    return nsga.feasible_pareto(pop_data), history, true_evals
