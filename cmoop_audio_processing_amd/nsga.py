"""Host-side constrained NSGA-II with a linearly annealed penalty.

The caller of the hot path, restated as a small library so a whole search can run
against the GPU evaluator (BASELINE config[0]: pop=4 gen=2 plumbing; hypervolume at
equal generation count).  It stays on the host, as the north-star prescribes.
Follows /root/reference/nsga_penalty.py:
  get_lambda :217-219 (guarded like sa_nsga_penalty.py:130-132, quirk Q8)
  dominates / fast_non_dominated_sort / crowding_distance :448-524
  tournament_selection :528-538, crossover :541-579, mutate :581-603
  nsga2 main loop :610-776 (empty-feasible-set guard instead of quirk Q10)
Pinned by tests/golden/nsga_ops_golden.json (outputs of the reference's own functions).

All randomness goes through one ``random.Random``; calls are made in the same
order as the reference makes them on the global ``random`` module, so a seeded
reference run and a seeded run of this loop draw identical gene sequences.
"""
from __future__ import annotations

import csv
import random
from copy import deepcopy
from typing import Callable, Dict, List, Optional, Sequence, Tuple

from . import genes as G

LAMBDA_INITIAL, LAMBDA_FINAL = 1.0, 50.0     # nsga_penalty.py:214-215
CROSSOVER_PROB, MUTATION_PROB = 0.9, 0.2     # :201-202
EPSILON = 1e-6                               # :203


def get_lambda(gen: int, max_gen: int) -> float:
    frac = gen / float(max_gen - 1) if max_gen > 1 else 1.0
    return LAMBDA_INITIAL + frac * (LAMBDA_FINAL - LAMBDA_INITIAL)


def dominates(a: Dict, b: Dict, lam: float) -> bool:
    """Penalised dominance: compare objs + lam*CV component-wise."""
    pa = [f + lam * a["CV"] for f in a["objs"]]
    pb = [f + lam * b["CV"] for f in b["objs"]]
    strictly = False
    for x, y in zip(pa, pb):
        if x > y:
            return False
        if x < y:
            strictly = True
    return strictly


def fast_non_dominated_sort(results: Sequence[Dict], lam: float) -> List[List[int]]:
    n = len(results)
    if n == 0:
        return []
    dominated_by = [[] for _ in range(n)]
    counts = [0] * n
    first: List[int] = []
    for p in range(n):
        for q in range(n):
            if dominates(results[p], results[q], lam):
                dominated_by[p].append(q)
            elif dominates(results[q], results[p], lam):
                counts[p] += 1
        if counts[p] == 0:
            first.append(p)
    fronts = [first]
    while True:
        nxt = []
        for p in fronts[-1]:
            for q in dominated_by[p]:
                counts[q] -= 1
                if counts[q] == 0:
                    nxt.append(q)
        if not nxt:
            return fronts
        fronts.append(nxt)


def crowding_distance(front: Sequence[int], results: Sequence[Dict]) -> Dict[int, float]:
    dist = {i: 0.0 for i in front}
    for m in range(len(results[0]["objs"])):
        order = sorted(front, key=lambda i: results[i]["objs"][m])
        dist[order[0]] = dist[order[-1]] = float("inf")
        lo, hi = results[order[0]]["objs"][m], results[order[-1]]["objs"][m]
        if hi - lo < EPSILON:
            continue
        for k in range(1, len(order) - 1):
            dist[order[k]] += (results[order[k + 1]]["objs"][m] - results[order[k - 1]]["objs"][m]) / (hi - lo)
    return dist


def tournament_selection(results: Sequence[Dict], lam: float, rng: random.Random, k: int = 2) -> int:
    idxs = rng.sample(range(len(results)), k)
    best = idxs[0]
    for i in idxs[1:]:
        if dominates(results[i], results[best], lam):
            best = i
    return best


def crossover(p1: Dict, p2: Dict, rng: random.Random) -> Tuple[Dict, Dict]:
    """Uniform crossover: each of the six genes swaps with probability 0.5."""
    c1, c2 = deepcopy(p1), deepcopy(p2)
    for key in G.GENE_KEYS:
        if rng.random() < 0.5:
            c1[key], c2[key] = p2[key], p1[key]
    return c1, c2


def mutate(ind: Dict, rng: random.Random, prob: float = MUTATION_PROB) -> Dict:
    out = deepcopy(ind)
    for key, opts in zip(G.GENE_KEYS, G.GENE_OPTIONS):
        if rng.random() < prob:
            out[key] = (not out[key]) if key in ("use_bn", "use_dropout") else rng.choice(opts)
    return out


def initialize_population(pop_size: int, rng: random.Random) -> List[Dict]:
    return [G.random_hparams(rng) for _ in range(pop_size)]


def feasible_pareto(pop_data: Sequence[Dict]) -> List[Dict]:
    feas = [ind for ind in pop_data if ind["CV"] == 0]
    if not feas:
        return []
    return [feas[i] for i in fast_non_dominated_sort(feas, LAMBDA_FINAL)[0]]


def accuracy_size_fpr(ind: Dict) -> Tuple[float, float, float]:
    """(accuracy, size_mb, fpr) of a result entry in any of the four packings: three objectives
    (nsga_penalty.py:438) or a bi-objective ablation with the third quantity under its tracked key
    (acc_fpr_nsga_1.py:304-309 'size_metric', acc_size_nsga_1.py:306-310 'fpr_metric', size_fpr_nsga_1.py:304-309 'acc_metric')."""
    o = ind["objs"]
    if len(o) == 3:
        return -o[0], o[1], o[2]
    if "size_metric" in ind:
        return -o[0], ind["size_metric"], o[1]
    if "fpr_metric" in ind:
        return -o[0], o[1], ind["fpr_metric"]
    return ind["acc_metric"], o[0], o[1]


def generation_records(gen: int, pop_data: Sequence[Dict]) -> List[Dict]:
    """Rows with the reference's per-generation column schema (nsga_penalty.py:708-719; the ablations write the same
    columns, reading the tracked quantity from its key: acc_size_nsga_1.py:476-483)."""
    out = []
    for ind in pop_data:
        acc, size_mb, fpr = accuracy_size_fpr(ind)
        out.append({"Generation": gen, "Accuracy": acc, "Size_MB": size_mb, "FPR": fpr, "CV": ind["CV"], **ind["hparams"]})
    return out


def nsga2(evaluate: Callable[[List[Dict]], List[Dict]], pop_size: int, max_gen: int, seed: int = 0,
          on_generation: Optional[Callable[[int, List[Dict]], None]] = None):
    """(pareto_set, per-generation record lists).  ``evaluate`` is
    compute_objectives_and_constraints (the GPU hot path or any stand-in)."""
    rng = random.Random(seed)
    pop_data = evaluate(initialize_population(pop_size, rng))
    history: List[List[Dict]] = []
    for gen in range(max_gen):
        lam = get_lambda(gen, max_gen)
        parents = [tournament_selection(pop_data, lam, rng) for _ in range(pop_size)]
        offspring: List[Dict] = []
        for i1, i2 in zip(parents[0::2], parents[1::2]):
            a, b = pop_data[i1]["hparams"], pop_data[i2]["hparams"]
            if rng.random() < CROSSOVER_PROB:
                c1, c2 = crossover(a, b, rng)
            else:
                c1, c2 = deepcopy(a), deepcopy(b)
            offspring += [mutate(c1, rng), mutate(c2, rng)]
        if pop_size % 2 == 1:
            offspring.append(mutate(deepcopy(pop_data[parents[-1]]["hparams"]), rng))
        offspring = offspring[:pop_size]
        combined = list(pop_data) + list(evaluate(offspring))
        nxt: List[Dict] = []
        for front in fast_non_dominated_sort(combined, lam):
            if len(nxt) + len(front) <= pop_size:
                nxt += [combined[i] for i in front]
            else:
                d = crowding_distance(front, combined)
                keep = sorted(front, key=lambda i: d[i], reverse=True)[:pop_size - len(nxt)]
                nxt += [combined[i] for i in keep]
                break
        pop_data = nxt
        recs = generation_records(gen, pop_data)
        history.append(recs)
        if on_generation:
            on_generation(gen, pop_data)
    return feasible_pareto(pop_data), history


def write_records_csv(path: str, history: Sequence[Sequence[Dict]]) -> None:
    rows = [r for gen in history for r in gen]
    if not rows:
        return
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)


# ---------------------------------------------------------------------------------------
# Hypervolume (compare.ipynb:215-233 uses pygmo, absent here):
# minimisation objectives (-Accuracy, Size_MB, FPR), reference point = per-objective max
# over the union of the compared fronts + 1e-3.  Exact 3-D sweep (slice along the 3rd axis).
# ---------------------------------------------------------------------------------------
def _hv2d(points: List[Tuple[float, float]], ref: Tuple[float, float]) -> float:
    pts = sorted(p for p in points if p[0] < ref[0] and p[1] < ref[1])
    area, best_y = 0.0, ref[1]
    for x, y in pts:
        if y < best_y:
            area += (ref[0] - x) * (best_y - y)
            best_y = y
    return area


def hypervolume(points: Sequence[Sequence[float]], ref: Sequence[float]) -> float:
    pts = [tuple(float(v) for v in p) for p in points if all(float(v) < float(r) for v, r in zip(p, ref))]
    if not pts:
        return 0.0
    dim = len(ref)
    if dim == 1:
        return float(ref[0]) - min(p[0] for p in pts)
    if dim == 2:
        return _hv2d([(p[0], p[1]) for p in pts], (float(ref[0]), float(ref[1])))
    if dim != 3:
        raise ValueError("hypervolume: 1 to 3 objectives")
    pts.sort(key=lambda p: p[2])
    vol = 0.0
    for i, p in enumerate(pts):
        z_next = pts[i + 1][2] if i + 1 < len(pts) else float(ref[2])
        if z_next > p[2]:
            vol += _hv2d([(q[0], q[1]) for q in pts[:i + 1]], (float(ref[0]), float(ref[1]))) * (z_next - p[2])
    return vol


def shared_reference_point(fronts: Sequence[Sequence[Sequence[float]]], margin: float = 1e-3) -> List[float]:
    allp = [p for f in fronts for p in f]
    return [max(p[m] for p in allp) + margin for m in range(len(allp[0]))]
