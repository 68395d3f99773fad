"""CPU: host NSGA-II operators against the reference-generated goldens (row N1), the loop with a
stand-in evaluator, and the hypervolume metric (row N3) against brute force."""
import json
import os
import random

import numpy as np
import pytest

from cmoop_audio_processing_amd import genes as G, nsga


def test_sort_crowding_dominance_match_reference(golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "nsga_ops_golden.json")))["cases"]
    for c in cases:
        res, lam = c["results"], c["lam"]
        assert nsga.fast_non_dominated_sort(res, lam) == c["fronts"]
        assert [[nsga.dominates(a, b, lam) for b in res] for a in res] == c["dominates"]
        for front, exp in zip(c["fronts"], c["crowding"]):
            got = nsga.crowding_distance(front, res)
            for k, v in exp.items():
                g = got[int(k)]
                assert (g == float("inf")) if v == "inf" else g == pytest.approx(v, abs=1e-15)


def test_get_lambda_matches_reference(golden_dir):
    lam = json.load(open(os.path.join(golden_dir, "codec_golden.json")))["get_lambda"]
    assert [nsga.get_lambda(i, 30) for i in range(30)] == lam["nsga_penalty.py:MAX_GEN=30"]
    assert [nsga.get_lambda(i, 1) for i in range(1)] == lam["sa_nsga_penalty.py:MAX_GEN=1"]      # guarded (quirk Q8)


def test_operators_consume_rng_like_the_reference():
    # crossover: six random() calls in gene-key order; mutate: random() per gene, then choice()/flip
    p1 = {"filters": 16, "kernel_size": 3, "use_bn": True, "residual_blocks": 1, "fc_layers": 1, "use_dropout": False}
    p2 = {"filters": 64, "kernel_size": 5, "use_bn": False, "residual_blocks": 3, "fc_layers": 4, "use_dropout": True}
    r1, r2 = random.Random(5), random.Random(5)
    c1, c2 = nsga.crossover(p1, p2, r1)
    for key in G.GENE_KEYS:
        swap = r2.random() < 0.5
        assert c1[key] == (p2[key] if swap else p1[key]) and c2[key] == (p1[key] if swap else p2[key])
    r1, r2 = random.Random(9), random.Random(9)
    m = nsga.mutate(p1, r1, prob=0.5)
    exp = dict(p1)
    for key, opts in zip(G.GENE_KEYS, G.GENE_OPTIONS):
        if r2.random() < 0.5:
            exp[key] = (not exp[key]) if key in ("use_bn", "use_dropout") else r2.choice(opts)
    assert m == exp and p1["filters"] == 16            # input untouched


def fake_evaluate(pop):
    out = []
    for hp in pop:
        g = G.normalize_hparams(hp)
        size = G.model_size_mb(g, 0, 10)
        acc = 0.80 + 0.02 * g[3] + 0.01 * g[4] + (0.03 if g[2] else 0.0) + 0.0005 * g[0]
        fpr = 0.2 - 0.15 * acc
        cv = max(0.0, 0.9 - acc) + max(0.0, size - 2.5) + max(0.0, fpr - 0.1)
        out.append({"hparams": hp, "objs": [-acc, size, fpr], "CV": cv})
    return out


@pytest.mark.parametrize("pop,gens", [(4, 2), (5, 3), (15, 6), (40, 1)])
def test_nsga2_loop_with_stand_in_evaluator(pop, gens, tmp_path):
    calls = []

    def ev(p):
        calls.append(len(p))
        return fake_evaluate(p)
    pareto, hist = nsga.nsga2(ev, pop, gens, seed=3)
    assert calls == [pop] * (1 + gens)                       # pop * (1 + max_gen) true evaluations
    assert len(hist) == gens and all(len(h) == pop for h in hist)
    assert set(hist[0][0]) == {"Generation", "Accuracy", "Size_MB", "FPR", "CV", *G.GENE_KEYS}
    for ind in pareto:
        assert ind["CV"] == 0
    for a in pareto:
        for b in pareto:
            assert not nsga.dominates(a, b, nsga.LAMBDA_FINAL)
    # same seed -> same run; different seed -> (almost surely) different genes
    pareto2, hist2 = nsga.nsga2(fake_evaluate, pop, gens, seed=3)
    assert hist2 == hist
    nsga.write_records_csv(str(tmp_path / "gens.csv"), hist)
    assert sum(1 for _ in open(tmp_path / "gens.csv")) == 1 + pop * gens


def test_all_infeasible_population_returns_empty_front():
    bad = lambda p: [{"hparams": hp, "objs": [-0.1, 9.0, 0.5], "CV": 7.3} for hp in p]   # noqa: E731
    pareto, hist = nsga.nsga2(bad, 6, 5, seed=0)          # the reference raises IndexError here (quirk Q10)
    assert pareto == [] and len(hist) == 5


def brute_hv(points, ref, n=200000, seed=0):
    rs = np.random.RandomState(seed)
    pts = np.asarray(points, float)
    lo = pts.min(axis=0)
    x = lo + rs.rand(n, len(ref)) * (np.asarray(ref) - lo)
    dom = np.zeros(n, bool)
    for p in pts:
        dom |= np.all(x >= p, axis=1)
    return dom.mean() * np.prod(np.asarray(ref) - lo)


def test_hypervolume_exact_cases_and_monte_carlo():
    assert nsga.hypervolume([[0.0, 0.0, 0.0]], [1, 2, 3]) == pytest.approx(6.0)
    assert nsga.hypervolume([[0, 0, 0], [0.5, 0.5, 0.5]], [1, 1, 1]) == pytest.approx(1.0)      # dominated point adds nothing
    assert nsga.hypervolume([[0.0, 1.0], [1.0, 0.0]], [2, 2]) == pytest.approx(3.0)
    assert nsga.hypervolume([[2.0, 2.0, 2.0]], [1, 1, 1]) == 0.0
    rs = np.random.RandomState(1)
    pts = np.c_[-rs.uniform(0.85, 0.95, 12), rs.uniform(0.05, 2.3, 12), rs.uniform(0.005, 0.02, 12)]
    ref = nsga.shared_reference_point([pts.tolist()])
    hv = nsga.hypervolume(pts.tolist(), ref)
    assert hv == pytest.approx(brute_hv(pts, ref), rel=0.03)
    # permutation invariance
    assert nsga.hypervolume(pts[::-1].tolist(), ref) == pytest.approx(hv, rel=1e-12)
