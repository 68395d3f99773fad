#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the REFERENCE's own code.

Runs ONLY in the build container (needs /root/reference).  The reference's
scripts cannot be imported (no tensorflow/keras, hard-coded data paths --
SURVEY.md §8c), so this script parses each file with ``ast``, pulls out the
pure-Python/NumPy/sklearn function definitions on the hot path and executes
exactly those definitions in a scratch namespace.  Only INPUTS and OUTPUTS are
written to the JSON fixtures -- no reference source text is copied anywhere.

Fixtures (inputs + expected outputs of the reference functions):
  fpr_golden.json         calculate_fpr V1 (nsga_penalty.py:351-364), vectorised V1
                          (init_sa_nsga_local.py:137-143), V3 (sa_nsga_local.py:138-141),
                          and the y_true quirk of nsga_penalty.py:387
  objectives_golden.json  compute_objectives_and_constraints (nsga_penalty.py:418-442,
                          sa_nsga_penalty.py:231-253, acc_fpr_nsga_1.py:283-310) driven by a
                          stubbed evaluate_individual; compute_model_size_mb arithmetic
  codec_golden.json       hparams_to_vector / vector_to_hparams (mobo_penalty.py:305-338),
                          get_lambda (nsga_penalty.py:217-219, sa_nsga_penalty.py:130-132)
  nsga_ops_golden.json    dominates / fast_non_dominated_sort / crowding_distance
                          (nsga_penalty.py:448-524) -- for the host-loop row N1 (next)
  metrics_golden.json     GD / IGD / Spread / C-metric (compare.ipynb) and the Tchebycheff score/rank
                          ("Tchebycheff s_rank.ipynb") -- row N3
  surrogate_golden.json   SurrogateManager (sa_nsga_local.py:169-234) + select_infill_points
                          (sa_nsga_penalty.py:472-518) under a fixed numpy seed -- row N2
  prepare_dataset_golden.json  load_data + prepare_dataset of nsga_penalty.py:57-155 (StandardScaler RE-FIT on every
                          split, quirk Q1) and of mobo_penalty.py:32-82 (fit on train only), executed on six small .npy
                          files written to a temp dir with the installed scikit-learn: inputs, output arrays, shapes,
                          dtypes -- rows a1 / a2
"""
import ast
import json
import os
import random
import sys

import numpy as np
from sklearn.metrics import confusion_matrix

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def extract(path, names, extra_globals=None, assigns=()):
    """exec the named FunctionDefs (and simple top-level assignments) of one file."""
    src = open(os.path.join(REF, path)).read()
    tree = ast.parse(src)
    body = []
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
            body.append(node)
        elif isinstance(node, ast.Assign) and len(node.targets) == 1 and \
                isinstance(node.targets[0], ast.Name) and node.targets[0].id in assigns:
            body.append(node)
    found = {n.name for n in body if isinstance(n, (ast.FunctionDef, ast.ClassDef))}
    missing = set(names) - found
    if missing:
        raise SystemExit(f"{path}: functions not found: {missing}")
    ns = {"np": np, "confusion_matrix": confusion_matrix, "random": random, "print": lambda *a, **k: None}
    ns.update(extra_globals or {})
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return ns


def jf(x):
    return float(x)


def gen_fpr():
    v1 = extract("nsga_penalty.py", ["calculate_fpr"])["calculate_fpr"]
    v1b = extract("ablation_study/init_sa_nsga_local.py", ["calculate_fpr"])["calculate_fpr"]
    v3 = extract("ablation_study/sa_nsga_local.py", ["calculate_fpr"])["calculate_fpr"]
    rs = np.random.RandomState(20250704)
    cases = []

    def add(tag, y_true, y_pred, C):
        y_true = np.asarray(y_true, dtype=np.int64)
        y_pred = np.asarray(y_pred, dtype=np.int64)
        # the quirk: y_true = np.argmax(y_validation, axis=1) on an (N,1) array  (nsga_penalty.py:387)
        yq = np.argmax(y_true[:, None], axis=1)
        cases.append({
            "tag": tag, "C": int(C), "y_true": y_true.tolist(), "y_pred": y_pred.tolist(),
            "v1": jf(v1(y_true, y_pred, C)), "v1_vectorised": jf(v1b(y_true, y_pred, C)),
            "v3": jf(v3(y_true, y_pred, C)), "v1_quirk": jf(v1(yq, y_pred, C)),
        })

    for C in (10, 11, 35):
        for n in (1, 7, 64, 500):
            yt = rs.randint(0, C, size=n)
            yp = np.where(rs.rand(n) < 0.8, yt, rs.randint(0, C, size=n))
            add(f"random_C{C}_n{n}", yt, yp, C)
        yt = rs.randint(0, C, size=200)
        add(f"perfect_C{C}", yt, yt, C)
        add(f"all_pred_zero_C{C}", yt, np.zeros(200, int), C)
        add(f"single_true_class_C{C}", np.full(50, 3), rs.randint(0, C, size=50), C)   # FP+TN==0 rows / V3 drop
        add(f"missing_classes_C{C}", rs.randint(0, 3, size=90), rs.randint(0, 4, size=90), C)
    add("balanced_C10", np.repeat(np.arange(10), 30), np.tile(np.arange(10), 30), 10)
    json.dump({"source": "nsga_penalty.py:351-364; init_sa_nsga_local.py:137-143; sa_nsga_local.py:138-141; nsga_penalty.py:387",
               "cases": cases}, open(os.path.join(OUT, "fpr_golden.json"), "w"))
    print("fpr_golden.json", len(cases), "cases")


def gen_objectives():
    rs = np.random.RandomState(7)
    out = {"cases": []}
    scripts = [
        ("nsga_penalty.py", dict(MIN_ACCURACY=0.9, MAX_MODEL_SIZE=2.5, MAX_FPR=0.1)),
        ("sa_nsga_penalty.py", dict(MIN_ACCURACY=0.75, MAX_MODEL_SIZE=2.5, MAX_FPR=0.09)),
        ("mobo_penalty.py", dict(MIN_ACCURACY=0.90, MAX_MODEL_SIZE=2.5, MAX_FPR=0.09)),
        ("ablation_study/acc_fpr_nsga_1.py", dict(MIN_ACCURACY=0.90, MAX_MODEL_SIZE=2.5, MAX_FPR=0.09)),
        ("ablation_study/acc_size_nsga_1.py", dict(MIN_ACCURACY=0.90, MAX_MODEL_SIZE=2.5, MAX_FPR=0.09)),
        ("ablation_study/size_fpr_nsga_1.py", dict(MIN_ACCURACY=0.90, MAX_MODEL_SIZE=2.5, MAX_FPR=0.09)),
    ]
    triples = [(float(rs.uniform(0.5, 0.99)), float(rs.uniform(0.03, 6.0)), float(rs.uniform(0.0, 0.2))) for _ in range(12)]
    triples += [(0.9, 2.5, 0.1), (0.75, 2.5, 0.09), (0.0, 51.97, 1.0)]
    for path, thr in scripts:
        if path == "mobo_penalty.py":
            continue  # run_mobo inlines the assembly; thresholds recorded for the preset table only
        calls = iter(triples)
        ns = extract(path, ["compute_objectives_and_constraints"],
                     dict(thr, evaluate_individual=lambda hp: next(calls)))
        pop = [{"id": i} for i in range(len(triples))]
        res = ns["compute_objectives_and_constraints"](pop)
        recs = []
        for t, r in zip(triples, res):
            rec = {"acc": t[0], "size_mb": t[1], "fpr": t[2], "objs": [jf(v) for v in r["objs"]], "CV": jf(r["CV"])}
            for extra in ("size_metric", "fpr_metric", "acc_metric"):
                if extra in r:
                    rec[extra] = jf(r[extra])
            assert r["hparams"] is pop[len(recs)]          # the result holds a REFERENCE to the caller's dict
            recs.append(rec)
        out["cases"].append({"script": path, "thresholds": thr, "records": recs})

    class _Stub:
        def __init__(self, n):
            self.n = n

        def count_params(self):
            return self.n
    size = extract("nsga_penalty.py", ["compute_model_size_mb"])["compute_model_size_mb"]
    counts = [19674, 20058, 20123, 21683, 324074, 13624714, 13626339, 4890634, 880106, 8106, 8298, 129418, 4915914, 1]
    out["size_mb"] = [{"params": c, "size_mb": jf(size(_Stub(c)))} for c in counts]
    json.dump(out, open(os.path.join(OUT, "objectives_golden.json"), "w"))
    print("objectives_golden.json", sum(len(c["records"]) for c in out["cases"]), "records")


def gen_codec():
    opts = ["FILTER_OPTIONS", "KERNEL_SIZE_OPTIONS", "USE_BN_OPTIONS", "RESIDUAL_BLOCK_OPTIONS",
            "FC_LAYER_OPTIONS", "USE_DROPOUT_OPTIONS"]
    ns = extract("mobo_penalty.py", ["hparams_to_vector", "vector_to_hparams"], assigns=opts)
    rs = np.random.RandomState(11)
    enc, dec = [], []
    for f in ns["FILTER_OPTIONS"]:
        for k in ns["KERNEL_SIZE_OPTIONS"]:
            for bn in ns["USE_BN_OPTIONS"]:
                for r in ns["RESIDUAL_BLOCK_OPTIONS"]:
                    for fc in ns["FC_LAYER_OPTIONS"]:
                        for dr in ns["USE_DROPOUT_OPTIONS"]:
                            hp = {"filters": f, "kernel_size": k, "use_bn": bn, "residual_blocks": r,
                                  "fc_layers": fc, "use_dropout": dr}
                            enc.append({"hparams": hp, "vector": [jf(v) for v in ns["hparams_to_vector"](hp)]})
    vecs = rs.rand(64, 6).tolist() + [[0.5] * 6, [0.25] * 6, [0.75] * 6, [1 / 6.0] * 6, [0.0] * 6, [1.0] * 6]
    for v in vecs:
        dec.append({"vector": v, "hparams": ns["vector_to_hparams"](np.asarray(v))})
    lam = {}
    for path, mg in (("nsga_penalty.py", 30), ("sa_nsga_penalty.py", 30), ("sa_nsga_penalty.py", 1)):
        g = extract(path, ["get_lambda"], dict(MAX_GEN=mg, LAMBDA_INITIAL=1.0, LAMBDA_FINAL=50.0))["get_lambda"]
        lam[f"{path}:MAX_GEN={mg}"] = [jf(g(i)) for i in range(mg)]
    json.dump({"encode": enc, "decode": dec, "get_lambda": lam}, open(os.path.join(OUT, "codec_golden.json"), "w"))
    print("codec_golden.json", len(enc), "encode,", len(dec), "decode")


def gen_nsga_ops():
    ns = extract("nsga_penalty.py", ["dominates", "fast_non_dominated_sort", "crowding_distance"], dict(EPSILON=1e-6))
    rs = np.random.RandomState(3)
    cases = []
    for n in (1, 2, 5, 15, 40):
        for lam in (1.0, 25.5, 50.0):
            res = [{"objs": [-float(rs.uniform(0.5, 0.99)), float(rs.choice([0.08, 0.5, 1.2, 3.0, 7.5])), float(rs.uniform(0, 0.15))],
                    "CV": float(max(0.0, rs.uniform(-0.2, 0.3)))} for _ in range(n)]
            fronts = ns["fast_non_dominated_sort"](res, lam)
            crowd = [{str(k): (v if np.isfinite(v) else "inf") for k, v in ns["crowding_distance"](fr, res).items()} for fr in fronts]
            dom = [[bool(ns["dominates"](a, b, lam)) for b in res] for a in res]
            cases.append({"lam": lam, "results": res, "fronts": fronts, "crowding": crowd, "dominates": dom})
    json.dump({"source": "nsga_penalty.py:448-524", "cases": cases}, open(os.path.join(OUT, "nsga_ops_golden.json"), "w"))
    print("nsga_ops_golden.json", len(cases), "cases")


def gen_surrogate():
    """SurrogateManager (sa_nsga_local.py:169-234, the return_std variant) and select_infill_points
    (sa_nsga_penalty.py:472-518) under a fixed numpy seed (the GPs draw their optimiser restarts from
    numpy's global RNG)."""
    import pandas as pd
    from copy import deepcopy
    from sklearn.compose import ColumnTransformer
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import ConstantKernel as C, Matern, WhiteKernel
    from sklearn.preprocessing import OneHotEncoder, StandardScaler
    env = dict(pd=pd, deepcopy=deepcopy, ColumnTransformer=ColumnTransformer, GaussianProcessRegressor=GaussianProcessRegressor,
               C=C, Matern=Matern, WhiteKernel=WhiteKernel, OneHotEncoder=OneHotEncoder, StandardScaler=StandardScaler, EPSILON=1e-6)
    SM = extract("ablation_study/sa_nsga_local.py", ["SurrogateManager"], env)["SurrogateManager"]
    infill = extract("sa_nsga_penalty.py", ["select_infill_points"], dict(EPSILON=1e-6))["select_infill_points"]
    rnd = random.Random(42)

    def hp():
        return {"filters": rnd.choice([16, 32, 64]), "kernel_size": rnd.choice([3, 5]), "use_bn": rnd.choice([True, False]),
                "residual_blocks": rnd.choice([1, 2, 3]), "fc_layers": rnd.choice([1, 2, 3, 4]), "use_dropout": rnd.choice([True, False])}

    def res(h):
        acc = 0.7 + 0.03 * h["residual_blocks"] + 0.01 * h["fc_layers"] + (0.04 if h["use_bn"] else 0) + 0.0007 * h["filters"] + rnd.uniform(-0.02, 0.02)
        size = 0.05 * h["filters"] / 16 * h["kernel_size"] ** 2 / 9 * 2 ** h["residual_blocks"]
        fpr = 0.2 - 0.15 * acc + rnd.uniform(0, 0.01)
        cv = max(0.0, 0.9 - acc) + max(0.0, size - 2.5) + max(0.0, fpr - 0.09)
        return {"hparams": h, "objs": [-acc, size, fpr], "CV": cv}
    train1 = [hp() for _ in range(12)]
    res1 = [res(h) for h in train1]
    train2 = [hp() for _ in range(2)] + [dict(train1[3])]            # one duplicate genotype: keep='last'
    res2 = [res(h) for h in train2]
    query = [hp() for _ in range(16)]
    sm = SM()
    np.random.seed(123)
    sm.update(train1, res1)
    p1, s1 = sm.predict(query, return_std=True)
    struct1 = sm.predict_and_structure(query)
    idx, _ = infill(struct1, 5)
    np.random.seed(124)
    sm.update(train2, res2)
    p2, s2 = sm.predict(query, return_std=True)

    def tolist(d):
        return {k: [float(x) for x in v] for k, v in d.items()}
    out = {"train1": train1, "res1": [{"objs": r["objs"], "CV": r["CV"]} for r in res1],
           "train2": train2, "res2": [{"objs": r["objs"], "CV": r["CV"]} for r in res2], "query": query,
           "seed1": 123, "seed2": 124, "pred1": tolist(p1), "std1": tolist(s1), "pred2": tolist(p2), "std2": tolist(s2),
           "structured1": [{"objs": [float(v) for v in r["objs"]], "CV": float(r["CV"])} for r in struct1],
           "infill_indices_top5": [int(i) for i in idx], "n_training_rows_after_update2": int(len(sm.training_data))}
    # select_infill_points on hand-made predicted sets (feasible-first, normalised-sum score, CV ranking)
    rs = np.random.RandomState(8)
    sel_cases = []
    for n, k in ((6, 2), (15, 3), (15, 20), (4, 1)):
        pred = [{"hparams": {"id": i}, "objs": [float(-rs.uniform(0.6, 0.95)), float(rs.uniform(0.05, 3)), float(rs.uniform(0, 0.2))],
                 "CV": float(rs.choice([0.0, 0.0, rs.uniform(0, 0.4)]))} for i in range(n)]
        ii, hh = infill(pred, k)
        sel_cases.append({"predicted": pred, "k": k, "indices": [int(i) for i in ii]})
    out["select_cases"] = sel_cases
    json.dump(out, open(os.path.join(OUT, "surrogate_golden.json"), "w"))
    print("surrogate_golden.json", len(query), "queries,", len(sel_cases), "selection cases")


def gen_metrics():
    """GD / IGD / Spread / C-metric / true front from compare.ipynb (cell 0) and the Tchebycheff score from
    'Tchebycheff s_rank.ipynb': the notebooks' FunctionDefs only, executed on random fronts."""
    from scipy.spatial.distance import cdist
    import pandas as pd

    def nb_funcs(path, names):
        nb = json.load(open(os.path.join(REF, path)))
        body = []
        for c in nb["cells"]:
            if c["cell_type"] != "code":
                continue
            src = "".join(c["source"])
            try:
                tree = ast.parse(src)
            except SyntaxError:
                continue
            body += [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
        ns = {"np": np, "cdist": cdist}
        exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
        return ns
    f = nb_funcs("compare.ipynb", ["dominates_min", "generational_distance", "inverted_gd", "spread_metric", "coverage_metric"])
    t = nb_funcs("Tchebycheff s_rank.ipynb", ["tchebycheff_score"])
    rs = np.random.RandomState(21)
    cases = []
    for na, nb_ in ((5, 7), (12, 3), (1, 4), (9, 9)):
        A = np.c_[-rs.uniform(0.85, 0.95, na), rs.uniform(0.05, 2.4, na), rs.uniform(0.005, 0.02, na)]
        B = np.c_[-rs.uniform(0.85, 0.95, nb_), rs.uniform(0.05, 2.4, nb_), rs.uniform(0.005, 0.02, nb_)]
        allp = np.vstack([A, B])
        mask = np.ones(len(allp), bool)
        for i in range(len(allp)):
            for j in range(len(allp)):
                if i != j and f["dominates_min"](allp[j], allp[i]):
                    mask[i] = False
                    break
        tf = allp[mask]
        sp = f["spread_metric"](A, tf)
        cases.append({"A": A.tolist(), "B": B.tolist(), "true_front": tf.tolist(),
                      "gd": jf(f["generational_distance"](A, tf)), "igd": jf(f["inverted_gd"](A, tf)),
                      "spread": None if np.isnan(sp) else jf(sp), "c_ab": jf(f["coverage_metric"](A, B)),
                      "c_ba": jf(f["coverage_metric"](B, A))})
    acc, size, fpr = rs.uniform(0.85, 0.95, 10), rs.uniform(0.05, 2.4, 10), rs.uniform(0.005, 0.02, 10)
    acc[3], size[3], fpr[3] = acc[7], size[7], fpr[7]                       # a tie
    F = np.column_stack([1.0 - acc, size, fpr])
    sc = t["tchebycheff_score"](F, F.min(axis=0), np.ones(3) / 3.0)
    ranks = pd.Series(sc).rank(method="min", ascending=True).astype(int).tolist()
    json.dump({"cases": cases, "tcheby": {"acc": acc.tolist(), "size": size.tolist(), "fpr": fpr.tolist(),
                                          "scores": [jf(v) for v in sc], "ranks": ranks}},
              open(os.path.join(OUT, "metrics_golden.json"), "w"))
    print("metrics_golden.json", len(cases), "cases")


def gen_prepare_dataset():
    """The reference's OWN load_data + prepare_dataset (NumPy + sklearn.preprocessing.StandardScaler, both importable
    here) run on six .npy files of the layout nsga_penalty.py:64-71 reads.  float32 and float64 feature files: the
    reference stores what StandardScaler returns (same dtype as its input for floating input)."""
    import tempfile
    from sklearn.preprocessing import StandardScaler
    rs = np.random.RandomState(85155)
    cases = []
    for dtype in ("float32", "float64"):
        n, T, F = {"train": 7, "val": 5, "test": 4}, 6, 8
        raw = {k: (2.5 + (1.0 + 0.3 * np.arange(F)) * rs.randn(n[k], T, F) + (0.8 if k == "val" else 0.0)).astype(dtype) for k in n}
        raw["train"][:, :, 3] = 1.25                    # a constant feature: StandardScaler's zero-variance -> scale 1 handling
        lab = {k: rs.randint(0, 10, size=n[k]).astype(np.int64) for k in n}
        with tempfile.TemporaryDirectory() as d:
            for k, fn in (("train", "train"), ("val", "val"), ("test", "test")):
                np.save(os.path.join(d, f"X_{fn}.npy"), raw[k])
                np.save(os.path.join(d, f"y_{fn}.npy"), lab[k])
            rec = {"dtype": dtype, "T": T, "F": F,
                   "inputs": {f"X_{k}": raw[k].astype(np.float64).tolist() for k in n} | {f"y_{k}": lab[k].tolist() for k in n}}
            for script, tag in (("nsga_penalty.py", "refit"), ("mobo_penalty.py", "train_only")):
                ns = extract(script, ["load_data", "prepare_dataset"], dict(StandardScaler=StandardScaler))
                X_train, y_train, X_val, y_val, X_test, y_test = ns["prepare_dataset"](d)
                rec[tag] = {"source": script,
                            "X_train": np.asarray(X_train, np.float64).tolist(), "X_val": np.asarray(X_val, np.float64).tolist(),
                            "X_test": np.asarray(X_test, np.float64).tolist(),
                            "y_train": np.asarray(y_train).tolist(), "y_val": np.asarray(y_val).tolist(), "y_test": np.asarray(y_test).tolist(),
                            "X_shape": list(X_train.shape), "y_shape": list(y_train.shape), "X_dtype": str(X_train.dtype), "y_dtype": str(y_train.dtype)}
            cases.append(rec)
    json.dump({"source": "nsga_penalty.py:57-155 (load_data, prepare_dataset: fit_transform per split); mobo_penalty.py:32-82 (fit on train only); "
                         "executed with the installed scikit-learn's StandardScaler on six temp .npy files",
               "cases": cases}, open(os.path.join(OUT, "prepare_dataset_golden.json"), "w"))
    print("prepare_dataset_golden.json", len(cases), "cases")


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (build container only)")
    gen_fpr()
    gen_objectives()
    gen_codec()
    gen_nsga_ops()
    gen_surrogate()
    gen_metrics()
    gen_prepare_dataset()
