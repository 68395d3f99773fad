"""Host-side pieces of bench.py that run without a GPU: the synthetic clip generator, the cpu_baseline leg (the only
place outside tests/ and smoke() that may call the oracle) and the committed bench lines' contract keys."""
import glob
import json
import os

import numpy as np
import torch

import bench
from cmoop_audio_processing_amd import genes as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_synthetic_clips_are_seeded_balanced_and_finite():
    w1, y1 = bench.synth_waveforms(40, 10, 1234, torch.device("cpu"), n_samples=1600, chunk=16)
    w2, y2 = bench.synth_waveforms(40, 10, 1234, torch.device("cpu"), n_samples=1600, chunk=16)
    assert w1.shape == (40, 1600) and y1.dtype == torch.int32
    assert torch.equal(w1, w2) and torch.equal(y1, y2)                      # seeded
    assert np.bincount(y1.numpy(), minlength=10).tolist() == [4] * 10       # class-balanced (10 x 3000 in the real run)
    assert torch.isfinite(w1).all() and 0.01 < float(w1.abs().mean()) < 2.0


def test_cpu_baseline_leg_reports_the_contract_fields():
    pop = [(16, 3, 0, 1, 1, 0), (16, 3, 1, 1, 2, 1), (32, 3, 0, 1, 1, 0)]
    rs = np.random.RandomState(0)
    xs, ys = rs.randn(64, 21, 12).astype(np.float32), rs.randint(0, 10, 64)
    cb = bench.cpu_baseline(pop, G.VARIANT_A, 10, 21, 12, n_train=640, n_val=64, epochs=2, X_sample=xs, y_sample=ys, budget_s=2.0)
    assert set(cb) == {"value", "unit", "cores", "kind", "sample"}
    assert cb["kind"] == "port" and cb["unit"] == "candidate-evals/hour" and cb["cores"] >= 1 and cb["value"] > 0
    assert "extrapolated" in cb["sample"]


def test_committed_bench_lines_keep_the_driver_contract():
    need = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline"}
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_line*.json")))
    assert lines, "no committed bench lines under profiles/"
    for f in lines:
        d = json.load(open(f))
        assert need <= set(d), (f, need - set(d))
        r = d["roofline"]
        assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(r), f
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and d["vs_baseline"] is None and "workload" in d["config"]
        assert d["unit"] == "candidate-evals/hour" and d["higher_is_better"] is True
