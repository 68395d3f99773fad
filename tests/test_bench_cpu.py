"""Host-side pieces of bench.py that run without a GPU: the synthetic clip generator, the cpu_baseline leg (the only
place outside tests/ and smoke() that may call the oracle) and the committed bench lines' contract keys."""
import glob
import io
import json
import os
import subprocess
import sys
import time
from contextlib import redirect_stdout

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


def test_plan_steps_fits_whole_generations_into_the_budget():
    # first generation took 72 s and ended 200 s after start; the timed region must end by 425 s
    assert bench.plan_steps(72.0, 200.0, 425.0, 20) == 1 + 2          # 2 more fit (2 * 75.6 = 151 <= 225 < 3 * 75.6)
    assert bench.plan_steps(72.0, 200.0, 425.0, 2) == 2                # never more than requested
    assert bench.plan_steps(400.0, 420.0, 425.0, 20) == 1              # nothing fits: still >= 1 (the one already run)
    assert bench.plan_steps(10.0, 500.0, 425.0, 20) == 1               # already past the deadline
    assert bench.plan_steps(9.0, 100.0, 425.0, 20) == 20               # everything fits: all K
    assert bench.plan_steps(1.0, 0.0, 1e9, 1) == 1


def test_driver_argv_resolves_to_a_bounded_run_with_a_stub_evaluator():
    """The driver's exact flags (--gpus 1 --steps 20 --warmup 5) under a small budget: main()'s budgeting logic with the
    stub evaluator (no GPU, no kernels) times >= 1 whole generation, fewer than requested, and prints ONE JSON line in
    time.  A stub line is marked as such: it is a harness rehearsal, never a measurement."""
    buf = io.StringIO()
    t0 = time.perf_counter()
    with redirect_stdout(buf):
        rc = bench.main(["--gpus", "1", "--steps", "20", "--warmup", "5", "--stub", "--stub-ms-per-gflop", "100",
                         "--budget-s", "6"], t_origin=time.perf_counter())
    wall = time.perf_counter() - t0
    assert rc == 0 and wall < 7.5, wall            # budget 6 s; the slack absorbs sleep jitter of the stub on a loaded host
    lines = [ln for ln in buf.getvalue().splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["steps_requested"] == 20 and 1 <= d["steps"] < 20 and d["warmup"] <= 2 and d["warmup_requested"] == 5
    assert d["n_gpus"] == 1 and d["n_ranks_seen"] == 1 and "stub" in d["data"]
    assert abs(d["value"] - 40 * d["steps"] / (d["ms_per_step"] * d["steps"] / 3.6e6)) < 1e-3 * d["value"]
    assert d["ms_per_step"] * d["steps"] / 1e3 <= wall
    assert d["budget"]["seconds_since_start_at_print"] <= 7.5


def test_gpus_2_starts_two_ranks_itself_and_reports_them():
    """`python bench.py --gpus 2` WITHOUT torchrun must start the two ranks itself (it silently ran one);
    rehearsed with the stub evaluator over gloo.  Both ranks must have evaluated candidates (shared queue)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub", "--steps", "3", "--warmup", "1",
                        "--budget-s", "30", "--stub-ms-per-gflop", "6"], capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["steps"] >= 1
    per_rank = d["config"]["candidates_per_rank_last_step"]
    # 8 workers per rank start with the 8 + 8 items of the deterministic deal; the other 24 go through the shared counter
    assert len(per_rank) == 2 and sum(per_rank) == 40 and min(per_rank) >= 8, per_rank
    # the N > 1 line explains itself -- per-rank candidates, busy seconds, counter round trips, all-gather
    mg = d["multi_gpu"]["per_rank"]
    assert [m["rank"] for m in mg] == [0, 1] and [m["candidates"] for m in mg] == per_rank
    assert all(m["dealt_at_start"] == 8 and m["busy_wall_s"] > 0 and m["all_gather_ms"] > 0 for m in mg)
    assert sum(m["store_fetch_adds"] for m in mg) >= 24 and all(m["store_add_us_mean"] > 0 for m in mg if m["store_fetch_adds"])
    assert d["config"]["protocol"] == "fixed" and d["config"]["epochs_per_candidate"] == 10      # E_fixed = 10 is the default


def test_gpus_8_rehearsal_deals_five_candidates_to_every_rank():
    """The driver's 8-rank launch, rehearsed with the stub evaluator over gloo on the CPU: bench.py starts its 8 ranks,
    every rank runs min(8, ceil(40 / 8)) = 5 workers and gets exactly the 5 candidates of the deterministic deal
    (evaluator.queue_plan) -- no burst of fetch-adds decides who trains the large ones."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--stub", "--steps", "2", "--warmup", "1",
                        "--budget-s", "60", "--stub-ms-per-gflop", "6"], capture_output=True, text=True, timeout=400, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["n_ranks_seen"] == 8 and d["steps"] >= 1
    assert d["config"]["candidates_per_rank_last_step"] == [5] * 8


def test_world_size_mismatch_fails_loudly():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub"], capture_output=True, text=True,
                       timeout=120, env=env)
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr and r.stdout.strip() == ""


def test_spawn_path_counts_gpus_without_touching_hip(monkeypatch):
    """`--gpus N` without torchrun must not initialise HIP in the parent before it starts the ranks --
    torch.cuda.device_count() may fall back to hipGetDeviceCount.  The parent counts devices from the visibility variables
    or the KFD topology only, refuses to spawn under rocprofv3 (whose preloaded library has already initialised the GPU),
    and never calls into torch.cuda."""
    import argparse

    def boom(*a, **k):
        raise AssertionError("the spawn path must not call torch.cuda")
    monkeypatch.setattr(torch.cuda, "device_count", boom)
    monkeypatch.setattr(torch.cuda, "is_available", boom)
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1")
    assert bench.count_visible_gpus() == 2
    args = argparse.Namespace(gpus=4, same_device=False, stub=False)
    assert bench.spawn_ranks(args, ["--gpus", "4"]) == 2                      # 4 wanted, 2 visible: refused before any spawn
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.count_visible_gpus() == 0
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    n = bench.count_visible_gpus()
    assert n is None or n >= 0                                                # sysfs topology (absent in this container -> None or 0)
    monkeypatch.setenv("ROCPROFILER_SOMETHING", "1")
    assert bench.running_under_rocprof()
    assert bench.spawn_ranks(argparse.Namespace(gpus=2, same_device=True, stub=True), ["--gpus", "2", "--stub"]) == 2   # refused under a profiler


def test_world_one_under_torchrun_is_the_plain_n1_path():
    """VERDICT r2 item 10 (SCALE-vs-BENCH cross-check): the driver's N=1 SCALE point may come through torchrun with
    WORLD_SIZE=1; it must be the same code path as plain `--gpus 1` (no process group, no collective): same workload
    description, same schedule fields."""
    outs = []
    for env_extra in ({}, {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"}):
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        env.update(env_extra)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--stub", "--steps", "1", "--warmup", "1",
                            "--budget-s", "20", "--stub-ms-per-gflop", "2"], capture_output=True, text=True, timeout=120, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0]))
    a, b = outs
    assert a["config"] == b["config"] and a["n_gpus"] == b["n_gpus"] == 1 and "multi_gpu" not in a and "multi_gpu" not in b
