#!/usr/bin/env python3
"""bench.py -- candidate-net evals/hour of the population-fitness hot path on MI355X.

One "step" = one pass of the hot path over one generation:
``compute_objectives_and_constraints(population)`` for pop=40 candidates
(BASELINE.json configs[1]: "pop=40 gen=1 fitness eval on 1xMI355X: HIP MFCC +
tiny-CNN train, GSC 10-keyword"; reference call site nsga_penalty.py:613/670,
the loop itself :418-442).  Inputs are synthetic 1 s @ 16 kHz clips (SURVEY.md
§8d), turned into standardised [N,101,40] log-mel features by the HIP front end
BEFORE the timed region, so the timed region starts with features resident in
HBM.  Every candidate trains for a fixed epoch budget -- E = 10 epochs with early
stopping off, SURVEY.md section 8d's throughput protocol -- so CPU and GPU do
identical algorithmic work, then runs the inference/confusion readout.
``--protocol reference`` runs the protocol the reference really runs instead
(EarlyStopping(val_loss, patience 5), <= 300 epochs, nsga_penalty.py:377-384) on
the hard synthetic set and reports the epochs distribution next to evals/hour.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Wall-clock budget.  One generation of the headline workload (pop 40, N=30 000,
101x40, fp32, E=10) takes about five minutes on one MI355X, so K and W are UPPER BOUNDS
under ``--budget-s`` (seconds from process start by which the JSON line must be
out; default 450 for the driver's 600 s limit):
* warm-up = up to W light untimed passes: the whole population for one epoch on
  a few hundred clips -- loads every kernel family and fills the buffer cache;
* the timed region holds as many WHOLE generations as fit (>= 1, <= K): the
  first one is timed, then ``plan_steps`` decides how many more fit.  The line
  reports ``steps`` = generations actually timed and ``steps_requested`` = K;
* the CPU-baseline leg runs before the timed region on a bounded sample
  (``--cpu-baseline-s``) and the isolated-kernel leg is a few launches, so
  ``roofline`` and ``cpu_baseline`` are always in the line.

N > 1: ``--gpus N`` without a torchrun environment starts the N ranks itself
(child ``python -m torch.distributed.run``, before any GPU call in the parent)
and relays rank 0's JSON line.  Every rank holds the same seeded population;
the candidates are drained from ONE longest-first queue (``--schedule
dynamic``) or LPT buckets (``static``) and the objective vectors are exchanged
with one RCCL all_gather per step.  The population is fixed at --pop (strong
scaling: the reference shards ONE generation); --weak multiplies it by N.

Rank 0 prints ONE JSON line (contract in the task statement), including
``roofline`` and, at N=1, ``cpu_baseline`` (the torch-CPU oracle timed on this
box's host cores on a bounded sample).  ``roofline.achieved`` is a WORK RATE: the
dominant MFMA kernel's algorithmic FLOPs / its HIP-event duration with the
heaviest candidate of the population training ALONE (one launching thread, no CU
sharing) inside this run, right after the timed region; the same kernel's in-job
per-launch figure is kept as ``in_job_stretched_by_cu_sharing`` (begin->end
durations of co-scheduled kernels overlap, so their sum exceeds the step), next
to ``aggregate_timed_region`` (all executed FLOPs / wall / GPUs).
"""
import argparse
import ctypes as C
import json
import os
import random
import socket
import subprocess
import sys
import threading
import time

T_PROCESS_START = time.perf_counter()

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
# opt-in modes: the bf16 MFMA runs at 16x the fp32 MFMA rate (same guide); bf16x3 spends six bf16 MFMAs per product
PEAK_BY_MODE = {"fp32": PEAK_FP32_MFMA_TFLOPS, "bf16x3": round(16 * PEAK_FP32_MFMA_TFLOPS / 6, 1), "bf16": 16 * PEAK_FP32_MFMA_TFLOPS}

#: PMC traffic of the dominant kernel instantiations on their heaviest layer shape, from the committed separate
#: `rocprofv3 --pmc` passes (FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md); bench.py cannot collect
#: counters itself (they need their own profiler passes), so the line cites the file the numbers come from.
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic_per_kernel.json")


def synth_waveforms(n, classes, seed, device, n_samples=16000, chunk=2000, hard=False, hard_snr_db=-17.0):
    """per class 3 sinusoids log-spaced in 200-4000 Hz with random phase, amplitude
    U(0.1,1), plus N(0,1)-shaped noise at 0 dB SNR.  Seeded torch generator (Philox on GPU).

    hard=True (hypervolume runs only, never the throughput metric): ``hard_snr_db`` SNR (default -17 dB: a per-frame FFT
    plus 101 frames of averaging still recover the partials, but only partly and better for larger nets) and neighbouring
    classes share two of their three partials, so accuracies spread like the reference's published Pareto range
    (0.88-0.931) instead of saturating at 1.0."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    y = (torch.arange(n, device=device) % classes).to(torch.int32)
    perm = torch.randperm(n, generator=g, device=device)
    y = y[perm].contiguous()
    if hard:
        base = torch.logspace(np.log10(200.0), np.log10(4000.0), classes + 2, device=device)
        freqs = torch.stack([base[c:c + 3] for c in range(classes)])            # class c and c+1 share two partials
    else:
        freqs = torch.logspace(np.log10(200.0), np.log10(4000.0), classes * 3, device=device).reshape(classes, 3)
    noise_gain = float(10.0 ** (-hard_snr_db / 20.0)) if hard else 1.0         # hard_snr_db vs 0 dB
    t = torch.arange(n_samples, device=device, dtype=torch.float32) / 16000.0
    wav = torch.empty((n, n_samples), dtype=torch.float32, device=device)
    for s in range(0, n, chunk):
        yy = y[s:s + chunk].long()
        f = freqs[yy]                                                        # [c,3]
        ph = torch.rand((len(yy), 3), generator=g, device=device) * (2 * np.pi)
        sig = torch.sin(2 * np.pi * f[:, :, None] * t[None, None, :] + ph[:, :, None]).sum(1)
        sig = sig / sig.pow(2).mean(dim=1, keepdim=True).sqrt()
        noise = torch.randn((len(yy), n_samples), generator=g, device=device) * noise_gain
        amp = 0.1 + 0.9 * torch.rand((len(yy), 1), generator=g, device=device)
        wav[s:s + chunk] = amp * (sig + noise) * 0.5
    return wav, y


def usable_cpus():
    """CPUs this process may really use: min(os.cpu_count, affinity, cgroup quota) -- a GPU box may show 128 cores and
    grant the job a 16-CPU share; the CPU baseline reports and uses THIS number of threads."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(pop, variant, classes, T, F, n_train, n_val, epochs, X_sample, y_sample, budget_s=15.0):
    """see cpu_baseline_rate; ``epochs``: one int (fixed protocol) or the epochs each candidate ran (reference protocol)"""
    rate, cores, sample = cpu_baseline_rate(pop, variant, classes, T, F, X_sample, y_sample, budget_s)
    return cpu_baseline_line(pop, variant, classes, T, F, n_train, n_val, epochs, rate, cores, sample)


def cpu_baseline_line(pop, variant, classes, T, F, n_train, n_val, epochs, rate, cores, sample):
    from cmoop_audio_processing_amd import genes as G
    fl = np.array([G.fwd_flops_per_sample(g, variant, classes, T, F) for g in pop], dtype=np.float64)
    ep = np.full(len(pop), epochs, dtype=np.float64) if np.isscalar(epochs) else np.asarray(epochs, dtype=np.float64)
    per_eval = fl * (3 * n_train * ep + n_val * ep + n_val)
    hours = per_eval.sum() / rate / 3600.0
    e_txt = f"E={int(epochs)}" if np.isscalar(epochs) else f"the epochs each candidate ran on the GPU (mean {ep.mean():.1f})"
    return {"value": len(pop) / hours, "unit": "candidate-evals/hour", "cores": int(cores), "kind": "port",
            "sample": "; ".join(sample) + f"; extrapolated by closed-form FLOPs to pop={len(pop)}, "
                      f"N_train={n_train}, {e_txt} ({rate / 1e9:.0f} GFLOP/s sustained)"}


def cpu_baseline_rate(pop, variant, classes, T, F, X_sample, y_sample, budget_s=15.0):
    """Time the oracle (torch-CPU restatement of the reference path) on this box's host cores
    on a bounded sample: train steps of batch 64 + inference on 64 rows for the cheapest,
    median and most expensive candidate of the population; extrapolate by closed-form FLOPs.
    Bounded by ``budget_s`` of wall-clock (a third per candidate; at least one train step each).
    -> (algorithmic FLOP/s the oracle sustains, threads used, sample description)"""
    import torch
    from cmoop_audio_processing_amd import genes as G
    from oracle import net as ON
    cores = usable_cpus()
    torch.set_num_threads(cores)
    fl = [G.fwd_flops_per_sample(g, variant, classes, T, F) for g in pop]
    order = np.argsort(fl)
    picks = [int(order[0]), int(order[len(order) // 2]), int(order[-1])]
    cfg = ON.OracleConfig(variant=variant, classes=classes, batch=64)
    done_flops, spent, sample = 0.0, 0.0, []
    t_leg = time.perf_counter()
    for n_pick, i in enumerate(picks):
        left = budget_s - (time.perf_counter() - t_leg)
        share = max(left / (len(picks) - n_pick), 0.0)
        net = ON.OracleNet(pop[i], cfg, 1)
        xb, yb = X_sample[:64], y_sample[:64]
        tw = time.perf_counter()
        net.train_step(xb, yb)                      # warm-up (allocator, MKLDNN primitives); counted if it is all we can afford
        t_warm = time.perf_counter() - tw
        steps, t0 = 0, time.perf_counter()
        while steps < 24 and (time.perf_counter() - t0) + 1.5 * t_warm < share * 0.8:
            net.train_step(xb, yb)
            steps += 1
        if steps == 0:                              # budget too small for a second step: use the warm-up step itself
            steps, t_train = 1, t_warm
        else:
            t_train = time.perf_counter() - t0
        t1 = time.perf_counter()
        net.evaluate(xb, yb)
        t_eval = time.perf_counter() - t1
        done_flops += fl[i] * 64 * (3 * steps + 1)
        spent += t_train + t_eval
        sample.append(f"gene{tuple(pop[i])}:{steps} train steps+1 eval batch in {t_train + t_eval:.1f}s")
    rate = done_flops / spent                                        # algorithmic FLOP/s the oracle sustains
    return rate, cores, sample


def plan_steps(first_step_s, now_s, deadline_s, steps_requested, safety=1.05):
    """How many whole generations the timed region holds in total (>= 1, <= steps_requested), given that the first
    took ``first_step_s`` and ended at ``now_s``; later ones are assumed to take ``safety`` x as long and must
    finish by ``deadline_s`` (all on one clock)."""
    if steps_requested <= 1:
        return 1
    per = max(first_step_s * safety, 1e-9)
    more = int(max(0.0, deadline_s - now_s) // per)
    return 1 + max(0, min(steps_requested - 1, more))


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def count_visible_gpus():
    """GPUs this process would see, WITHOUT touching HIP / HSA (torch.cuda.device_count() falls back to
    hipGetDeviceCount when the amdsmi probe fails, which initialises the runtime in a parent that then starts
    torch.distributed.run): the visibility variables if set, else the KFD topology (nodes with SIMDs are GPUs).
    None when neither source exists (the ranks then report a shortfall themselves)."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([t for t in v.split(",") if t.strip() != ""])
    nodes = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for d in os.listdir(nodes):
            try:
                props = dict(ln.split()[:2] for ln in open(os.path.join(nodes, d, "properties")) if len(ln.split()) >= 2)
            except OSError:
                continue
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        return n
    except OSError:
        return None


def running_under_rocprof():
    return any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")


def spawn_ranks(args, argv):
    """--gpus N without a torchrun environment: start the N ranks as a CHILD process tree (never an exec of this
    process, and before anything here has touched the GPU), relay rank 0's JSON line, exit with the child's code."""
    if running_under_rocprof():
        # the profiler's preloaded library has already initialised the GPU in THIS process: starting
        # torch.distributed.run from here is the exec hop the pool forbids
        print("bench.py: --gpus N under rocprofv3 is refused (the preloaded tool has initialised the GPU; profile one rank: --gpus 1)",
              file=sys.stderr)
        return 2
    if not (args.same_device or args.stub):
        have = count_visible_gpus()                 # sysfs / environment only: nothing here may touch HIP before the spawn
        if have is not None and have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible", file=sys.stderr)
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", CMOOP_BENCH_T0_OFFSET=str(time.perf_counter() - T_PROCESS_START))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for ln in r.stdout.splitlines():
        s = ln.strip()
        if s.startswith("{") and '"metric"' in s:
            line = s
    if line is not None:
        print(line, flush=True)
    elif r.stdout:
        sys.stderr.write(r.stdout[-4000:])
    if r.returncode == 0 and line is None:
        print("bench.py: the ranks exited 0 but printed no JSON line", file=sys.stderr)
        return 3
    return r.returncode


class StubEvaluator:
    """CPU rehearsal of the HARNESS logic only (tests; ``--stub``): no GPU, no kernels, no oracle.  A candidate
    'costs' --stub-ms-per-gflop of sleep per closed-form forward GFLOP/sample; results are placeholders.  Lines
    produced with it carry "data": "stub" and are never a measurement."""

    def __init__(self, cfg, variant, classes, T, F, ms_per_gflop):
        from cmoop_audio_processing_amd import evaluator as E
        self.E, self.config, self.variant, self.classes, self.T, self.F = E, cfg, variant, classes, T, F
        self.ms_per_gflop = ms_per_gflop
        self.slots = int(getattr(cfg, "n_slots", 8))
        self.evals_done, self._gen = 0, 0
        self.last_rank_of = []
        E._queue_serial[0] += 1
        self._prefix = f"cmoop/stubqueue/{E._queue_serial[0]}"

    def compute_objectives_and_constraints(self, population):
        from cmoop_audio_processing_amd import genes as G
        gl = [G.normalize_hparams(hp) for hp in population]
        costs = [G.fwd_flops_per_sample(g, self.variant, self.classes, self.T, self.F) / 1e9 for g in gl]
        self._gen += 1
        dist = self.E._dist()
        rank = float(dist.get_rank()) if dist is not None else 0.0

        def one(i):
            time.sleep(costs[i] * self.ms_per_gflop * 1e-3)
            return [0.5, G.model_size_mb(gl[i], self.variant, self.classes), 0.05, rank]

        def local(pull, workers):
            import concurrent.futures as cf
            with cf.ThreadPoolExecutor(max_workers=workers) as ex:      # as many pullers as the real evaluator's worker threads
                parts = list(ex.map(lambda _: {i: one(i) for i in iter(pull, -1)}, range(workers)))
            return {i: r for part in parts for i, r in part.items()}
        self.last_queue_stats = {}
        res = self.E.queued_map(local, costs, 4, f"{self._prefix}/{self._gen}", slots=self.slots, stats=self.last_queue_stats)
        self.evals_done += len(gl)
        self.last_rank_of = [int(r) for r in res[:, 3]]
        self.last_seconds = [costs[i] * self.ms_per_gflop * 1e-3 for i in range(len(gl))]
        return [{"hparams": hp, "objs": [-r[0], r[1], r[2]], "CV": 0.0} for hp, r in zip(population, res)]


def main(argv=None, t_origin=None):
    """t_origin: perf_counter value the budget clock starts at (default: this process' start; tests pass 'now')."""
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2, help="upper bound on timed generations (whole generations that fit --budget-s)")
    ap.add_argument("--warmup", type=int, default=1, help="upper bound on light untimed warm-up passes (each: the population, 1 epoch, a few hundred clips)")
    ap.add_argument("--budget-s", type=float, default=450.0,
                    help="seconds from process start by which the JSON line must be printed (driver limit 600 s)")
    ap.add_argument("--cpu-baseline-s", type=float, default=15.0, help="wall-clock cap of the CPU-baseline leg")
    ap.add_argument("--pop", type=int, default=40)
    ap.add_argument("--clips", type=int, default=30000, help="synthetic clips (80/10/10 split)")
    ap.add_argument("--epochs", type=int, default=10, help="fixed epoch budget per candidate (SURVEY section 8d: E_fixed = 10, early stopping off)")
    ap.add_argument("--protocol", default="fixed", choices=["fixed", "reference"],
                    help="fixed: --epochs epochs, early stopping off (the throughput metric). reference: what the reference runs -- "
                         "EarlyStopping(val_loss, patience 5), <= 300 epochs (nsga_penalty.py:159-161,377-384) on the hard synthetic set; "
                         "reports the epochs distribution; one generation, builder-side (minutes to tens of minutes)")
    ap.add_argument("--lone-steps", type=int, default=40, help="train steps of the lone-heaviest-candidate roofline leg (0: skip)")
    ap.add_argument("--lone-gene", default="", help="f,k,bn,R,fc,dr: gene of the lone-candidate leg (default: the heaviest of the population)")
    ap.add_argument("--lone-only", action="store_true",
                    help="setup + the lone-candidate leg only (no population run): the command to put under rocprofv3 --kernel-trace --stats")
    ap.add_argument("--variant", default="A")
    ap.add_argument("--classes", type=int, default=10)
    ap.add_argument("--slots", type=int, default=8, help="candidates in flight per GPU (6-16 measure the same; 4 is 9 % slower)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--weak", action="store_true", help="population = pop * gpus")
    ap.add_argument("--schedule", default="dynamic", choices=["dynamic", "static"],
                    help="N>1 candidate placement: one shared longest-first queue (c10d store counter) or LPT buckets by FLOPs")
    ap.add_argument("--profile-every", type=int, default=25)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gemm-mode", default="fp32", choices=["fp32", "bf16x3", "bf16"],
                    help="fp32 = exact fp32 MFMA (the product path, what the reference computes in). Opt-in bf16 matrix-core modes: "
                         "bf16x3 = every fp32 GEMM operand split exactly into three bf16 values, six bf16 MFMA terms (fp32-accurate, "
                         "not bit-exact); bf16 = operands rounded to bf16, fp32 accumulation (BASELINE configs[4] 'bf16 train')")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; 'gloo' + --same-device rehearses N ranks on one GPU")
    ap.add_argument("--same-device", action="store_true", help="every rank uses cuda:0 (single-GPU rehearsal of the N>1 path)")
    ap.add_argument("--stub", action="store_true", help="harness rehearsal without a GPU (tests only): StubEvaluator, gloo; never a measurement")
    ap.add_argument("--stub-ms-per-gflop", type=float, default=2.0)
    args = ap.parse_args(argv)
    if args.stub:
        args.backend = "gloo"

    under_rocprof = running_under_rocprof()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args, argv)

    # the clock the budget runs on starts with the (parent) process: torchrun children inherit the parent's offset
    if t_origin is None:
        t_origin = T_PROCESS_START - float(os.environ.get("CMOOP_BENCH_T0_OFFSET", "0") or 0.0)

    def since_start():
        return time.perf_counter() - t_origin

    # under rocprofv3 the tool library crashes on hipExtLaunchKernelGGL (ROCm 7.2): fall back to plain
    # hipEventRecord pairs around the sampled launches there (slightly inflated when streams overlap)
    if under_rocprof:
        os.environ["CMOOP_PROFILE_PAIRS"] = "1"
        try:    # keep the module map next to the profile: a tool-side fault is then attributable from the record alone
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open("/proc/self/maps") as f, open(os.path.join(ROOT, "gpurun_out", f"bench_maps_pid{os.getpid()}.txt"), "w") as g:
                g.write(f.read())
        except OSError:
            pass
    if args.gemm_mode != "fp32":
        os.environ["CMOOP_GEMM_MODE"] = args.gemm_mode      # read once by the library
    peak = PEAK_BY_MODE[args.gemm_mode]                      # the roofline of the arithmetic actually used

    import torch
    import torch.distributed as dist
    from cmoop_audio_processing_amd import EvalConfig, genes as G

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to mislabel the run", file=sys.stderr)
        return 2
    if args.same_device:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not args.stub:
            torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    elif not args.stub:
        torch.cuda.set_device(0)
    dev = torch.device("cpu") if args.stub else torch.device("cuda", local_rank if world > 1 else 0)
    coll_dev = dev if (world > 1 and args.backend == "nccl") else torch.device("cpu")
    variant = G.VARIANT_NAMES[args.variant]

    # heartbeat on stderr (stdout carries only the JSON line): long steps must not look hung
    stop_hb = threading.Event()
    phase = {"name": "setup"}

    def heartbeat():
        while not stop_hb.wait(60.0):
            print(f"[bench rank {rank}] {phase['name']}, {since_start():.0f} s since start", file=sys.stderr, flush=True)
    threading.Thread(target=heartbeat, daemon=True).start()

    n_tr, n_va = int(args.clips * 0.8), int(args.clips * 0.1)
    cpu_line, frontend_info = None, None
    if args.stub:
        T, F = 101, 40
        Xtr = ytr = Xva = yva = None
    else:
        from cmoop_audio_processing_amd import PopulationEvaluator, _lib, frontend
        # ---- untimed setup: synthetic clips -> HIP front end -> StandardScaler (nsga_penalty quirk Q1: refit per split)
        wav, y = synth_waveforms(args.clips, args.classes, 1234, dev, hard=(args.protocol == "reference"))
        frontend.log_mel(wav[:64])                       # warm-up (code object load, tables)
        torch.cuda.synchronize()
        t_fe = time.perf_counter()
        feats = frontend.log_mel(wav)                    # synchronous: returns after the kernel finished
        t_fe = time.perf_counter() - t_fe
        fe_bytes = wav.numel() * 4 + feats.numel() * 4   # algorithmic HBM bytes: clips in + log-mel out
        frontend_info = {"clips": int(args.clips), "ms": round(t_fe * 1e3, 3), "clips_per_s": round(args.clips / t_fe),
                         "algorithmic_GBps": round(fe_bytes / t_fe / 1e9, 1), "hbm_peak_GBps": 8000,
                         "frac_of_hbm_peak": round(fe_bytes / t_fe / 8e12, 4)}
        del wav
        Xtr, ytr = feats[:n_tr].contiguous(), y[:n_tr].contiguous()
        Xva, yva = feats[n_tr:n_tr + n_va].contiguous(), y[n_tr:n_tr + n_va].contiguous()
        del feats
        frontend.prepare_dataset(Xtr, Xva, None, mode="refit")
        T, F = int(Xtr.shape[1]), int(Xtr.shape[2])

    n_pop = args.pop * (world if args.weak else 1)
    rng = random.Random(args.seed)          # initialize_population: random.choice per gene (nsga_penalty.py:402-415)
    pop = [G.random_hparams(rng) for _ in range(n_pop)]
    genes = [G.normalize_hparams(hp) for hp in pop]

    # ---- CPU baseline FIRST (rank 0, N=1): bounded, and no timeout later in the run can lose it
    cpu_rate = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.stub and not args.lone_only:
        phase["name"] = "cpu baseline"
        xs, ys = Xtr[:64].cpu().numpy(), ytr[:64].cpu().numpy()
        cpu_rate = cpu_baseline_rate(genes, variant, args.classes, T, F, xs, ys, budget_s=args.cpu_baseline_s)

    reference_protocol = args.protocol == "reference"
    if reference_protocol:     # what the reference runs: nsga_penalty.py:159-161 (EPOCHS 300, BATCH 64, PATIENCE 5), :382-383
        cfg = EvalConfig.preset("nsga_penalty", variant=args.variant, classes=args.classes, epochs=300, patience=5,
                                early_stop=True, seed=args.seed, n_slots=args.slots, profile_every=args.profile_every,
                                schedule=args.schedule, fpr_variant="v1")
    else:
        cfg = EvalConfig.preset("nsga_penalty", variant=args.variant, classes=args.classes, epochs=args.epochs,
                                early_stop=False, seed=args.seed, n_slots=args.slots, profile_every=args.profile_every,
                                schedule=args.schedule)
    if args.stub:
        ev = StubEvaluator(cfg, variant, args.classes, T, F, args.stub_ms_per_gflop)
        ev_warm = ev
    else:
        ev = PopulationEvaluator(Xtr, ytr, Xva, yva, cfg)
        # warm-up evaluator: same population, one epoch on the first few hundred clips (every kernel family of every
        # candidate gets loaded and its buffers enter the cache; buffer sizes depend on the batch, not on N)
        from dataclasses import replace
        wn, wv = min(n_tr, 8 * cfg.batch), min(n_va, 2 * cfg.eval_batch)
        ev_warm = PopulationEvaluator(Xtr[:wn], ytr[:wn], Xva[:wv], yva[:wv], replace(cfg, epochs=1, profile_every=0))

    def barrier():
        if not args.stub:
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            if not args.stub:
                torch.cuda.synchronize()

    def agree(value, op):
        """the same float on every rank (MAX / MIN over ranks); outside N=1 one tiny all_reduce"""
        if world == 1:
            return float(value)
        t = torch.tensor([float(value)], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=op)
        return float(t.item())

    def read_profile():
        return [{"kernel": k, "launches": n, "ms": ms, "flops": fl} for (k, n, ms, fl) in _lib.profile_entries()]

    def per_kernel_rows(entries):
        return [{"kernel": e["kernel"], "launches": e["launches"], "avg_ms": round(e["ms"] / max(e["launches"], 1), 5),
                 "gflop_per_launch": round(e["flops"] / max(e["launches"], 1) / 1e9, 3),
                 "tflops": round(e["flops"] / (e["ms"] * 1e-3) / 1e12, 2) if e["ms"] > 0 else None,
                 "frac_of_peak": round(e["flops"] / (e["ms"] * 1e-3) / 1e12 / peak, 4) if e["ms"] > 0 else None}
                for e in sorted(entries, key=lambda e: -e["ms"])]

    def lone_candidate_leg(n_steps):
        """The heaviest candidate of the population training ALONE: one launching thread, one stream, nothing else on the
        chip; every MFMA launch of every step HIP-event-timed (profile_every = 1), in the trainer's own launch order, on
        the resident training tensor (batch 64).  Its per-kernel durations are work rates -- the in-job ones are not (the
        begin->end durations of co-scheduled kernels overlap).  `rocprofv3 --kernel-trace --stats -- python3 bench.py
        --lone-only` traces exactly this leg (profiles/)."""
        from dataclasses import replace as _replace
        from cmoop_audio_processing_amd.session import NetSession
        fl = [G.fwd_flops_per_sample(g, variant, args.classes, T, F) for g in genes]
        gene = genes[int(np.argmax(fl))]
        if args.lone_gene:
            gene = tuple(int(v) for v in args.lone_gene.split(","))
            fl = [G.fwd_flops_per_sample(gene, variant, args.classes, T, F)]
        lcfg = _replace(cfg, profile_every=1, n_slots=1, early_stop=False, epochs=1)
        rows = min(n_tr // cfg.batch, n_steps + 3) * cfg.batch
        with NetSession(gene, lcfg, T, F, args.seed) as net:
            net.set_gather_rows(n_tr)
            for i in range(3):                                  # untimed: code objects, row tables, buffer cache
                net.train_step(Xtr, ytr, None, row0=(i * cfg.batch) % max(rows, 1), B=cfg.batch)
            _lib.check(_lib.lib().cmoop_profile_reset())
            torch.cuda.synchronize()
            t_l = time.perf_counter()
            done = 0
            for i in range(n_steps):
                net.train_step(Xtr, ytr, None, row0=((i + 3) * cfg.batch) % max(rows, 1), B=cfg.batch)
                done += 1
                if since_start() > args.budget_s - 6.0 and not args.lone_only:
                    break
            t_l = time.perf_counter() - t_l
        ent = read_profile()
        step_fl = 3.0 * fl[int(np.argmax(fl))] * cfg.batch
        return {"gene": list(gene), "train_steps": done, "ms_per_step_incl_sync": round(t_l / max(done, 1) * 1e3, 3),
                "step_tflops": round(step_fl * done / t_l / 1e12, 2) if t_l > 0 else None,
                "what": "heaviest candidate of the population alone on the GPU: one launching thread, one stream, batch 64 on the resident "
                        "training tensor, every MFMA launch HIP-event-timed in the trainer's own launch order; host sync after each step",
                "per_kernel": per_kernel_rows(ent)}, ent

    warm_done = 0
    first = elapsed = 0.0
    steps, res, n_ranks_seen = 0, [], world
    epochs_run = []
    if not args.lone_only:
        phase["name"] = "warm-up"
        for _ in range(min(args.warmup, 2)):
            ev_warm.compute_objectives_and_constraints(pop)
            warm_done += 1
        if not args.stub:
            _lib.check(_lib.lib().cmoop_profile_reset())

        # ---- timed region: barrier + sync on both sides, whole generations, MAX over ranks
        reserve_s = min(30.0, 0.1 * args.budget_s)         # roofline legs + JSON after the timed region
        deadline = args.budget_s - reserve_s
        phase["name"] = "timed generation 1"
        barrier()
        t0 = time.perf_counter()
        res = ev.compute_objectives_and_constraints(pop)
        if not args.stub:
            torch.cuda.synchronize()
        first = agree(time.perf_counter() - t0, dist.ReduceOp.MAX if world > 1 else None)
        # the reference protocol's epochs are data-dependent: its line is ONE generation
        want = 1 if reference_protocol else max(1, args.steps)
        steps = int(agree(plan_steps(first, since_start(), deadline, want), dist.ReduceOp.MIN if world > 1 else None))
        for k in range(1, steps):
            phase["name"] = f"timed generation {k + 1} of {steps}"
            res = ev.compute_objectives_and_constraints(pop)
        barrier()
        elapsed = agree(time.perf_counter() - t0, dist.ReduceOp.MAX if world > 1 else None)
        n_ranks_seen = int(agree(1.0, dist.ReduceOp.SUM)) if world > 1 else 1
        epochs_run = list(getattr(ev, "last_epochs_run", []) or [])
    phase["name"] = "roofline legs"

    # ---- multi-GPU diagnostics of the last generation (outside the timed region): one small all_gather
    multi_gpu = None
    if world > 1 and not args.lone_only:
        qs = dict(getattr(ev, "last_queue_stats", {}) or {})
        mine = [float(sum(1 for r_ in (getattr(ev, "last_rank_of", []) or []) if r_ == rank)),
                float(sum(sec for r_, sec in zip(getattr(ev, "last_rank_of", []) or [], getattr(ev, "last_seconds", []) or []) if r_ == rank)),
                float(qs.get("local_s", 0.0)), float(qs.get("store_adds", 0)), float(qs.get("store_add_us_mean", 0.0)),
                float(qs.get("store_add_us_max", 0.0)), float(qs.get("all_gather_ms", 0.0)), float(qs.get("dealt_here", 0))]
        send = torch.tensor(mine, dtype=torch.float64, device=coll_dev)
        recv = torch.empty((world * len(mine),), dtype=torch.float64, device=coll_dev)
        dist.all_gather_into_tensor(recv, send)
        allr = recv.cpu().numpy().reshape(world, len(mine))
        multi_gpu = {"per_rank": [{"rank": r, "candidates": int(a[0]), "sum_candidate_seconds": round(a[1], 2),
                                   "busy_wall_s": round(a[2], 2), "dealt_at_start": int(a[7]), "store_fetch_adds": int(a[3]),
                                   "store_add_us_mean": round(a[4], 1), "store_add_us_max": round(a[5], 1),
                                   "all_gather_ms": round(a[6], 3)} for r, a in enumerate(allr)],
                     "note": "last timed generation; busy_wall_s = wall time the rank's worker threads spent training; the slowest rank's "
                             "busy_wall_s + all_gather_ms is the generation. RCCL = torch.distributed backend 'nccl' (backend in use: "
                             + args.backend + ")"}

    # ---- roofline
    roofline = None
    lone = None
    if not args.stub:
        L = _lib.lib()
        entries = read_profile() if not args.lone_only else []
        if rank == 0 and args.lone_steps > 0 and since_start() < args.budget_s - 15.0 or args.lone_only:
            lone, lone_entries = lone_candidate_leg(args.lone_steps)
        else:
            lone_entries = []
        if entries or lone_entries:
            # dominant = the instantiation that carries the most algorithmic FLOPs of the generation
            dom = max(entries or lone_entries, key=lambda e: e["flops"])
            in_job = None
            if entries:
                tot_ms, tot_fl = sum(e["ms"] for e in entries), sum(e["flops"] for e in entries)
                in_job = {"kernel": dom["kernel"], "tflops": round(dom["flops"] / (dom["ms"] * 1e-3) / 1e12, 3) if dom["ms"] > 0 else None,
                          "avg_launch_ms": round(dom["ms"] / max(dom["launches"], 1), 5), "sampled_launches": dom["launches"],
                          "all_mfma_kernels_tflops": round(tot_fl / (tot_ms * 1e-3) / 1e12, 3) if tot_ms > 0 else None,
                          "concurrent_streams": args.slots,
                          "note": "begin->end durations with up to 4 hardware queues' kernels sharing the CUs: NOT a work rate (their sum "
                                  "exceeds the step); kept for comparison with rocprofv3 traces of the whole job",
                          "per_kernel": per_kernel_rows(entries)}
            src = next((e for e in lone_entries if e["kernel"] == dom["kernel"]), None)
            measured = "lone heaviest candidate, HIP events (hipExtLaunchKernelGGL start/stop), this run"
            if src is None:       # the heaviest candidate does not launch the job's dominant instantiation: fall back to the in-job figure
                src, measured = dom, "in-job sampled launches (stretched by CU sharing)"
            achieved = src["flops"] / (src["ms"] * 1e-3) / 1e12 if src["ms"] > 0 else 0.0
            traffic, traffic_note = None, "no committed PMC summary found (profiles/pmc_traffic_per_kernel.json)"
            try:
                tj = json.load(open(TRAFFIC_FILE))
                ent = tj["kernels"].get(dom["kernel"])
                if ent:
                    traffic = ent["hbm_bytes_per_launch"]
                    traffic_note = (f"{tj['source']}: FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE of {dom['kernel']} on {ent['shape']} "
                                    f"= {ent['hbm_bytes_per_launch'] / 1e6:.0f} MB per launch against {ent['algorithmic_bytes_per_launch'] / 1e6:.0f} MB "
                                    "algorithmic; separate --pmc passes, not collected in this run")
            except (OSError, KeyError, ValueError):
                pass
            roofline = {"bound": "mfma", "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s",
                        "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_note": traffic_note,
                        "kernel": dom["kernel"], "measured": measured,
                        "launches": src["launches"], "gflop_per_launch": round(src["flops"] / max(src["launches"], 1) / 1e9, 3),
                        "avg_launch_ms": round(src["ms"] / max(src["launches"], 1), 5),
                        "timing": "hipEventRecord pairs (under rocprofv3)" if under_rocprof else "hipExtLaunchKernelGGL start/stop events",
                        "lone_candidate": lone, "in_job_stretched_by_cu_sharing": in_job}

        # ---- the same MFMA kernels as bare back-to-back launches (no net around them), single stream
        if rank == 0 and roofline is not None and not args.lone_only:
            iso = []
            for (B, H, W, Cin, Cout, KS) in ((64, 101, 40, 64, 64, 5), (64, 51, 20, 128, 128, 5), (64, 26, 10, 256, 256, 5)):
                if since_start() > args.budget_s - 8.0:
                    break
                x = torch.randn((B, H, W, Cin), device=dev)
                w = torch.randn((Cout, KS, KS, Cin), device=dev) * 0.05
                b = torch.randn((Cout,), device=dev)
                yb = torch.randn((B, H, W, Cout), device=dev)
                torch.cuda.synchronize()
                fl = 2.0 * B * H * W * Cout * KS * KS * Cin
                ent = {"conv": f"B{B} {H}x{W} {Cin}->{Cout} k{KS}", "gflop": round(fl / 1e9, 2)}
                for mode, nm in ((0, "fwd"), (1, "dgrad"), (2, "wgrad")):
                    ms = C.c_double()
                    _lib.check(L.cmoop_conv_time(mode, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(yb), B, H, W, Cin, Cout, KS, 10,
                                                 C.byref(ms)))
                    ent[nm + "_tflops"] = round(fl / ms.value / 1e9, 1)
                iso.append(ent)
            if iso:
                roofline["isolated_single_stream"] = iso
                roofline["isolated_frac_best"] = round(max(max(e["fwd_tflops"], e["dgrad_tflops"], e["wgrad_tflops"]) for e in iso) / peak, 4)

    rc = 0
    if rank == 0:
        if args.lone_only:
            line = {"metric": "lone heaviest candidate, train steps/s (profiling aid of bench.py, NOT the headline metric)",
                    "value": round(1e3 / lone["ms_per_step_incl_sync"], 2) if lone else None, "unit": "train-steps/s", "n_gpus": 1,
                    "dtype": "f32", "data": "synthetic", "higher_is_better": True,
                    "config": {"workload": f"gene {tuple(lone['gene']) if lone else None} alone, batch 64, features {T}x{F}, N_train={n_tr}"},
                    "roofline": roofline}
            print(json.dumps(line), flush=True)
            stop_hb.set()
            return 0
        evals = n_pop * steps
        value = evals / (elapsed / 3600.0)
        # FLOPs actually executed: the read-out pass reuses the last epoch's validation pass (same weights, deterministic
        # inference), so P = 0 extra inference passes here (the reference runs predict() again: P = 1)
        ep_of = epochs_run if (reference_protocol and len(epochs_run) == len(genes)) else [args.epochs] * len(genes)
        work = sum(G.eval_flops(g, variant, args.classes, T, F, n_tr, n_va, int(e), 0) for g, e in zip(genes, ep_of)) * steps
        if roofline is not None:   # chip-level view: all algorithmic conv/dense FLOPs of the step / wall time
            roofline["aggregate_timed_region"] = {"achieved": round(work / elapsed / 1e12 / world, 2),
                                                  "frac": round(work / elapsed / 1e12 / world / peak, 4),
                                                  "note": "all algorithmic conv+dense FLOPs of the timed region / wall time / GPUs"}
        if cpu_rate is not None:
            cpu_line = cpu_baseline_line(genes, variant, args.classes, T, F, n_tr, n_va,
                                         [int(e) for e in ep_of] if reference_protocol else args.epochs, *cpu_rate)
        per_rank = [0] * world
        for r_ in getattr(ev, "last_rank_of", []) or []:
            if 0 <= r_ < world:
                per_rank[r_] += 1
        if reference_protocol:
            proto_txt = ("the reference's protocol: EarlyStopping(val_loss, patience 5), <= 300 epochs (nsga_penalty.py:159-161,377-384), "
                         "hard synthetic set (-17 dB SNR, shared partials)")
        else:
            proto_txt = f"tiny-CNN train E={args.epochs} fixed epochs (SURVEY 8d throughput protocol, early stopping off)"
        line = {
            "metric": "candidate-net evals/hour (pop=40, GSC-v2)", "value": round(value, 2), "unit": "candidate-evals/hour",
            "n_gpus": world, "n_ranks_seen": n_ranks_seen, "steps": steps, "steps_requested": args.steps, "warmup": warm_done,
            "warmup_requested": args.warmup,
            "warmup_kind": "light pass: whole population, 1 epoch on the first 512 train / 512 val clips (loads every kernel, fills the buffer cache)",
            "ms_per_step": round(elapsed * 1e3 / steps, 2),
            "higher_is_better": True, "scaling": "weak" if args.weak else "strong", "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16x3": "f32 operands as 3 x bf16 (opt-in bf16x3 mode, fp32-accurate)",
                      "bf16": "bf16 operands, f32 accumulate (opt-in bf16-train mode)"}[args.gemm_mode],
            "data": "stub (harness rehearsal, NOT a measurement)" if args.stub else "synthetic",
            "config": {"workload": f"pop={n_pop} gen=1 fitness eval (topology {args.variant}, {args.classes} classes): "
                                   f"HIP log-mel front end (untimed) + {proto_txt}, "
                                   f"batch 64, N_train={n_tr}, N_val={n_va}, features {T}x{F}",
                       "protocol": args.protocol,
                       "population": n_pop, "epochs_per_candidate": args.epochs if not reference_protocol else None,
                       "n_train": n_tr, "n_val": n_va,
                       "slots_per_gpu": args.slots,
                       "parallelism": f"candidates over {world} GPU(s): " + ("one shared longest-first queue (c10d store counter)"
                                                                             if args.schedule == "dynamic" else "LPT buckets by closed-form FLOPs")
                                      + ", one all_gather of objective vectors per generation",
                       "candidates_per_rank_last_step": per_rank,
                       "n1_path": "world 1 takes the same code path with or without torchrun (no process group is created)"},
            "budget": {"budget_s": args.budget_s, "seconds_since_start_at_print": None, "first_step_s": round(first, 2)},
            "whole_job_tflops": round(work / elapsed / 1e12, 2),
            "mean_val_accuracy": round(float(np.mean([-r["objs"][0] for r in res])), 4),
            "frontend_untimed": frontend_info,
            "roofline": roofline,
        }
        if reference_protocol and epochs_run:
            ep = np.asarray(epochs_run, dtype=np.float64)
            secs = np.asarray(getattr(ev, "last_seconds", []) or [0.0], dtype=np.float64)
            line["reference_protocol"] = {
                "epochs_run": {"min": int(ep.min()), "median": float(np.median(ep)), "mean": round(float(ep.mean()), 2), "max": int(ep.max()),
                               "per_candidate": [int(e) for e in ep]},
                "slowest_candidate_share_of_generation": round(float(secs.max() / elapsed), 4) if elapsed > 0 else None,
                "accuracy_range": [round(float(min(-r["objs"][0] for r in res)), 4), round(float(max(-r["objs"][0] for r in res)), 4)],
                "note": "one generation of the protocol the reference runs; epochs are data-dependent, so evals/hour here is not comparable "
                        "with the fixed-E throughput metric"}
        if multi_gpu is not None:
            line["multi_gpu"] = multi_gpu
        if cpu_line is not None:
            line["cpu_baseline"] = cpu_line
        line["budget"]["seconds_since_start_at_print"] = round(since_start(), 1)
        print(json.dumps(line), flush=True)
        if n_ranks_seen != args.gpus:
            print(f"bench.py: {n_ranks_seen} ranks took part, --gpus {args.gpus}", file=sys.stderr)
            rc = 2
    stop_hb.set()
    if world > 1:
        dist.barrier()          # rank 0 ran the roofline legs and printed: every rank leaves the group together
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
