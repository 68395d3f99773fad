#!/usr/bin/env python3
"""bench.py -- candidate-net evals/hour of the population-fitness hot path on MI355X.

One "step" = one pass of the hot path over one generation:
``compute_objectives_and_constraints(population)`` for pop=40 candidates
(BASELINE.json configs[1]: "pop=40 gen=1 fitness eval on 1xMI355X: HIP MFCC +
tiny-CNN train, GSC 10-keyword"; reference call site nsga_penalty.py:613/670).
Inputs are synthetic 1 s @ 16 kHz clips, turned into
standardised [N,101,40] log-mel features by the HIP front end BEFORE the timed
region, so the timed region starts with features resident in HBM.  Every
candidate trains for a fixed epoch budget (early stopping off) so CPU and GPU do
identical algorithmic work, then runs the inference/confusion readout.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: every rank holds the same seeded population, evaluates its LPT shard and
the objective vectors are exchanged with one RCCL all_gather per step.  The
population is fixed at --pop (strong scaling: the reference shards ONE
generation); --weak multiplies it by N.

Rank 0 prints ONE JSON line (contract in the task statement), including
``roofline`` (HIP-event timings of the dominant MFMA kernel sampled inside the
timed region) and, at N=1, ``cpu_baseline`` (the torch-CPU oracle timed on this
box's host cores on a bounded sample).
"""
import argparse
import ctypes as C
import json
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
# opt-in modes: the bf16 MFMA runs at 16x the fp32 MFMA rate (same guide); bf16x3 spends six bf16 MFMAs per product
PEAK_BY_MODE = {"fp32": PEAK_FP32_MFMA_TFLOPS, "bf16x3": round(16 * PEAK_FP32_MFMA_TFLOPS / 6, 1), "bf16": 16 * PEAK_FP32_MFMA_TFLOPS}


def synth_waveforms(n, classes, seed, device, n_samples=16000, chunk=2000):
    """per class 3 sinusoids log-spaced in 200-4000 Hz with random phase, amplitude
    U(0.1,1), plus N(0,1)-shaped noise at 0 dB SNR.  Seeded torch generator (Philox on GPU)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    y = (torch.arange(n, device=device) % classes).to(torch.int32)
    perm = torch.randperm(n, generator=g, device=device)
    y = y[perm].contiguous()
    freqs = torch.logspace(np.log10(200.0), np.log10(4000.0), classes * 3, device=device).reshape(classes, 3)
    t = torch.arange(n_samples, device=device, dtype=torch.float32) / 16000.0
    wav = torch.empty((n, n_samples), dtype=torch.float32, device=device)
    for s in range(0, n, chunk):
        yy = y[s:s + chunk].long()
        f = freqs[yy]                                                        # [c,3]
        ph = torch.rand((len(yy), 3), generator=g, device=device) * (2 * np.pi)
        sig = torch.sin(2 * np.pi * f[:, :, None] * t[None, None, :] + ph[:, :, None]).sum(1)
        sig = sig / sig.pow(2).mean(dim=1, keepdim=True).sqrt()
        noise = torch.randn((len(yy), n_samples), generator=g, device=device)
        amp = 0.1 + 0.9 * torch.rand((len(yy), 1), generator=g, device=device)
        wav[s:s + chunk] = amp * (sig + noise) * 0.5
    return wav, y


def cpu_baseline(pop, variant, classes, T, F, n_train, n_val, epochs, X_sample, y_sample, budget_s=25.0):
    """Time the oracle (torch-CPU restatement of the reference path) on this box's host cores
    on a bounded sample: train steps of batch 64 + inference on 64 rows for the cheapest,
    median and most expensive candidate of the population; extrapolate by closed-form FLOPs."""
    import torch
    from cmoop_audio_processing_amd import genes as G
    from oracle import net as ON
    cores = torch.get_num_threads()
    fl = [G.fwd_flops_per_sample(g, variant, classes, T, F) for g in pop]
    order = np.argsort(fl)
    picks = [int(order[0]), int(order[len(order) // 2]), int(order[-1])]
    cfg = ON.OracleConfig(variant=variant, classes=classes, batch=64)
    done_flops, spent, sample = 0.0, 0.0, []
    for i in picks:
        net = ON.OracleNet(pop[i], cfg, 1)
        xb, yb = X_sample[:64], y_sample[:64]
        net.train_step(xb, yb)                      # warm-up (allocator, MKLDNN primitives)
        steps = 0
        t0 = time.perf_counter()
        while True:
            net.train_step(xb, yb)
            steps += 1
            el = time.perf_counter() - t0
            if el > budget_s / 4 or steps >= 6:
                break
        t1 = time.perf_counter()
        net.evaluate(xb, yb)
        t2 = time.perf_counter()
        done_flops += fl[i] * 64 * (3 * steps + 1)
        spent += (t2 - t0)
        sample.append(f"gene{tuple(pop[i])}:{steps} train steps+1 eval batch in {t2 - t0:.1f}s")
    rate = done_flops / spent                                        # algorithmic FLOP/s the oracle sustains
    per_eval = np.array(fl, dtype=np.float64) * (3 * n_train * epochs + n_val * epochs + n_val)
    hours = per_eval.sum() / rate / 3600.0
    return {"value": len(pop) / hours, "unit": "candidate-evals/hour", "cores": int(cores), "kind": "port",
            "sample": "; ".join(sample) + f"; extrapolated by closed-form FLOPs to pop={len(pop)}, "
                      f"N_train={n_train}, E={epochs} ({rate / 1e9:.0f} GFLOP/s sustained)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--pop", type=int, default=40)
    ap.add_argument("--clips", type=int, default=30000, help="synthetic clips (80/10/10 split)")
    ap.add_argument("--epochs", type=int, default=2, help="fixed epoch budget per candidate; 10 = full SURVEY §8d protocol")
    ap.add_argument("--variant", default="A")
    ap.add_argument("--classes", type=int, default=10)
    ap.add_argument("--slots", type=int, default=8, help="candidates in flight per GPU (6-16 measure the same; 4 is 9 % slower)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--weak", action="store_true", help="population = pop * gpus")
    ap.add_argument("--profile-every", type=int, default=25)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gemm-mode", default="fp32", choices=["fp32", "bf16x3", "bf16"],
                    help="fp32 = exact fp32 MFMA (the product path, what the reference computes in). Opt-in bf16 matrix-core modes: "
                         "bf16x3 = every fp32 GEMM operand split exactly into three bf16 values, six bf16 MFMA terms (fp32-accurate, "
                         "not bit-exact); bf16 = operands rounded to bf16, fp32 accumulation (BASELINE configs[4] 'bf16 train')")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; 'gloo' + --same-device rehearses N ranks on one GPU")
    ap.add_argument("--same-device", action="store_true", help="every rank uses cuda:0 (single-GPU rehearsal of the N>1 path)")
    args = ap.parse_args()

    # under rocprofv3 the tool library crashes on hipExtLaunchKernelGGL (ROCm 7.2): fall back to plain
    # hipEventRecord pairs around the sampled launches there (slightly inflated when streams overlap)
    under_rocprof = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    if under_rocprof:
        os.environ["CMOOP_PROFILE_PAIRS"] = "1"
    if args.gemm_mode != "fp32":
        os.environ["CMOOP_GEMM_MODE"] = args.gemm_mode      # read once by the library

    global PEAK_FP32_MFMA_TFLOPS
    PEAK_FP32_MFMA_TFLOPS = PEAK_BY_MODE[args.gemm_mode]   # the roofline of the arithmetic actually used

    import torch
    import torch.distributed as dist
    from cmoop_audio_processing_amd import EvalConfig, PopulationEvaluator, _lib, frontend, genes as G

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.same_device:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    else:
        torch.cuda.set_device(0)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    variant = G.VARIANT_NAMES[args.variant]

    # ---- untimed setup: synthetic clips -> HIP front end -> StandardScaler (nsga_penalty quirk Q1: refit per split)
    wav, y = synth_waveforms(args.clips, args.classes, 1234, dev)
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
    n_tr, n_va = int(args.clips * 0.8), int(args.clips * 0.1)
    Xtr, ytr = feats[:n_tr].contiguous(), y[:n_tr].contiguous()
    Xva, yva = feats[n_tr:n_tr + n_va].contiguous(), y[n_tr:n_tr + n_va].contiguous()
    del feats
    frontend.prepare_dataset(Xtr, Xva, None, mode="refit")
    T, F = int(Xtr.shape[1]), int(Xtr.shape[2])

    cfg = EvalConfig.preset("nsga_penalty", variant=args.variant, classes=args.classes, epochs=args.epochs,
                            early_stop=False, seed=args.seed, n_slots=args.slots, profile_every=args.profile_every)
    ev = PopulationEvaluator(Xtr, ytr, Xva, yva, cfg)
    n_pop = args.pop * (world if args.weak else 1)
    rng = random.Random(args.seed)          # initialize_population: random.choice per gene (nsga_penalty.py:402-415)
    pop = [G.random_hparams(rng) for _ in range(n_pop)]
    genes = [G.normalize_hparams(hp) for hp in pop]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # heartbeat on stderr (stdout carries only the JSON line): long steps must not look hung
    import threading
    stop_hb = threading.Event()

    def heartbeat():
        t_hb = time.perf_counter()
        while not stop_hb.wait(60.0):
            print(f"[bench rank {rank}] running, {time.perf_counter() - t_hb:.0f} s", file=sys.stderr, flush=True)
    threading.Thread(target=heartbeat, daemon=True).start()

    for _ in range(args.warmup):
        ev.compute_objectives_and_constraints(pop)
    _lib.check(_lib.lib().cmoop_profile_reset())
    barrier()
    t0 = time.perf_counter()
    res = None
    for _ in range(args.steps):
        res = ev.compute_objectives_and_constraints(pop)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- roofline of the dominant MFMA kernel, from HIP events recorded inside the timed region
    L = _lib.lib()
    cnt = C.c_int32()
    _lib.check(L.cmoop_profile_count(C.byref(cnt)))
    entries = []
    for i in range(cnt.value):
        name = C.create_string_buffer(128)
        n, ms, fl = C.c_int64(), C.c_double(), C.c_double()
        _lib.check(L.cmoop_profile_entry(i, name, 128, C.byref(n), C.byref(ms), C.byref(fl)))
        entries.append({"kernel": name.value.decode(), "launches": n.value, "ms": ms.value, "flops": fl.value})
    roofline = None
    if entries:
        # dominant = the instantiation that carries the most algorithmic FLOPs of the step (with 8 candidates in
        # flight, total sampled time instead picks whichever long-grid kernel shared the chip with most others)
        dom = max(entries, key=lambda e: e["flops"])
        achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0
        tot_ms = sum(e["ms"] for e in entries)
        tot_fl = sum(e["flops"] for e in entries)
        roofline = {"bound": "mfma", "achieved": round(achieved, 3), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": None,
                    "traffic_note": "PMC needs its own rocprofv3 passes (one counter each), so it is not collected in this run: "
                                    "profiles/r01_pmc_fetch_write_final_kernels.csv holds FETCH_SIZE / WRITE_SIZE of this kernel on its "
                                    "heaviest layer shape (175 MB per launch against 68 MB algorithmic; compute-bound)",
                    "kernel": dom["kernel"], "sampled_launches": dom["launches"],
                    "timing": "hipEventRecord pairs (under rocprofv3)" if under_rocprof else "hipExtLaunchKernelGGL start/stop events",
                    "avg_launch_ms": round(dom["ms"] / max(dom["launches"], 1), 5),
                    "all_mfma_kernels_tflops": round(tot_fl / (tot_ms * 1e-3) / 1e12, 3) if tot_ms > 0 else None,
                    "per_kernel": [{"kernel": e["kernel"], "launches": e["launches"],
                                    "avg_ms": round(e["ms"] / max(e["launches"], 1), 5),
                                    "tflops": round(e["flops"] / (e["ms"] * 1e-3) / 1e12, 2) if e["ms"] > 0 else None}
                                   for e in sorted(entries, key=lambda e: -e["ms"])]}

    # ---- the same MFMA kernels ALONE on the GPU (single stream, after the timed region): with several
    # candidates in flight the per-launch durations above are stretched by the kernels they share the
    # CUs with, so they understate kernel quality; this leg is the per-kernel roofline fraction.
    if rank == 0 and roofline is not None:
        iso = []
        for (B, H, W, Cin, Cout, KS) in ((64, 101, 40, 64, 64, 5), (64, 51, 20, 128, 128, 5), (64, 26, 10, 256, 256, 5)):
            x = torch.randn((B, H, W, Cin), device=dev)
            w = torch.randn((Cout, KS, KS, Cin), device=dev) * 0.05
            b = torch.randn((Cout,), device=dev)
            yb = torch.randn((B, H, W, Cout), device=dev)
            torch.cuda.synchronize()
            fl = 2.0 * B * H * W * Cout * KS * KS * Cin
            ent = {"conv": f"B{B} {H}x{W} {Cin}->{Cout} k{KS}", "gflop": round(fl / 1e9, 2)}
            for mode, nm in ((0, "fwd"), (1, "dgrad"), (2, "wgrad")):
                ms = C.c_double()
                _lib.check(L.cmoop_conv_time(mode, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(yb), B, H, W, Cin, Cout, KS, 20,
                                             C.byref(ms)))
                ent[nm + "_tflops"] = round(fl / ms.value / 1e9, 1)
            iso.append(ent)
        roofline["isolated_single_stream"] = iso
        roofline["isolated_frac_best"] = round(max(max(e["fwd_tflops"], e["dgrad_tflops"], e["wgrad_tflops"]) for e in iso)
                                               / PEAK_FP32_MFMA_TFLOPS, 4)
        roofline["concurrent_streams"] = args.slots

    if rank == 0:
        evals = n_pop * args.steps
        value = evals / (elapsed / 3600.0)
        work = sum(G.eval_flops(g, variant, args.classes, T, F, n_tr, n_va, args.epochs, 1) for g in genes) * args.steps
        if roofline is not None:   # chip-level view: all algorithmic conv/dense FLOPs of the step / wall time
            roofline["aggregate_timed_region"] = {"achieved": round(work / elapsed / 1e12, 2),
                                                  "frac": round(work / elapsed / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)}
        line = {
            "metric": "candidate-net evals/hour (pop=40, GSC-v2)", "value": round(value, 2), "unit": "candidate-evals/hour",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed * 1e3 / args.steps, 2),
            "higher_is_better": True, "scaling": "weak" if args.weak else "strong", "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16x3": "f32 operands as 3 x bf16 (opt-in bf16x3 mode, fp32-accurate)",
                      "bf16": "bf16 operands, f32 accumulate (opt-in bf16-train mode)"}[args.gemm_mode],
            "data": "synthetic",
            "config": {"workload": f"pop={n_pop} gen=1 fitness eval (topology {args.variant}, {args.classes} classes): "
                                   f"HIP log-mel front end (untimed) + tiny-CNN train E={args.epochs} fixed epochs, "
                                   f"batch 64, N_train={n_tr}, N_val={n_va}, features {T}x{F}",
                       "population": n_pop, "epochs_per_candidate": args.epochs, "n_train": n_tr, "n_val": n_va,
                       "slots_per_gpu": args.slots, "parallelism": f"candidates sharded over {world} GPU(s), LPT by FLOPs"},
            "whole_job_tflops": round(work / elapsed / 1e12, 2),
            "mean_val_accuracy": round(float(np.mean([-r["objs"][0] for r in res])), 4),
            "frontend_untimed": frontend_info,
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            xs = Xtr[:64].cpu().numpy()
            ys = ytr[:64].cpu().numpy()
            line["cpu_baseline"] = cpu_baseline(genes, variant, args.classes, T, F, n_tr, n_va, args.epochs, xs, ys)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
