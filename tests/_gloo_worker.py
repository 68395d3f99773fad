"""world_size-2 gloo worker: exercises the candidate sharding + single all_gather path on CPU."""
import os
import sys

import numpy as np
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmoop_audio_processing_amd import genes as G, queued_map, sharded_map  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    import random
    rng = random.Random(0)   # replicated, seeded host state (SPMD)
    pop = [G.normalize_hparams(G.random_hparams(rng)) for _ in range(11)]
    costs = [float(G.fwd_flops_per_sample(g, 0, 10, 101, 40)) for g in pop]
    calls = []

    def local_fn(idx):
        calls.extend(idx)
        # fake objective vector that encodes the candidate and the rank that evaluated it
        return np.array([[G.param_count(pop[i], 0, 10), G.model_size_mb(pop[i], 0, 10), i, rank, 1.0] for i in idx], dtype=np.float64)

    out = sharded_map(local_fn, costs, 5, device="cpu")
    buckets = G.lpt_assign(costs, world)
    assert sorted(calls) == sorted(buckets[rank]), (calls, buckets)
    for i, g in enumerate(pop):
        assert out[i, 0] == G.param_count(g, 0, 10) and out[i, 1] == G.model_size_mb(g, 0, 10) and out[i, 2] == i
        assert out[i, 3] == [r for r, b in enumerate(buckets) if i in b][0]
    # fewer candidates than ranks and the empty generation
    out1 = sharded_map(lambda idx: np.array([[7.0] for _ in idx]).reshape(len(idx), 1), [1.0], 1, device="cpu")
    assert out1.shape == (1, 1) and out1[0, 0] == 7.0
    out0 = sharded_map(lambda idx: np.zeros((len(idx), 2)), [], 2, device="cpu")
    assert out0.shape == (0, 2)
    # ---- cross-rank dynamic queue (queued_map): adversarial ACTUAL costs the closed-form estimate cannot see --
    # 41 candidates of equal estimated cost, one of which really takes 20x (an early-stopped run that converges
    # slowly).  Static LPT buckets put 20/21 candidates on each rank -> one rank needs ~40 units, the other 20;
    # the shared longest-first queue lets the other rank drain the rest: both finish within 10 % of each other.
    import time
    unit = 0.01
    actual = [1.0] * 41
    actual[17] = 20.0
    est = [1.0] * 41

    def row(i):
        return [float(i), float(i * i), 0.5]

    def run_static():
        t0 = time.perf_counter()
        out = sharded_map(lambda idx: np.array([(time.sleep(actual[i] * unit), row(i))[1] for i in idx]).reshape(len(idx), 3),
                          est, 3, device="cpu")
        return out, time.perf_counter() - t0

    busy = {}

    def local_pull(pull):
        t0 = time.perf_counter()
        mine = {}
        while True:
            i = pull()
            if i < 0:
                break
            time.sleep(actual[i] * unit)
            mine[i] = row(i)
        busy["t"] = time.perf_counter() - t0
        return mine

    dist.barrier()
    out_q = queued_map(local_pull, est, 3, "test/queue/1", device="cpu")
    dist.barrier()
    out_s, t_static = run_static()          # includes waiting for the slower rank at the all_gather: the static makespan
    assert np.array_equal(out_q, out_s) and out_q[:, 0].tolist() == list(range(41))        # bit-identical to the static path
    import torch
    t = torch.tensor([busy["t"], t_static], dtype=torch.float64)
    ts = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(ts, t)
    tq = [float(x[0]) for x in ts]
    static_makespan = max(float(x[1]) for x in ts)
    # (sleep-based simulated time: a few units of absolute slack absorb scheduler jitter on a loaded CI host)
    assert max(tq) <= 1.10 * min(tq) + 4 * unit, f"queue left the ranks unbalanced: {tq}"
    assert max(tq) < 0.9 * static_makespan, f"queue no better than the static split: {tq} vs {static_makespan}"   # ~30.5 vs ~40 units
    # ---- deterministic head of the queue: with slots, each rank's workers START with the rank's share of queue_plan's
    # capacity-constrained LPT deal (no race for the large candidates), the counter hands out only the rest
    from cmoop_audio_processing_amd.evaluator import queue_plan
    costs_h = [9.0, 1.0, 8.0, 1.0, 7.0, 1.0, 6.0, 1.0, 5.0, 1.0, 1.0]
    order_h, W_h, heads_h = queue_plan(costs_h, world, 2)
    assert W_h == 2 and heads_h == [[0, 6], [2, 4]]
    first = []

    def local_heads(pull, workers):
        assert workers == W_h
        mine = {}
        for k in range(workers):                 # the workers' first pulls, before anything touches the shared counter
            i = pull()
            first.append(i)
            mine[i] = [float(i), float(rank)]
        dist.barrier()
        for i in iter(pull, -1):
            mine[i] = [float(i), float(rank)]
        return mine
    stats = {}
    out_h = queued_map(local_heads, costs_h, 2, "test/queue/heads", device="cpu", slots=2, stats=stats)
    assert first == heads_h[rank], (first, heads_h[rank])
    # r3: the diagnostics that make a multi-GPU line explain itself -- this rank's share of the deal, its fetch-adds on the
    # shared counter (7 queued candidates + one exhausted pull per rank in total) with their round trips, the all-gather
    assert stats["dealt_here"] == 2 and stats["store_adds"] >= 1 and stats["store_add_us_mean"] > 0 and stats["local_s"] > 0
    assert stats["all_gather_ms"] > 0 and stats["store_add_us_max"] >= stats["store_add_us_mean"]
    tot = torch.tensor([float(stats["store_adds"])], dtype=torch.float64)
    dist.all_reduce(tot)
    assert int(tot.item()) == 7 + world, tot
    assert out_h[:, 0].tolist() == [float(i) for i in range(11)]
    assert [int(out_h[i, 1]) for i in (0, 6)] == [0, 0] and [int(out_h[i, 1]) for i in (2, 4)] == [1, 1]
    # empty generation and fewer candidates than ranks through the queue
    assert queued_map(lambda pull: {}, [], 2, "test/queue/2", device="cpu").shape == (0, 2)
    one = queued_map(lambda pull: {i: [7.0] for i in iter(pull, -1)}, [1.0], 1, "test/queue/3", device="cpu")
    assert one.shape == (1, 1) and one[0, 0] == 7.0
    dist.barrier()
    if rank == 0:
        print("GLOO_WORKER_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
