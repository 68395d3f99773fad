"""world_size-2 gloo worker: exercises the candidate sharding + single all_gather path on CPU."""
import os
import sys

import numpy as np
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmoop_audio_processing_amd import genes as G, sharded_map  # noqa: E402


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
    dist.barrier()
    if rank == 0:
        print("GLOO_WORKER_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
