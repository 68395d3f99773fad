"""CPU: host logic, the C-ABI library's exports, and the N>1 sharding path over gloo."""
import ctypes as C
import os
import random
import subprocess
import sys

import numpy as np
import pytest

from cmoop_audio_processing_amd import _lib, genes as G
from cmoop_audio_processing_amd.evaluator import EvalConfig, calculate_fpr, compute_model_size_mb, sharded_map

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_loads_and_exports_every_declared_symbol():
    _lib.build()
    L = _lib.lib()
    names = _lib.declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), n
    assert L.cmoop_abi_version() == 3


def test_abi_host_only_closed_forms_match_python():
    L = _lib.lib()
    for variant in (0, 1):
        for classes in (10, 11, 35):
            for g in G.all_genes()[::7]:
                arr = (C.c_int32 * 6)(*g)
                n, f = C.c_int64(), C.c_double()
                _lib.check(L.cmoop_param_count(arr, variant, classes, C.byref(n)))
                _lib.check(L.cmoop_fwd_flops(arr, variant, classes, 101, 40, C.byref(f)))
                assert n.value == G.param_count(g, variant, classes)
                assert f.value == float(G.fwd_flops_per_sample(g, variant, classes, 101, 40))
    bad = (C.c_int32 * 6)(48, 3, 1, 1, 1, 0)
    n = C.c_int64()
    assert L.cmoop_param_count(bad, 0, 10, C.byref(n)) != 0
    assert b"search space" in L.cmoop_last_error()


def test_abi_fpr_matches_reference_goldens(golden_dir):
    import json
    for c in json.load(open(os.path.join(golden_dir, "fpr_golden.json")))["cases"]:
        assert calculate_fpr(c["y_true"], c["y_pred"], c["C"], "v1") == pytest.approx(c["v1"], abs=1e-15)
        assert calculate_fpr(c["y_true"], c["y_pred"], c["C"], "v3") == pytest.approx(c["v3"], abs=1e-15)
        assert calculate_fpr(c["y_true"], c["y_pred"], c["C"], "v1_quirk") == pytest.approx(c["v1_quirk"], abs=1e-15)


def test_abi_permutation_matches_oracle_rng():
    from cmoop_audio_processing_amd.session import epoch_permutation
    from oracle import rng
    for seed, epoch, n in ((0, 0, 1), (1, 2, 17), (123456789, 299, 24000)):
        p = epoch_permutation(seed, epoch, n)
        assert np.array_equal(p, rng.epoch_permutation(seed, epoch, n))
        assert sorted(p.tolist()) == list(range(n))


def test_presets_follow_the_reference_scripts():
    a = EvalConfig.preset("nsga_penalty")
    assert (a.variant, a.classes, a.restore_best, a.acc_readout, a.fpr_variant) == ("A", 10, False, "last", "v1_quirk")
    assert (a.min_accuracy, a.max_model_size, a.max_fpr, a.epochs, a.batch, a.patience) == (0.9, 2.5, 0.1, 300, 64, 5)
    b = EvalConfig.preset("sa_nsga_penalty")
    assert (b.variant, b.classes, b.restore_best, b.acc_readout, b.fpr_variant) == ("B", 11, True, "evaluate", "v1")
    assert (b.min_accuracy, b.max_fpr) == (0.75, 0.09)
    assert EvalConfig.preset("sa_nsga_local").fpr_variant == "v3"
    with pytest.raises(KeyError):
        EvalConfig.preset("nope")


def test_size_mb_helper_and_hparams_normalisation():
    hp = {"filters": np.int64(32), "kernel_size": 3, "use_bn": 1, "residual_blocks": np.int32(2), "fc_layers": 2, "use_dropout": 0}
    assert compute_model_size_mb(hp, "A", 10) == 324074 * 4 / 1024 ** 2
    assert G.normalize_hparams(hp) == (32, 3, 1, 2, 2, 0)
    with pytest.raises(ValueError):
        G.normalize_hparams(dict(hp, kernel_size=4))


def test_random_hparams_draws_like_the_reference():
    # initialize_population: random.choice per gene in dict-key order (nsga_penalty.py:405-412)
    r1, r2 = random.Random(3), random.Random(3)
    hp = G.random_hparams(r1)
    exp = {"filters": r2.choice([16, 32, 64]), "kernel_size": r2.choice([3, 5]), "use_bn": r2.choice([True, False]),
           "residual_blocks": r2.choice([1, 2, 3]), "fc_layers": r2.choice([1, 2, 3, 4]), "use_dropout": r2.choice([True, False])}
    assert hp == exp and list(hp) == list(exp)


def test_sharded_map_without_process_group_is_local():
    out = sharded_map(lambda idx: np.array([[i, 2 * i] for i in idx], dtype=np.float64), [3.0, 1.0, 2.0], 2)
    assert out.tolist() == [[0, 0], [1, 2], [2, 4]]


def test_sharding_world_size_2_gloo():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:      # a free port: a fixed one can still be in TIME_WAIT
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "_gloo_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "GLOO_WORKER_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "cmoop_audio_processing_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn


def test_compute_mode_reaches_the_abi_struct():
    """EvalConfig.compute -> cmoop_config.gemm_mode (include/cmoop.h CMOOP_GEMM_*); the default is the reference's fp32."""
    import pytest
    from cmoop_audio_processing_amd import EvalConfig
    assert EvalConfig().compute == "fp32" and EvalConfig().to_struct().gemm_mode == 0
    assert EvalConfig(compute="bf16x3").to_struct().gemm_mode == 2
    assert EvalConfig(compute="bf16").to_struct().gemm_mode == 3
    with pytest.raises(KeyError):
        EvalConfig(compute="fp16").to_struct()
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "cmoop.h")).read()
    assert "#define CMOOP_GEMM_BF16X3 2" in hdr and "#define CMOOP_GEMM_BF16 3" in hdr and "int32_t gemm_mode;" in hdr


def test_queued_map_without_process_group_is_a_local_longest_first_queue():
    from cmoop_audio_processing_amd import queued_map
    seen = []

    def local(pull):
        out = {}
        for i in iter(pull, -1):
            seen.append(i)
            out[i] = [float(i), 2.0 * i]
        return out
    out = queued_map(local, [3.0, 1.0, 2.0, 3.0], 2, "unused")
    assert seen == [0, 3, 2, 1]                                   # longest first, ties by index
    assert out.tolist() == [[0, 0], [1, 2], [2, 4], [3, 6]]
    assert queued_map(lambda pull: {}, [], 3, "unused").shape == (0, 3)
    import pytest
    with pytest.raises(Exception):                                # a worker that drops a candidate is an error, not a NaN
        queued_map(lambda pull: {}, [1.0], 1, "unused")


def test_wgrad_slice_count_is_not_monotone_in_the_batch_and_the_abi_reports_it():
    """the trainer sized its wgrad slab workspace from the FULL batch although a partial last batch
    can ask for MORE slices (gene (32,3,*,1,*,*) topology A, the block's second conv at 51x20, 64->64 k3: 98 slices at B=64, 109 at B=28..51).
    The fix sizes for the worst B in 1..batch and clamps; this pins the host heuristic through the C ABI (no GPU)."""
    import ctypes as C
    from cmoop_audio_processing_amd import _lib, genes as G
    L = _lib.lib()

    def slices(B, H, W, Cin, Cout, KS, stride=1):
        out = C.c_int32()
        _lib.check(L.cmoop_wgrad_slices(B, H, W, Cin, Cout, KS, stride, C.byref(out)))
        return out.value
    # that layer's weight gradient runs on the halo-tiled kernel since round 3 (one round of 512 workgroups: 128 slices); the
    # pinned non-monotonicity belongs to the implicit-GEMM heuristic, read in a child process with the halo kernel off
    import subprocess
    code = ("import ctypes as C, sys; sys.path.insert(0, %r); from cmoop_audio_processing_amd import _lib; L = _lib.lib(); o = C.c_int32();\n"
            "r = []\n"
            "for b in range(1, 65):\n"
            "    _lib.check(L.cmoop_wgrad_slices(b, 51, 20, 64, 64, 3, 1, C.byref(o))); r.append(o.value)\n"
            "print(r[63], max(r), r[27])") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, CMOOP_HALO_WGRAD="0"), timeout=120)
    assert out.stdout.split() == ["98", "109", "109"], out.stdout + out.stderr      # the non-monotonicity that overflowed the workspace
    partial = [slices(b, 51, 20, 64, 64, 3) for b in range(1, 65)]
    assert slices(64, 51, 20, 64, 64, 3) == 128 and max(partial) == 128 and all(s >= 1 for s in partial)
    # every conv/dense layer of every gene: the worst-case need over B is what Net::build_plan now allocates; here
    # we only check the ABI answers for all of them and that the need is bounded (slab cap: 16M floats + one slice)
    worst = 0
    for g in [(16, 3, 0, 3, 4, 0), (32, 5, 0, 2, 2, 0), (64, 5, 0, 3, 1, 0), (64, 3, 0, 1, 4, 0)]:
        h, w = 101, 40
        for spec in G.layer_specs(g, 0, 10):
            if spec["kind"] != "conv" or spec["cin"] == 1:
                continue
            if spec["name"].endswith("_skip"):
                need = max(slices(b, h, w, spec["cin"], spec["cout"], 1, 2) * spec["cout"] * (spec["cin"] + 1) for b in (1, 17, 40, 64))
            else:
                if spec["name"].endswith("conv1") and spec["name"].startswith("res"):
                    pass
                need = max(slices(b, h, w, spec["cin"], spec["cout"], spec["k"]) * spec["cout"] * (spec["k"] ** 2 * spec["cin"] + 1)
                           for b in (1, 17, 40, 64))
            worst = max(worst, need)
            if spec["name"] in ("conv2",) or spec["name"].endswith("conv2"):
                h, w = (h + 1) // 2, (w + 1) // 2
    assert 0 < worst < (1 << 26)


def test_result_packing_matches_every_reference_script(golden_dir):
    """compute_objectives_and_constraints' result entries -- three objectives (nsga_penalty.py:418-442,
    sa_nsga_penalty.py:231-253) and the bi-objective ablations with their tracked keys (acc_fpr_nsga_1.py:283-310,
    acc_size_nsga_1.py:283-311, size_fpr_nsga_1.py:283-310) -- against the outputs of the reference's own functions
    (tests/golden/make_golden.py), bit for bit, including key order and the reference to the caller's dict; and the
    per-generation CSV row read back from any packing."""
    import json
    from cmoop_audio_processing_amd.evaluator import EvalConfig, pack_result
    from cmoop_audio_processing_amd.nsga import accuracy_size_fpr, generation_records
    preset_of = {"nsga_penalty.py": "nsga_penalty", "sa_nsga_penalty.py": "sa_nsga_penalty",
                 "ablation_study/acc_fpr_nsga_1.py": "acc_fpr_nsga_1", "ablation_study/acc_size_nsga_1.py": "acc_size_nsga_1",
                 "ablation_study/size_fpr_nsga_1.py": "size_fpr_nsga_1"}
    g = json.load(open(os.path.join(golden_dir, "objectives_golden.json")))
    seen = set()
    for case in g["cases"]:
        cfg = EvalConfig.preset(preset_of[case["script"]])
        t = case["thresholds"]
        used = {"all": ("MIN_ACCURACY", "MAX_MODEL_SIZE", "MAX_FPR"), "acc_fpr": ("MIN_ACCURACY", "MAX_FPR"),
                "acc_size": ("MIN_ACCURACY", "MAX_MODEL_SIZE"), "size_fpr": ("MAX_MODEL_SIZE", "MAX_FPR")}[cfg.objectives]
        ours = {"MIN_ACCURACY": cfg.min_accuracy, "MAX_MODEL_SIZE": cfg.max_model_size, "MAX_FPR": cfg.max_fpr}
        assert all(ours[k] == t[k] for k in used)
        seen.add(cfg.objectives)
        for r in case["records"]:
            hp = {"filters": 16}
            out = pack_result(hp, r["acc"], r["size_mb"], r["fpr"], cfg)
            assert out["hparams"] is hp and out["objs"] == r["objs"] and out["CV"] == r["CV"]
            for key in ("size_metric", "fpr_metric", "acc_metric"):
                assert (key in out) == (key in r) and (key not in r or out[key] == r[key])
            assert accuracy_size_fpr(out) == (r["acc"], r["size_mb"], r["fpr"])
            row = generation_records(3, [out])[0]
            assert (row["Generation"], row["Accuracy"], row["Size_MB"], row["FPR"], row["CV"], row["filters"]) == \
                   (3, r["acc"], r["size_mb"], r["fpr"], r["CV"], 16)
    assert seen == {"all", "acc_fpr", "acc_size", "size_fpr"}
    with pytest.raises(ValueError):
        pack_result({}, 0.9, 1.0, 0.1, EvalConfig(objectives="nope"))


def test_queue_plan_deals_the_first_items_by_capacity_constrained_lpt():
    """The cross-rank queue's deterministic head (evaluator.queue_plan): with 8 ranks x 8 worker threads a 40-candidate
    generation would be drained in ONE burst of fetch-adds and the assignment would be a race; instead no rank starts
    more than ceil(n / world) candidates at once and the items the workers start with are dealt longest-first to the
    least-loaded rank with a free worker.  bench.py's pop-40 population on 8 ranks: 5 per rank, every candidate dealt,
    max load within 25 % of the mean (an arbitrary burst can put the 8 largest on one rank: 2-3x the mean)."""
    import random
    from cmoop_audio_processing_amd.evaluator import queue_plan
    rng = random.Random(0)
    genes = [G.normalize_hparams(G.random_hparams(rng)) for _ in range(40)]
    costs = [float(G.fwd_flops_per_sample(g, 0, 10, 101, 40)) for g in genes]
    order, W, heads = queue_plan(costs, 8, 8)
    assert W == 5 and sorted(i for h in heads for i in h) == list(range(40)) and all(len(h) == 5 for h in heads)
    assert order == sorted(range(40), key=lambda i: (-costs[i], i))
    loads = [sum(costs[i] for i in h) for h in heads]
    assert max(loads) <= 1.25 * (sum(costs) / 8), (max(loads), sum(costs) / 8)
    assert all(h == sorted(h, key=lambda i: (-costs[i], i)) for h in heads)          # each rank starts its longest first
    burst = sum(sorted(costs, reverse=True)[:8])                                      # what a race could give one rank
    assert burst > 2.0 * max(loads)
    # more candidates than workers: only the first world * W positions are dealt, the rest stays in the shared queue
    order, W, heads = queue_plan(costs, 2, 8)
    assert W == 8 and [len(h) for h in heads] == [8, 8] and sorted(i for h in heads for i in h) == sorted(order[:16])
    # fewer candidates than ranks, one candidate, none
    assert queue_plan([3.0, 1.0], 4, 8) == ([0, 1], 1, [[0], [1], [], []])
    assert queue_plan([], 4, 8) == ([], 1, [[], [], [], []])
    # single process: slots bound the workers, the head is the first W of the order
    order, W, heads = queue_plan([1.0, 5.0, 2.0], 1, 2)
    assert (order, W, heads) == ([1, 2, 0], 2, [[1, 2]])


def test_queued_map_with_slots_runs_the_workers_it_plans():
    from cmoop_audio_processing_amd import queued_map
    seen = {}

    def local(pull, workers):
        seen["workers"] = workers
        return {i: [float(i)] for i in iter(pull, -1)}
    out = queued_map(local, [1.0, 4.0, 2.0, 3.0, 5.0], 1, "unused", slots=3)
    assert seen["workers"] == 3 and out[:, 0].tolist() == [0.0, 1.0, 2.0, 3.0, 4.0]


def test_shape_time_share_tool_walks_the_same_layers_as_the_model():
    """tools/shape_time_share.py (where the generation's GEMM time goes, by layer shape) re-derives the conv stack of
    topology A from the gene; it must list exactly the convs genes.layer_specs describes (minus the C_in = 1 first conv)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("shape_time_share", os.path.join(ROOT, "tools", "shape_time_share.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for g in G.all_genes():
        ours = [(ci, co, k, st) for (_, _, ci, co, k, st, _) in mod.conv_shapes(g)]
        ref = [(l["cin"], l["cout"], l["k"], l["stride"]) for l in G.layer_specs(g, G.VARIANT_A, 10) if l["kind"] == "conv"][1:]
        assert ours == ref, g
    # spatial sizes: SAME pools halve with ceil (101x40 -> 51x20 -> 26x10 -> 13x5)
    assert [(h, w) for (h, w, *_rest) in mod.conv_shapes((16, 3, 0, 3, 1, 0))] == \
           [(101, 40)] + [(51, 20)] * 3 + [(26, 10)] * 3 + [(13, 5)] * 3


def test_every_launch_variant_of_every_gene_has_a_gpu_parity_case():
    """the kernels that carry the benchmark must not run uncompared.  cmoop_conv_launch_plan (host-only
    twin of the launchers' tile / split-K / operand-path choice) gives the launch-path variant the trainer uses for a conv
    layer; enumerate forward (with / without the BatchNorm statistics epilogue), dgrad and wgrad of EVERY conv layer of
    EVERY gene of the search space, both topologies, at the reference's batch (64), a partial last batch (37) and the
    inference batch (256) on 101x40 features -- each variant must be launched by one of the production-shape parity cases
    of tests/test_gpu_production_shapes.py (same plan function applied to its shape list)."""
    import itertools
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import launch_variants as LV
    from _production_shapes import PRODUCTION_CONVS
    covered = set()
    for (B, H, W, Ci, Co, KS, st) in PRODUCTION_CONVS:
        covered |= {LV.plan(0, B, H, W, Ci, Co, KS, st, 1), LV.plan(0, B, H, W, Ci, Co, KS, st, 0),
                    LV.plan(1, B, H, W, Ci, Co, KS, st), LV.plan(2, B, H, W, Ci, Co, KS, st)}
    genes = list(itertools.product((16, 32, 64), (3, 5), (0, 1), (1, 2, 3), (1,), (0,)))     # fc / dropout genes add no conv shape
    for variant in (0, 1):
        used = LV.variants_of(genes, variant, 101, 40)
        missing = {k: sorted(v)[:3] for k, v in used.items() if k not in covered}
        assert not missing, f"launch-path variants without a GPU parity case (topology {variant}): {missing}"
    assert len(covered) >= 30
    # the names are the ones rocprofv3 prints: the LDS-DMA instantiation carries MODE = 1
    assert "igemm_fwd_kernel<128, 32, 32, 4, 1>+tab" in covered and "igemm_wgrad_kernel<128, 128, 32>+tab+slabs" in covered and "igemm_wgrad_kernel<64, 64, 64>+tab+slabs" in covered
    # round 3: the halo-tiled direct convolution and weight gradient of the stride-1 layers with many output pixels
    assert {"halo_fwd_kernel<5, 128, 64, 2, false>+stats", "halo_fwd_kernel<3, 256, 32, 4, false>", "halo_fwd_kernel<5, 128, 64, 2, true>+bal", "halo_wgrad_kernel<5, 10>+slabs", "halo_wgrad_kernel<3, 5>+slabs"} <= covered


def test_halo_tile_bound_covers_every_tile_of_every_geometry():
    """The halo-tiled direct convolution sizes its LDS image and its per-thread staging slots from a closed-form bound on the
    rows a flat 128- / 256-pixel tile can span (image boundaries add gap rows).  cmoop_halo_tile_check walks every tile of
    a geometry with the kernel's own row arithmetic: the bound must cover the worst tile and fit the staging slots -- for
    the search space's layer sizes, the BirdCLEF-shaped ones, odd widths and batches that put 1 ... 3 images into a tile."""
    L = _lib.lib()
    eligible = 0
    for (H, W) in [(101, 40), (51, 20), (26, 10), (13, 5), (7, 3), (32, 32), (16, 16), (8, 8), (41, 20), (21, 12), (11, 6), (6, 3),
                   (3, 40), (2, 2), (1, 40), (5, 1), (64, 64), (128, 128), (9, 17), (100, 7)]:
        for B in (1, 2, 37, 64, 256):
            for (Cin, Cout) in ((16, 16), (32, 32), (64, 64), (128, 256)):
                for KS in (3, 5):
                    b, n, c = C.c_int32(), C.c_int32(), C.c_int32()
                    _lib.check(L.cmoop_halo_tile_check(B, H, W, Cin, Cout, KS, C.byref(b), C.byref(n), C.byref(c)))
                    if b.value:
                        eligible += 1
                        assert n.value <= b.value <= c.value, ((B, H, W, Cin, Cout, KS), n.value, b.value, c.value)
    assert eligible > 300
    # the layers of the benchmark are eligible; a 128-wide row is not (its halo does not fit the staging slots)
    b, n, c = C.c_int32(), C.c_int32(), C.c_int32()
    _lib.check(L.cmoop_halo_tile_check(64, 101, 40, 64, 64, 5, C.byref(b), C.byref(n), C.byref(c)))
    assert (b.value, n.value) == (11, 10)           # the closed form is one row conservative here
    _lib.check(L.cmoop_halo_tile_check(64, 128, 128, 64, 64, 5, C.byref(b), C.byref(n), C.byref(c)))
    assert b.value == 0


def test_32bit_byte_offset_guard_refuses_oversized_plans():
    """ADVICE r2 / VERDICT r2 item 9: buffer descriptors, row tables and per-row offsets hold BYTES in 32 bits, so every
    tensor a GEMM launch addresses must stay below 2^29 elements; beyond that the hardware range check would return
    zeros silently.  cmoop_plan_check (and net creation / the population calls) refuse instead.  All genes x {101x40,
    128x128} x {batch 64, eval_batch 256, 1024}: the reference's own sizes pass, the oversized ones raise."""
    import itertools
    L = _lib.lib()
    refused = 0
    for gene in itertools.product((16, 32, 64), (3, 5), (0, 1), (1, 2, 3), (1, 4), (0, 1)):
        g = (C.c_int32 * 6)(*gene)
        for variant in (0, 1):
            for (T, F) in ((101, 40), (128, 128)):
                for B in (64, 256, 1024):
                    biggest = B * T * F * gene[0]                     # the first conv's output is the largest tensor of a candidate
                    rc = L.cmoop_plan_check(g, variant, T, F, B)
                    if biggest < 2 ** 29:
                        assert rc == 0, (gene, variant, T, F, B, L.cmoop_last_error())
                    else:
                        assert rc != 0 and b"2^29" in L.cmoop_last_error(), (gene, variant, T, F, B)
                        refused += 1
    assert refused > 0                                                 # 128x128 x 64 filters x eval_batch 1024 = 2^30 elements
    # a shape just above the limit at the layer level (host-only launch plan uses the same guard)
    buf = C.create_string_buffer(200)
    assert L.cmoop_conv_launch_plan(0, 2 ** 13, 256, 256, 16, 16, 3, 1, 0, buf, 200) != 0      # 2^13 * 2^16 * 16 = 2^33
    assert L.cmoop_conv_launch_plan(0, 64, 101, 40, 16, 16, 3, 1, 0, buf, 200) == 0
