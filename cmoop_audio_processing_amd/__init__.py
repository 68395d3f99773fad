"""MI355X-native population-fitness evaluator for the SA-NSGA-II audio-NAS loop.

Drop-in for ONE path of sumansamui/CMOOP_Audio_Processing:
``compute_objectives_and_constraints -> evaluate_individual`` (reference
nsga_penalty.py:368-442, sa_nsga_penalty.py:205-253), implemented as
hand-written HIP kernels for gfx950 behind a C ABI (include/cmoop.h).
Importing this package never touches the GPU; the HIP library is loaded on
first use and there is no CPU fallback.
"""
from . import genes  # noqa: F401
from .evaluator import (AudioNASProblem, EvalConfig, PopulationEvaluator, calculate_fpr,  # noqa: F401
                        compute_model_size_mb, compute_objectives_and_constraints, evaluate_individual, install,
                        queued_map, sharded_map)

__all__ = ["genes", "EvalConfig", "PopulationEvaluator", "AudioNASProblem", "install", "evaluate_individual",
           "compute_objectives_and_constraints", "compute_model_size_mb", "calculate_fpr", "sharded_map", "queued_map"]
