"""CPU oracle for the population-fitness hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU, the algorithm of the reference's
``evaluate_individual`` / ``compute_objectives_and_constraints`` path
(/root/reference/nsga_penalty.py:225-442, sa_nsga_penalty.py:137-253).  It
exists to CHECK the HIP implementation; it is never the thing shipped or
measured.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it.  The product package
(``cmoop_audio_processing_amd``) must not import it and fails loudly when its
HIP library is missing.

Where the arithmetic really lives
---------------------------------
The reference delegates every FLOP of this path to third-party code that is NOT
under /root/reference and is not installed in this image:
``tensorflow==2.18.0`` + ``keras==3.6.0`` (requirements.txt:152,72) for model
build / fit / evaluate / predict, and ``scikit-learn==1.5.2`` (:138) for
``confusion_matrix`` and ``StandardScaler``.  The oracle therefore restates the
*published* Keras-3 semantics (glorot-uniform init, TF "SAME" padding, BN
eps=1e-3 momentum=0.99 biased variance, inverted dropout, clipped sparse CE on
probabilities, Keras-form Adam, EarlyStopping wait/patience logic) on top of
torch-CPU fp32 primitives, with gradients from torch autograd -- i.e.
independent of the hand-written HIP backward kernels.

Pinning status
--------------
* PINNED by outputs of reference code executed in the build container
  (pure-Python/NumPy/sklearn functions AST-extracted from the reference files by
  ``tests/golden/make_golden.py``; fixtures in ``tests/golden/*.json``):
  ``calculate_fpr`` V1 / V3 and the ``y_true`` quirk of nsga_penalty.py:387, the
  ``compute_model_size_mb`` arithmetic, objective/CV assembly, the [0,1]^6 gene
  codec, ``get_lambda``.
* PINNED by closed-form known answers: parameter counts.
* PARITY UNPINNED: training dynamics, accuracy and FPR *values* of a trained
  net (the reference holds no tests, golden vectors or seeds for them and
  Keras/TF cannot run here), and the audio front end (absent from the
  reference altogether; librosa==0.11.0 is pinned in requirements.txt:80 but
  has no call site, so the oracle restates librosa's published mel-spectrogram
  algorithm).  For these the oracle is the definition the HIP path is held to.
* NO REFERENCE COUNTERPART: ``OracleConfig.compute = "bf16"`` (oracle/net.py) restates the build's own opt-in
  bf16-train mode (BASELINE.json configs[4]); the reference trains in fp32 only.
"""
