"""Conv launch shapes of the production-shape parity tests (tests/test_gpu_production_shapes.py) -- kept importable
without a GPU so that the host-only coverage test (tests/test_host_cpu.py) can compare the launch-path variants these
cases exercise with the variants the trainer uses for every gene of the search space."""

# B, H, W, Cin, Cout, KS, stride -- every conv launch shape of the bench's population (genes of random.Random(0),
# topology A, 101x40, batch 64) whose implicit GEMM carries >= 1 % of the generation, plus the skip projections
PRODUCTION_CONVS = [
    (64, 101, 40, 64, 64, 5, 1),     # 128x64 tile, 16-deep chunks (many-wave grid)
    (64, 101, 40, 64, 64, 3, 1),
    (64, 101, 40, 32, 32, 5, 1),     # 128x32 tile on the LDS-DMA operand path
    (64, 101, 40, 32, 32, 3, 1),
    (64, 101, 40, 16, 16, 5, 1),     # 128x16 tile, Cin = 16 (16-deep chunks)
    (64, 101, 40, 16, 16, 3, 1),
    (64, 51, 20, 64, 128, 5, 1),     # 128x128 forward, wgrad<128, 64/128>
    (64, 51, 20, 128, 128, 5, 1),    # the dominant launch of the bench: igemm_fwd_kernel<128,128,32,2,0>, wgrad<128,128>
    (64, 51, 20, 128, 128, 3, 1),
    (64, 51, 20, 32, 64, 5, 1),
    (64, 51, 20, 16, 32, 3, 1),
    (64, 26, 10, 128, 256, 5, 1),
    (64, 26, 10, 256, 256, 5, 1),    # under-filled grid: split-K slabs + combine
    (64, 26, 10, 256, 256, 3, 1),
    (64, 26, 10, 64, 128, 3, 1),
    (64, 13, 5, 256, 512, 5, 1),
    (64, 13, 5, 512, 512, 5, 1),     # K = 12 800, 33 row tiles: split-K
    (64, 13, 5, 512, 512, 3, 1),
    (64, 13, 5, 128, 256, 3, 1),
    (64, 51, 20, 64, 128, 1, 2),     # skip projections: 1x1 stride 2, scatter-accumulating dgrad
    (64, 26, 10, 128, 256, 1, 2),
    (64, 13, 5, 256, 512, 1, 2),
    (64, 51, 20, 16, 32, 1, 2),
    (37, 101, 40, 64, 64, 5, 1),     # a partial last batch (Keras keeps it): other slice counts, same tiles
    (256, 26, 10, 256, 256, 5, 1),   # the inference batch (eval_batch 256) of a deep layer: un-split grid
    (37, 51, 20, 16, 32, 5, 1),      # partial batches move some layers onto other tiles / split-K: covered too
    (37, 51, 20, 32, 64, 5, 1),
    (37, 51, 20, 32, 32, 3, 1),
    (64, 26, 10, 32, 64, 3, 1),
    (37, 26, 10, 32, 64, 3, 1),
    (37, 26, 10, 64, 64, 3, 1),      # too few K chunks for the balanced partition: the uniform split-K path of the 128x64 tile
    # halo-tiled direct convolution (halo_fwd_kernel): tiles that cross image boundaries (26x10 = 260, 13x5 = 65 pixels per
    # image against 128-pixel tiles), several 64-column workgroups per tile, the inference batch
    (256, 51, 20, 64, 64, 3, 1),
    (256, 13, 5, 256, 512, 3, 1),
    (64, 51, 20, 16, 32, 5, 1),      # 256 x 32 halo tile, one 16-channel chunk
    (64, 51, 20, 32, 32, 3, 1),
    (64, 32, 32, 32, 64, 3, 1),      # BASELINE configs[3] (BirdCLEF-shaped 128x128 patches) after two / three pools: other row widths
    (64, 16, 16, 64, 128, 5, 1),     # (W = 32: halo pitch 40; W = 16, k5: pitch 24, tiles of 8 image rows)
    (51, 51, 20, 32, 64, 3, 1),      # the partial batch of tests/test_gpu_net.py::test_wgrad_workspace_partial_batches_advice_r1
    (51, 51, 20, 64, 64, 3, 1),      # (ragged row slices of the halo weight gradient)
    # the skip projections at the inference batch: many-wave grids of the implicit GEMM's 128-row tiles
    (256, 51, 20, 16, 32, 1, 2),
    (256, 51, 20, 32, 64, 1, 2),
    (256, 51, 20, 64, 128, 1, 2),
    (256, 26, 10, 128, 256, 1, 2),
]
