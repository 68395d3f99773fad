"""CPU: dataset adaptors (row N4)."""
import numpy as np
import pytest

from cmoop_audio_processing_amd import datasets as D, nsga


def test_npy_loader_shapes_and_label_axis(tmp_path):
    rs = np.random.RandomState(0)
    for n, name in ((12, "train"), (4, "test"), (5, "val")):
        np.save(tmp_path / f"X_{name}.npy", rs.randn(n, 7, 3).astype(np.float32))
        np.save(tmp_path / f"y_{name}.npy", rs.randint(0, 10, n))
    Xtr, Xte, Xva, ytr, yte, yva = D.load_npy_splits(str(tmp_path))
    assert Xtr.shape == (12, 7, 3) and Xte.shape == (4, 7, 3) and Xva.shape == (5, 7, 3)
    assert ytr.shape == (12, 1) and yte.shape == (4, 1) and yva.shape == (5, 1)      # nsga_penalty.py:74-76
    with pytest.raises(FileNotFoundError):
        D.load_npy_splits(str(tmp_path / "nope"))


def test_stratified_split_fractions_and_determinism():
    rs = np.random.RandomState(1)
    y = np.repeat(np.arange(11), 40)
    X = rs.randn(len(y), 5, 4).astype(np.float32)
    a = D.stratified_50_25_25(X, y)
    b = D.stratified_50_25_25(X, y)
    assert [len(v) for v in a[1::2]] == [220, 110, 110]
    assert all(np.array_equal(p, q) for p, q in zip(a, b))                 # random_state=42
    for ys in a[1::2]:
        assert len(set(np.bincount(ys).tolist())) == 1                      # perfectly stratified here


def test_two_stage_population_roundtrip(tmp_path):
    recs = [{"Accuracy": 0.93, "Size_MB": 1.2, "FPR": 0.01, "filters": 32, "kernel_size": 3, "use_bn": True,
             "residual_blocks": 2, "fc_layers": 2, "use_dropout": False},
            {"Accuracy": 0.80, "Size_MB": 3.0, "FPR": 0.12, "filters": 64, "kernel_size": 5, "use_bn": 0,
             "residual_blocks": 3, "fc_layers": 4, "use_dropout": 1}]
    p = tmp_path / "final.csv"
    D.records_to_csv(str(p), recs)
    pop = D.read_two_stage_population(str(p), 0.90, 2.5, 0.09)
    assert pop[0]["hparams"] == {"filters": 32, "kernel_size": 3, "use_bn": True, "residual_blocks": 2, "fc_layers": 2, "use_dropout": False}
    assert pop[0]["objs"] == [-0.93, 1.2, 0.01] and pop[0]["CV"] == 0
    assert pop[1]["hparams"]["use_bn"] is False and pop[1]["hparams"]["use_dropout"] is True
    assert pop[1]["CV"] == pytest.approx((0.90 - 0.80) + (3.0 - 2.5) + (0.12 - 0.09))
    assert nsga.fast_non_dominated_sort(pop, 50.0)[0] == [0]


def test_h5_loader_reports_missing_dependency(tmp_path):
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError):
            D.load_mel_h5(str(tmp_path / "x.h5"))
