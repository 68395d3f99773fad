"""CPU: front-quality metrics (row N3) against the notebook functions' outputs."""
import json
import os

import numpy as np
import pytest

from cmoop_audio_processing_amd import metrics as M


def test_metrics_match_reference_notebook(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "metrics_golden.json")))
    for c in g["cases"]:
        tf = M.true_front(np.vstack([c["A"], c["B"]]))
        assert np.allclose(tf, c["true_front"])
        assert M.generational_distance(c["A"], tf) == pytest.approx(c["gd"], rel=1e-12, abs=1e-15)
        assert M.inverted_gd(c["A"], tf) == pytest.approx(c["igd"], rel=1e-12, abs=1e-15)
        sp = M.spread_metric(c["A"], tf)
        assert (np.isnan(sp) and c["spread"] is None) or sp == pytest.approx(c["spread"], rel=1e-12)
        assert M.coverage_metric(c["A"], c["B"]) == c["c_ab"] and M.coverage_metric(c["B"], c["A"]) == c["c_ba"]
    t = g["tcheby"]
    scores, ranks = M.tchebycheff_rank(t["acc"], t["size"], t["fpr"])
    assert np.allclose(scores, t["scores"], rtol=1e-13) and ranks.tolist() == t["ranks"]


def test_degenerate_inputs():
    assert M.coverage_metric([[0, 0, 0]], []) == 0
    assert np.isnan(M.spread_metric([[0, 0, 0]], [[0, 0, 0], [1, 1, 1]]))
    assert M.generational_distance([[0, 0, 0]], [[0, 0, 0]]) == 0.0
