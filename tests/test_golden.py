"""The oracle and the scene generator still produce the committed golden vectors
(tests/golden/oracle_small.npz, made by tests/golden/make_golden.py)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def test_oracle_reproduces_golden_vectors(oracle):
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_golden
    want = np.load(os.path.join(HERE, "golden", "oracle_small.npz"))
    got = make_golden.compute()
    for k in ("xyz", "viewmatrix", "projmatrix", "radii", "tiles_touched", "point_list", "ranges", "n_contrib", "dpix"):
        assert np.array_equal(got[k], want[k]), k          # inputs and every integer result: exact
    for k in ("means2D", "conic_opacity", "rgb"):
        assert np.array_equal(got[k], want[k]), k          # IEEE +,-,*,/,sqrt only: exact on any conforming host
    for k in ("out_color", "final_T"):
        assert np.abs(got[k] - want[k]).max() <= 1e-6, k   # libm expf may differ in the last ulp between hosts
    for k in want.files:
        if k.startswith("dL_"):
            d = np.abs(got[k].astype(np.float64) - want[k]).sum() / (np.abs(want[k]).sum() + 1e-30)
            assert d <= 1e-5, (k, d)
