#!/usr/bin/env python3
"""tests/golden/f_planar_close_eigenvalues.npz: the planar two-view problem of tests/test_host_abi.py
::test_loransac_f_planar_problem_with_close_eigenvalues (case 1189 of `tools/sweep_ransac_ref.py 1200 2`)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from common import synth_two_view
pts, _ = synth_two_view(2 * 100000 + 1189, n_in=544, n_out=960, planar_frac=0.9, noise=1.0)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "f_planar_close_eigenvalues.npz"), pts=pts, seed=766, error_type=0)
print(pts.shape)
