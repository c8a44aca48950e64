"""Wider sweep of the degenerate-input differential test (tests/test_gpu_fuzz.py) over many seeds; prints every mismatch."""
import sys, traceback, warnings
sys.path.insert(0, "."); sys.path.insert(0, "tests"); warnings.filterwarnings("ignore")
import numpy as np
from mulls_amd import lib
from oracle import pyoracle
from test_gpu_fuzz import degenerate_pair, random_params
from test_gpu_icp import compare
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
bad = 0
for mode in (3, 2, 1):
    ctx = lib.Context(0); ctx.set_nn_mode(mode)
    for kind in ("duplicates", "one_cell", "far_origin", "collinear", "sparse_far", "ragged"):
        for seed in range(nseeds):
            rng = np.random.default_rng(seed * 7919 + 13)
            pair = degenerate_pair(rng, kind); P = random_params(rng); P.apply_motion_undistortion = 0
            ro = pyoracle.icp(pair, P, trace_cap=32)[0]; rg = ctx.icp(pair, P, trace_cap=32)[0]
            try:
                compare(ro, rg, x_tol=1e-6)
            except AssertionError:
                bad += 1
                tb = traceback.format_exc().strip().split("\n")
                print("MISMATCH mode %d %s seed %d: %s | %s" % (mode, kind, seed, tb[-3].strip()[:150], tb[-1][:200]), flush=True)
    ctx.close()
print("done, mismatches:", bad)
