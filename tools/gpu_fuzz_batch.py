"""Degenerate-input differential sweep through the BATCH path: every block is one lock-step batch of 240 different degenerate
pairs (class-level LDS jobs: on-chip duplicate rule, fused rejection chain, hints, cost order, correspondence records) under one
random option point, each pair compared with the oracle.  usage: gpu_fuzz_batch.py [blocks]"""
import sys, traceback, warnings
sys.path.insert(0, "."); sys.path.insert(0, "tests"); warnings.filterwarnings("ignore")
import numpy as np
from mulls_amd import lib
from oracle import pyoracle
from test_gpu_fuzz import degenerate_pair, random_params
from test_gpu_icp import compare
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 6
kinds = ("duplicates", "one_cell", "far_origin", "collinear", "sparse_far", "ragged")
bad = 0
ctx = lib.Context(0)
for b in range(blocks):
    rng = np.random.default_rng(b * 104729 + 7)
    P = random_params(rng); P.apply_motion_undistortion = 0
    pairs = [degenerate_pair(rng, kinds[i % len(kinds)]) for i in range(240)]
    rg = ctx.icp_batch(pairs, P)
    for i, pair in enumerate(pairs):
        ro = pyoracle.icp(pair, P)[0]
        try:
            compare(ro, rg[i], check_trace=False, x_tol=1e-6)
        except AssertionError:
            bad += 1
            tb = traceback.format_exc().strip().split("\n")
            print("MISMATCH block %d pair %d (%s): %s | %s" % (b, i, kinds[i % len(kinds)], tb[-3].strip()[:150], tb[-1][:200]), flush=True)
    print("block", b, "done, used", P.used_feature_type, "shooting", P.normal_shooting_on, "iters", P.max_iter_num, flush=True)
ctx.close()
print("done, mismatches:", bad)
