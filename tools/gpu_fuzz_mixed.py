"""Degenerate-input differential sweep through MIXED batches (auto mode): every pair draws its six class-cloud sizes from the ranges of all three search
tiers (brute force / LDS tier / global-memory tier; sources on both sides of the class-level job limit), so that one batch holds class-level and
chunk-level jobs of both grid tiers on pathological data (ties everywhere, one cell, 1e5 m offsets, collinear, nothing in range) and on noisy copies
that run their iterations (tests/test_gpu_fuzz.py: mixed_tier_block).  Each pair is compared with the oracle.
usage: gpu_fuzz_mixed.py [blocks] [pairs per block]"""
import sys, time, traceback, warnings
sys.path.insert(0, "."); sys.path.insert(0, "tests"); warnings.filterwarnings("ignore")
import numpy as np
from mulls_amd import lib
from oracle import pyoracle
from test_gpu_fuzz import mixed_tier_block
from test_gpu_icp import compare
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 4
per = int(sys.argv[2]) if len(sys.argv) > 2 else 12
bad = 0
ctx = lib.Context(0)
for b in range(blocks):
    rng = np.random.default_rng(b * 15485863 + 11)
    P, pairs = mixed_tier_block(rng, per)
    t0 = time.time()
    rg = ctx.icp_batch(pairs, P)
    t1 = time.time()
    for i, pair in enumerate(pairs):
        ro = pyoracle.icp(pair, P)[0]
        try:
            compare(ro, rg[i], check_trace=False, x_tol=1e-6)
        except AssertionError:
            bad += 1
            tb = traceback.format_exc().strip().split("\n")
            print("MISMATCH block %d pair %d: %s | %s" % (b, i, tb[-3].strip()[:150], tb[-1][:200]), flush=True)
    its = [int(r.iters) for r in rg]
    print("block %d done: used %s max_iter %d keep_less %d, iterations run %s, codes %s, device %.2f s, oracle %.1f s"
          % (b, P.used_feature_type.decode(), P.max_iter_num, P.keep_less_source_points, its, sorted(set(int(r.code) for r in rg)), t1 - t0, time.time() - t1), flush=True)
ctx.close()
print("done, mismatches:", bad)
