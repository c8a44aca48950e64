import sys, warnings
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
warnings.filterwarnings("ignore")
from conftest import planes_scene, transformed_copy
from mulls_amd import abi, synth, lib
from oracle import pyoracle
ctx = lib.Context(0); ctx.set_nn_mode(2)
rng = np.random.default_rng(7)
tgt = planes_scene(rng)
good = abi.PairData(tgt, transformed_copy(tgt, np.linalg.inv(synth.se3(0.1, 0.05, 0.0, 0, 0, 0.01))))
far = abi.PairData(tgt, transformed_copy(tgt, synth.se3(200.0, 0, 0)))
empty = abi.PairData(tgt, [None] * 6)
P = abi.default_params(used_feature_type="111000", apply_intersection_filter=0)
pairs = [good, far, good, empty, far]
rb = ctx.icp_batch(pairs, P, trace_cap=24)
for i, pr in enumerate(pairs):
    ro = pyoracle.icp(pr, P, trace_cap=24)[0]
    print(i, "oracle", ro.code, ro.iters, list(ro.ncorr), "| hip", rb[i].code, rb[i].iters, list(rb[i].ncorr), list(rb[i].nsrc0), list(rb[i].ntgt0))
for i, pr in enumerate(pairs[:2]):
    r1 = ctx.icp(pr, P, trace_cap=24)[0]
    print("single", i, r1.code, r1.iters, list(r1.ncorr))
