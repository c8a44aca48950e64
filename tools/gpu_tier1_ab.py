"""Global-memory grid tier timing on the configurations that use it (one pair, BASELINE configs 3 and 5 come from gpu_configs.py):
usage: MULLS_HIP_LIB=<lib> python tools/gpu_tier1_ab.py"""
import sys, time, warnings
sys.path.insert(0, "."); warnings.filterwarnings("ignore")
import numpy as np
from mulls_amd import abi, synth, lib
ctx = lib.Context(0)
P = abi.kitti_params(converge_translation=0.0, converge_rotation_d=0.0)
pair, T = synth.make_pair(1)
none = {c: None for c in range(abi.NCLASS)}
dense, Td = synth.make_pair(301, n_beams=128, n_az=1875, elev_deg=(-25.0, 15.0), src_counts=none, tgt_counts=none, vertex_count=2000)
P5 = abi.default_params(used_feature_type="111111", max_iter_num=40, converge_translation=0.0, converge_rotation_d=0.0)
for name, pr, PP, nb in (("kitti pair x1", pair, P, 1), ("kitti pair x8", pair, P, 8), ("dense 240k pair", dense, P5, 1)):
    b = ctx.batch([pr] * nb); r = b.run(PP)
    ts = []
    for _ in range(10):
        t = time.perf_counter(); b.run(PP); ts.append(time.perf_counter() - t)
    pf = ctx.profile() if hasattr(ctx, "profile") else None
    print("%-16s median %.3f ms  (code %d, iters %d)" % (name, 1e3 * np.median(ts), r[0].code, r[0].iters), flush=True)
    b.close()
