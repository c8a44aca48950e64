import sys, time, warnings
sys.path.insert(0, "."); sys.path.insert(0, "tests"); warnings.filterwarnings("ignore")
import numpy as np
from mulls_amd import abi, synth, lib
from oracle import pyoracle
import test_gpu_large as tl
ctx = lib.Context(0)
none = {c: None for c in range(abi.NCLASS)}
pair5, T5 = synth.make_pair(301, n_beams=128, n_az=1875, elev_deg=(-25.0, 15.0), src_counts=none, tgt_counts=none, vertex_count=2000)
P5 = abi.default_params(used_feature_type="111111", weight_strategy="1111", max_iter_num=40, dis_thre_unit=1.4, dis_thre_min=0.5,
                        converge_translation=0.0005, converge_rotation_d=0.001, normal_bearing=20.0, sigma_thre=0.35)
pair3, T3 = tl.submap_pair.__wrapped__() if hasattr(tl.submap_pair, "__wrapped__") else (None, None)
for name, pair, P in (("config5 dense 128-beam", pair5, P5),):
    print(name, "src", [len(c) for c in pair.src], "tgt", [len(c) for c in pair.tgt])
    t = time.time(); ro = pyoracle.icp(pair, P, trace_cap=48)[0]; to = time.time() - t
    b = ctx.batch([pair]); rg = b.run(P, trace_cap=48)
    t = time.time(); rg = b.run(P, trace_cap=48); tg = time.time() - t
    ctx.set_profiling(True); b.run(P); pf = ctx.profile(); ctx.set_profiling(False)
    r = rg[0]
    print("  oracle: code %d iters %d %.1f ms | hip: code %d iters %d %.2f ms (resident)  ncorr %s" % (ro.code, ro.iters, to * 1e3, r.code, r.iters, tg * 1e3, list(r.ncorr)))
    print("  hip profile ms: setup %.3f nn %.3f filter %.3f accum %.3f launches %d" % (pf.ms_setup, pf.ms_nn, pf.ms_filter, pf.ms_accum, pf.launches_nn))
    print("  err vs gt", synth.pose_error(r.T_matrix(), T5))
