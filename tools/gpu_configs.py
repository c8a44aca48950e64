"""BASELINE.json configs other than the bench line, measured the same way (clouds resident, fixed iteration counts):
#3 scan-to-localmap against a ~1 M-point map (20 iterations), #5 synthetic 128-beam 240 k-point scans, all six classes,
40 iterations.  Oracle timed next to each.  usage: gpu_configs.py"""
import sys, time, warnings
sys.path.insert(0, "."); warnings.filterwarnings("ignore")
import numpy as np
from mulls_amd import abi, synth, lib
from oracle import pyoracle

ctx = lib.Context(0)
none = {c: None for c in range(5)}

def timed(fn, reps=5):
    fn(); ts = []
    for _ in range(reps):
        t = time.time(); r = fn(); ts.append(time.time() - t)
    return float(np.median(ts)) * 1e3, r

# ---- config 3
pair, T = synth.make_pair(7, n_beams=128, n_az=7500, elev_deg=(-25.0, 15.0), src_counts={0: 800, 1: 400, 2: 1200, 3: 300, 4: 200}, tgt_counts=none, vertex_count=2000)
P3 = abi.kitti_params(converge_translation=0.0, converge_rotation_d=0.0, used_feature_type="111110")
dev = ctx.local_map(pair.tgt, np.eye(4))
ms_dev, r = timed(lambda: dev.icp(pair.src, P3, init_guess=pair.init_guess, tgt_bound=pair.tgt_bound)[0])
ms_up, _ = timed(lambda: ctx.icp(pair, P3)[0])
t = time.time(); ro = pyoracle.icp(pair, P3)[0]; ms_or = (time.time() - t) * 1e3
dt, dr = synth.pose_error(r.T_matrix(), ro.T_matrix())
print("config 3  scan-to-localmap, map %d points (%s), source %s, 20 iterations" % (sum(len(c) for c in pair.tgt), [len(c) for c in pair.tgt[:5]], [len(c) for c in pair.src[:5]]))
print("          device-resident map %.2f ms | map uploaded per call %.2f ms | oracle %.1f ms | dT vs oracle %.1e m %.1e rad | code %d iters %d" % (ms_dev, ms_up, ms_or, dt, dr, r.code, r.iters))
Pm = abi.map_params(max_num_pts=10**7, kept_vertex_num=10**6, local_map_radius=100.0, map_based_dynamic_removal_on=1, tree_mode=1, tree_used="111110")
pose1 = np.linalg.inv(T)
dev.update(pair.src, pose1, Pm); dev.set(pair.tgt, np.eye(4))
t = time.time(); rep = dev.update(pair.src, pose1, Pm); ms_upd = (time.time() - t) * 1e3
t = time.time(); pyoracle.map_update(pair.tgt, np.eye(4), pair.src, pose1, Pm); ms_upd_o = (time.time() - t) * 1e3
print("          update_local_map with dynamic removal: device %.2f ms | oracle %.1f ms" % (ms_upd, ms_upd_o))
dev.close()

# ---- config 5
none6 = {c: None for c in range(abi.NCLASS)}
pair5, T5 = synth.make_pair(301, n_beams=128, n_az=1875, elev_deg=(-25.0, 15.0), src_counts=none6, tgt_counts=none6, vertex_count=2000)
P5 = abi.default_params(used_feature_type="111111", weight_strategy="1111", max_iter_num=40, dis_thre_unit=1.4, dis_thre_min=0.5,
                        converge_translation=0.0, converge_rotation_d=0.0, normal_bearing=20.0, sigma_thre=0.35)
b = ctx.batch([pair5])
ms5, r5 = timed(lambda: b.run(P5)[0])
t = time.time(); ro5 = pyoracle.icp(pair5, P5)[0]; ms5o = (time.time() - t) * 1e3
dt, dr = synth.pose_error(r5.T_matrix(), ro5.T_matrix())
ctx.set_profiling(True); b.run(P5); pf = ctx.profile(); ctx.set_profiling(False)
print("config 5  128-beam scans (%d / %d returns), all six classes, source %s, target %s, 40 iterations" % (pair5.n_raw[0], pair5.n_raw[1], [len(c) for c in pair5.src], [len(c) for c in pair5.tgt]))
print("          device %.2f ms (%.1f registrations/s, one pair at a time) | oracle %.1f ms | dT vs oracle %.1e m %.1e rad | code %d iters %d" % (ms5, 1e3 / ms5, ms5o, dt, dr, r5.code, r5.iters))
print("          kernel ms: setup %.3f search %.3f filter %.3f accumulate %.3f residual %.3f" % (pf.ms_setup, pf.ms_nn, pf.ms_filter, pf.ms_accum, pf.ms_residual))
