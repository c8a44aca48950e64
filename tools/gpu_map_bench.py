"""Local-map maintenance and scan-to-map registration with a device-resident target vs the CPU oracle.
usage: gpu_map_bench.py [n_az of the map scan, default 7500 (~1 M returns with 128 beams)]"""
import sys, time, warnings
sys.path.insert(0, "."); warnings.filterwarnings("ignore")
import numpy as np
from mulls_amd import abi, synth, lib
from oracle import pyoracle

n_az = int(sys.argv[1]) if len(sys.argv) > 1 else 7500
big = {c: None for c in range(5)}
pair, T_gt = synth.make_pair(7, n_beams=128, n_az=n_az, elev_deg=(-25.0, 15.0), src_counts={abi.GROUND: 800, abi.PILLAR: 400, abi.FACADE: 1200,
                                                                                               abi.BEAM: 300, abi.ROOF: 200}, tgt_counts=big, vertex_count=2000)
map0 = [pair.tgt[c] for c in range(6)]
frame = [pair.src[c] for c in range(6)]
print("map clouds:", [len(c) for c in map0], "frame:", [len(c) for c in frame])
ctx = lib.Context(0)
P = abi.map_params(max_num_pts=10**7, kept_vertex_num=10**6, local_map_radius=100.0, map_based_dynamic_removal_on=1, tree_mode=1, tree_used="111110")
pose1 = np.linalg.inv(T_gt)
t = time.time(); mo, fo, ro = pyoracle.map_update(map0, np.eye(4), frame, pose1, P); t_or = time.time() - t
dev = ctx.local_map(map0, np.eye(4))
dev.update(frame, pose1, P)  # warm-up (allocations, module load)
dev.set(map0, np.eye(4))
t = time.time(); rg = dev.update(frame, pose1, P); t_gpu = time.time() - t
assert list(rg.n) == list(ro.n) and list(rg.frame_n) == list(ro.frame_n) and list(rg.local_bound) == list(ro.local_bound)
print("update_local_map with dynamic removal: oracle %.1f ms | device %.2f ms (report %.2f ms) | sizes %s" % (t_or * 1e3, t_gpu * 1e3, rg.ms_total, list(rg.n)))

# scan-to-map registration: resident target vs re-uploaded target vs oracle
Pr = abi.kitti_params(converge_translation=0.0, converge_rotation_d=0.0, used_feature_type="111110")
dev.set(map0, np.eye(4))
host_pair = abi.PairData(map0, frame, init_guess=pair.init_guess, tgt_bound=pair.tgt_bound)
for name, fn in (("resident target", lambda: dev.icp(frame, Pr, init_guess=pair.init_guess, tgt_bound=pair.tgt_bound)),
                 ("uploaded target", lambda: ctx.icp(host_pair, Pr))):
    fn(); ts = []
    for _ in range(5):
        t = time.time(); r = fn()[0]; ts.append(time.time() - t)
    print("mm_lls_icp vs %d-point map, %s: %.2f ms (code %d, %d iters)" % (sum(len(c) for c in map0), name, np.median(ts) * 1e3, r.code, r.iters))
t = time.time(); r0 = pyoracle.icp(host_pair, Pr)[0]; print("oracle: %.1f ms (code %d, %d iters)" % ((time.time() - t) * 1e3, r0.code, r0.iters))
dt, dr = synth.pose_error(r.T_matrix(), r0.T_matrix()); print("dT vs oracle %.2e m %.2e rad" % (dt, dr))
ctx.set_profiling(True); dev.icp(frame, Pr, init_guess=pair.init_guess, tgt_bound=pair.tgt_bound); pf = ctx.profile(); ctx.set_profiling(False)
print("profile ms: setup %.3f nn %.3f filter %.3f accum %.3f resid %.3f launches %d | host step %.3f wait %.3f launch %.3f" % (
    pf.ms_setup, pf.ms_nn, pf.ms_filter, pf.ms_accum, pf.ms_residual, pf.launches_nn, pf.ms_host_step, pf.ms_host_wait, pf.ms_host_launch))
