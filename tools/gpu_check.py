"""Scratch GPU check: HIP path vs oracle on a few synthetic pairs + a first timing. Run on the GPU box."""
import sys, time, warnings
import numpy as np
sys.path.insert(0, ".")
warnings.filterwarnings("ignore")
from mulls_amd import abi, synth, lib
from oracle import pyoracle

ctx = lib.Context(0)
P = abi.kitti_params(dis_thre_unit=2.4)
pairs = []
for seed in range(4):
    pr, Tgt = synth.make_pair(seed + 1, n_az=1900)
    pairs.append((pr, Tgt))
ok = True
for pr, Tgt in pairs:
    ro = pyoracle.icp(pr, P, trace_cap=32)[0]
    rg_arr = ctx.icp(pr, P, trace_cap=32); rg = rg_arr[0]
    dt, dr = synth.pose_error(rg.T_matrix(), ro.T_matrix())
    print("oracle code %d it %d sigma %.6f | hip code %d it %d sigma %.6f | dT %.3e m %.3e rad | ncorr %s vs %s" % (
        ro.code, ro.iters, ro.sigma, rg.code, rg.iters, rg.sigma, dt, dr, list(ro.ncorr), list(rg.ncorr)))
    for k in range(min(ro.trace_len, rg.trace_len)):
        a, b = ro.trace[k], rg.trace[k]
        if list(a.ncorr) != list(b.ncorr) or list(a.nsrc) != list(b.nsrc):
            print("  iter", k, "counts differ", list(a.ncorr), list(b.ncorr), list(a.nsrc), list(b.nsrc)); ok = False
        dx = np.abs(np.array(a.x[:]) - np.array(b.x[:])).max()
        if dx > 1e-9: print("  iter", k, "dx", dx)
print("PARITY", "OK" if ok else "MISMATCH")

# timing: batch of replicated pairs, fixed 20 iterations
Pb = abi.kitti_params(converge_translation=0.0, converge_rotation_d=0.0)
import sys as _s
modes = [(3, "grid_lds"), (2, "grid_global"), (1, "brute")]
for mode, name in modes:
  ctx.set_nn_mode(mode); print("== nn tier:", name)
  for nb in (1, 16, 64, 256, 1024):
    if mode != 3 and nb > 256: continue
    batch = ctx.batch([pairs[i % 4][0] for i in range(nb)])
    res = batch.run(Pb)
    t = time.time(); reps = 3
    for _ in range(reps):
        res = batch.run(Pb)
    dt = (time.time() - t) / reps
    print("batch %4d: %.2f ms/run  %.1f reg/s  iters %d code %d" % (nb, dt * 1e3, nb / dt, res[0].iters, res[0].code))
    if nb >= 256:
        ctx.set_profiling(True); batch.run(Pb); pf = ctx.profile(); ctx.set_profiling(False)
        print("   profile ms: setup %.3f nn %.3f filter %.3f accum %.3f resid %.3f launches %d | host: step %.3f wait %.3f launch %.3f" % (
            pf.ms_setup, pf.ms_nn, pf.ms_filter, pf.ms_accum, pf.ms_residual, pf.launches_nn, pf.ms_host_step, pf.ms_host_wait, pf.ms_host_launch))
        batch.run(Pb); pf = ctx.profile()
        print("   unprofiled run host ms: step %.3f wait %.3f launch %.3f" % (pf.ms_host_step, pf.ms_host_wait, pf.ms_host_launch))
    batch.close()
