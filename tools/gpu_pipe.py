"""Scratch: one 1024-pair resident batch, a few runs (for rocprofv3 kernel traces of the pipelined loop)."""
import sys, time, warnings
sys.path.insert(0, ".")
warnings.filterwarnings("ignore")
from mulls_amd import abi, synth, lib
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = lib.Context(0)
pairs = [synth.make_pair(s + 1, n_az=1900)[0] for s in range(4)]
Pb = abi.kitti_params(converge_translation=0.0, converge_rotation_d=0.0)
batch = ctx.batch([pairs[i % 4] for i in range(nb)])
batch.run(Pb)
for prof in (False, True):
    ctx.set_profiling(prof)
    t = time.time()
    for _ in range(3):
        res = batch.run(Pb)
    dt = (time.time() - t) / 3
    pf = ctx.profile()
    print("profiling %d: %.2f ms/run %.0f reg/s | kernels ms: setup %.3f nn %.3f filter %.3f accum %.3f resid %.3f (launches %d) | host step %.3f wait %.3f launch %.3f" % (
        prof, dt * 1e3, nb / dt, pf.ms_setup, pf.ms_nn, pf.ms_filter, pf.ms_accum, pf.ms_residual, pf.launches_nn, pf.ms_host_step, pf.ms_host_wait, pf.ms_host_launch))
print("codes", sorted(set(r.code for r in res)), "iters", sorted(set(r.iters for r in res)))
