"""mulls_pipe: calls from host buffers in flight on alternating contexts — wall per call by depth, and each lane's own staging / gather / kernel times under the overlap
(usage: gpu_pipe_calls.py [pairs] [calls])"""
import sys, time, warnings, os
sys.path.insert(0, "."); warnings.filterwarnings("ignore")
import bench
from mulls_amd import abi, lib
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n_calls = int(sys.argv[2]) if len(sys.argv) > 2 else 12
try:
    print("host: %d cpus, cgroup cpu.max = %s" % (os.cpu_count(), open("/sys/fs/cgroup/cpu.max").read().strip()))
except Exception:
    print("host: %d cpus" % os.cpu_count())
scenes = bench.build_scenes(64, False, 16)
pairs = [bench.global_pair(scenes, g) for g in range(nb)]
P = bench.bench_params()
ctx = lib.Context(0); ctx.set_option(abi.OPT_LEAN_STAGING, 1)
m = (abi.make_pair_array(pairs), abi.make_result_array(nb))
ctx.icp_batch(pairs, P, marshalled=m)
ts = []
for _ in range(5):
    ctx.icp_batch(pairs, P, marshalled=m); ts.append(ctx.last_call_s)
pf = ctx.profile()
print("serial: %.2f ms per call (staging %.2f of which host gather %.2f; setup %.2f search %.2f accumulate %.2f)" % (sorted(ts)[2] * 1e3, pf.ms_stage, pf.ms_stage_pack, pf.ms_setup, pf.ms_nn, pf.ms_accum))
ctx.close()
for depth in (1, 2, 3, 4):
    pipe = lib.Pipe(0, depth); pipe.set_option(abi.OPT_LEAN_STAGING, 1)
    marsh = [(abi.make_pair_array(pairs), abi.make_result_array(nb)) for _ in range(depth + 1)]
    for k in range(depth):
        pipe.end(pipe.begin(pairs, P, marshalled=marsh[k]))
    pend = []; t0 = time.perf_counter()
    for k in range(n_calls):
        pend.append(pipe.begin(pairs, P, marshalled=marsh[k % len(marsh)]))
        if len(pend) > depth - 1:
            pipe.end(pend.pop(0))
    while pend:
        pipe.end(pend.pop(0))
    wall = time.perf_counter() - t0
    lanes = [pipe.lane_profile(l) for l in range(depth)]
    print("depth %d: %.2f ms per call = %.1f k registrations/s | lanes' last calls: %s" % (depth, wall / n_calls * 1e3, n_calls * nb / wall / 1e3,
          "; ".join("staging %.2f (gather %.2f)" % (l.ms_stage, l.ms_stage_pack) for l in lanes)))
    pipe.close()
