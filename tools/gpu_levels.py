"""Step times of the bench batch from the start of a process, by profiling level (diagnostics): usage gpu_levels.py <pairs> <level> [steps]"""
import sys, warnings, time
sys.path.insert(0, "."); warnings.filterwarnings("ignore")
import bench
from mulls_amd import abi, lib
nb, level = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
scenes = bench.build_scenes(64, False, 16)
pairs = [bench.global_pair(scenes, g) for g in range(nb)]
P = bench.bench_params()
ctx = lib.Context(0)
b = ctx.batch(pairs)
res = abi.make_result_array(nb)
ctx.set_profiling(level)
ts = []
t00 = time.perf_counter()
for _ in range(steps):
    t = time.perf_counter(); b.run(P, results=res); ts.append(time.perf_counter() - t)
    pf = ctx.profile()
    if ts[-1] > 0.04 or _ == 10:
        print("  step %d: %.1f ms; setup %.2f search %.2f filter %.2f accum %.2f residual %.2f | host launch %.2f wait %.2f stage %.2f" % (
            _, ts[-1] * 1e3, pf.ms_setup, pf.ms_nn, pf.ms_filter, pf.ms_accum, pf.ms_residual, pf.ms_host_launch, pf.ms_host_wait, pf.ms_stage))
print("level %d, %d steps from process start: %s" % (level, steps, " ".join("%.0f" % (1e3 * t) for t in ts)))
