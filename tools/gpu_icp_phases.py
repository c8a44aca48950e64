"""Phase times of the device-resident loop on the bench workload: usage gpu_icp_phases.py [pairs]"""
import sys, warnings, time
sys.path.insert(0, "."); warnings.filterwarnings("ignore")
import bench
from mulls_amd import abi, lib
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
scenes = bench.build_scenes(64, False, 16)
pairs = [bench.global_pair(scenes, g) for g in range(nb)]
P = bench.bench_params()
ctx = lib.Context(0)
b = ctx.batch(pairs)
res = abi.make_result_array(nb)
b.run(P, results=res)
ctx.set_profiling(True)
t = time.time(); b.run(P, results=res); dt = time.time() - t
pf = ctx.profile()
names = ["search", "counts", "normal eq", "solve", "residual", "total"]
print("%d pairs: %.2f ms wall, k_icp %.2f ms, setup %.2f ms, %.0f reg/s" % (nb, dt * 1e3, pf.ms_nn, pf.ms_setup, nb / dt))
for k in range(6):
    print("  %-10s %8.1f us per pair  (%.1f %%)" % (names[k], pf.icp_phase_ms[k] / nb * 1e3, 100 * pf.icp_phase_ms[k] / max(pf.icp_phase_ms[5], 1e-9)))
print("  fused pass by stage (us per pair): setup %.0f, stage1 %.0f, leftovers %.0f, stage3 %.0f, stage4 %.0f" % tuple(pf.icp_fused_ms[k] / nb * 1e3 for k in range(5)))
print("  search by iteration (us per pair):", " ".join("%.0f" % (pf.icp_search_ms[k] / nb * 1e3) for k in range(20)))
print("  codes", sorted(set(r.code for r in res)), "iters", sorted(set(r.iters for r in res)))
