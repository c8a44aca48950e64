"""Phase clocks of k_cert's one-pass class walk (cert_class_flat; MULLS_OPT_DEBUG_STOP = 20) on the bench workload: usage gpu_cert_phases.py [pairs]"""
import sys, warnings, time
sys.path.insert(0, "."); warnings.filterwarnings("ignore")
import bench
from mulls_amd import abi, lib
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
scenes = bench.build_scenes(64, False, 16)
pairs = [bench.global_pair(scenes, g) for g in range(nb)]
P = bench.bench_params()
ctx = lib.Context(0); ctx.set_nn_mode(3)
ctx.set_option(abi.OPT_SPLIT_MAX_PAIRS, 0)
b = ctx.batch(pairs)
res = abi.make_result_array(nb)
b.run(P, results=res)
ctx.set_option(abi.OPT_DEBUG_STOP, 20)
ctx.set_profiling(1)
t = time.time(); b.run(P, results=res); dt = time.time() - t
pf = ctx.profile()
wgs = max(pf.icp_fused_ms[5], 1.0)
names = ["set-up: descriptors, first loads issued, duplicate table armed", "k-candidate certificates of the leftover list", "rigid step + certificates + stores (three trips)", "leftover search + match count", "duplicate rule + rejection chain + counters"]
print("%d pairs: %.2f ms wall, search kernels %.2f ms; %d one-pass workgroups" % (nb, dt * 1e3, pf.ms_nn, wgs))
tot = sum(pf.icp_fused_ms[k] for k in range(5))
for k in range(5):
    print("  %-46s %7.2f us per workgroup  (%.1f %%)" % (names[k], pf.icp_fused_ms[k] / wgs * 1e3, 100 * pf.icp_fused_ms[k] / tot))
print("  total %.2f us per workgroup" % (tot / wgs * 1e3))
ks = pf.icp_search_ms
print("  points the plain certificate left over %d, given the k-candidate look %d, certified by it %d (%.1f %%), left to search %d" % (ks[0], ks[1], ks[2], 100.0 * ks[2] / max(ks[1], 1), ks[3]))
jobs = max(pf.icp_phase_ms[4], 1.0)
hn = ["prefetch + staging of the target cloud + table init", "chunk set-up: queries, cost sort", "search rounds", "duplicate rule + rejection chain + counters"]
ht = sum(pf.icp_phase_ms[k] for k in range(4))
print("heavy pass (k_nn_lds): %d class clouds over the run's iterations" % jobs)
for k in range(4):
    print("  %-52s %7.2f us per class cloud  (%.1f %%)" % (hn[k], pf.icp_phase_ms[k] / jobs * 1e3, 100 * pf.icp_phase_ms[k] / max(ht, 1e-9)))
print("  total %.2f us per class cloud" % (ht / jobs * 1e3))
