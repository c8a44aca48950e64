"""Certificates on real scans against the synthetic bench workload, iteration by iteration (MULLS_OPT_DEBUG_STOP = 20 counters; a run of k iterations minus a run of k - 1):
live source points, points the plain certificate leaves over, points the k-candidate look certifies, points searched, and the search kernels' time.
usage: gpu_real_vs_synth.py [pairs]"""
import sys, warnings, copy
sys.path.insert(0, "."); warnings.filterwarnings("ignore")
import numpy as np
import bench
from mulls_amd import abi, lib
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024


def series(name, pairs, P, iters):
    ctx = lib.Context(0); ctx.set_nn_mode(3)
    ctx.set_option(abi.OPT_SPLIT_MAX_PAIRS, 0)
    ctx.set_option(abi.OPT_KCERT_MIN, 1)
    b = ctx.batch(pairs)
    res = abi.make_result_array(len(pairs))
    print("== %s: %d pairs, source %s, target %s, classes %s, dis_thre_unit %.2f -> min %.2f" % (name, len(pairs), [len(c) for c in pairs[0].src], [len(c) for c in pairs[0].tgt],
          bytes(P.used_feature_type).decode()[:6], P.dis_thre_unit, P.dis_thre_min))
    print("   it   live pts   left over by the plain certificate   certified by the look   searched      search kernels   mean step of the iteration (m)")
    prev = np.zeros(6)
    for k in range(1, iters + 1):
        Pk = copy.copy(P); Pk.max_iter_num = k; Pk.converge_translation = 0.0; Pk.converge_rotation_d = 0.0
        ctx.set_option(abi.OPT_DEBUG_STOP, 0); ctx.set_profiling(0)
        b.run(Pk, results=res)
        ctx.set_option(abi.OPT_DEBUG_STOP, 20); ctx.set_profiling(1)
        b.run(Pk, results=res)
        pf = ctx.profile()
        cur = np.array([pf.nn_src_pts, pf.icp_search_ms[0], pf.icp_search_ms[2], pf.icp_search_ms[3], pf.ms_nn, 0.0])
        d = cur - prev; prev = cur
        live = max(d[0], 1.0)
        print("   %2d  %9d   %9d (%5.1f %%)                 %9d (%5.1f %%)      %9d (%5.1f %%)   %7.3f ms" % (k - 1, d[0], d[1], 100 * d[1] / live, d[2], 100 * d[2] / live, d[3], 100 * d[3] / live, d[4]))
    ctx.close()


scenes = bench.build_scenes(64, False, 16)
series("synthetic bench workload (configs[1])", [bench.global_pair(scenes, g) for g in range(nb)], bench.bench_params(), 12)
ds, dP = bench.demo_scenes()
series("the reference's demo scans (configs[0]), the reference's own class clouds", [bench.global_pair(ds, g) for g in range(nb)], dP, 10)
# the demo clouds with the bench's thresholds (is it the data or the parameters?)
Pb = bench.bench_params(); Pb.used_feature_type = dP.used_feature_type
series("demo scans with the bench's thresholds (1.4 -> 0.5 m, rate 1.1)", [bench.global_pair(ds, g) for g in range(nb)], Pb, 12)
