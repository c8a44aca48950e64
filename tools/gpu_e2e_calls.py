"""mulls_icp_batch call after call from host buffers (1024 pairs of the bench workload): wall time of every call and the library's own staging time,
to see the slow calls bench.py's value_end_to_end reports beside its median.  usage: gpu_e2e_calls.py [calls] [pairs]"""
import sys, time, warnings
sys.path.insert(0, "."); warnings.filterwarnings("ignore")
import numpy as np
import bench
from mulls_amd import abi, lib

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
scenes = bench.build_scenes(16, False, 8)
pairs = [bench.global_pair(scenes, g) for g in range(n)]
P = bench.bench_params()
ctx = lib.Context(0)
ctx.set_option(abi.OPT_LEAN_STAGING, 1)
m = (abi.make_pair_array(pairs), abi.make_result_array(n))
rows = []
for k in range(calls):
    t = time.perf_counter(); ctx.icp_batch(pairs, P, marshalled=m); w = time.perf_counter() - t
    pf = ctx.profile()
    rows.append((w * 1e3, ctx.last_call_s * 1e3, pf.ms_stage, pf.ms_stage_pack, pf.ms_host_launch, pf.ms_host_wait))
for k, r in enumerate(rows):
    print("call %2d: wall %7.2f ms  in library %7.2f  staging %6.2f (pack %5.2f)  host launch %6.2f wait %6.2f%s" % ((k,) + r + ("   <--" if r[1] > 1.5 * np.median([x[1] for x in rows]) else "",)))
