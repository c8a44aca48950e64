"""Counters of the global-memory tier's leftover search (MULLS_OPT_DEBUG_STOP = 21): queries over a run and the workgroup time they take.
usage: gpu_big_search_counts.py <cfg4|cfg2|s2m> [pairs]"""
import sys, warnings
sys.path.insert(0, "."); warnings.filterwarnings("ignore")
from mulls_amd import abi, lib, workloads as W
case = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1
pairs, P = {"cfg4": (lambda: (W.dense_batch(n), W.dense_params())), "cfg2": (lambda: (W.submap_batch(n), W.submap_params())),
            "s2m": (lambda: (W.s2m20k_batch(n), W.s2m_params()))}[case]()
ctx = lib.Context(0)
b = ctx.batch(pairs)
res = abi.make_result_array(n)
b.run(P, results=res)
ctx.set_option(abi.OPT_DEBUG_STOP, 21); ctx.set_profiling(1)
b.run(P, results=res)
pf = ctx.profile()
q, ms = pf.icp_fused_ms[0], pf.icp_fused_ms[1]
print("%s x %d: %d leftover queries over the run, %.2f ms of workgroup time (summed over workgroups) = %.3f us per query" % (case, n, q, ms, 1e3 * ms / max(q, 1)))
