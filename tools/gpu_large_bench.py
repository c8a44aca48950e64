"""The large BASELINE configurations as resident batches: ms per run, registrations/s, the library's kernel split (hipEvents), SURVEY 8(d)'s algorithmic
bytes against the 8 TB/s HBM peak.  usage: gpu_large_bench.py <case> [pairs] [reps] [--check]
  cases: cfg4 (128-beam dense pair, six classes, 40 iterations)   cfg2 (scan vs ~1 M-point map, 20 iterations)
         s2m (20 000-point local map, 11.5 k ground class)        s2mcap (the same map with the ground class capped at 9728)
  --check: the first pair against the oracle (integer outputs equal, dT)"""
import sys, time, warnings
sys.path.insert(0, "."); warnings.filterwarnings("ignore")
import numpy as np
from mulls_amd import abi, synth, lib, workloads as W

args = [a for a in sys.argv[1:] if not a.startswith("--")]
case = args[0] if args else "cfg4"
n = int(args[1]) if len(args) > 1 else 1
reps = int(args[2]) if len(args) > 2 else 5
t0 = time.time()
if case == "cfg4":
    pairs, P = W.dense_batch(n), W.dense_params()
elif case == "cfg2":
    pairs, P = W.submap_batch(n), W.submap_params()
elif case == "s2m":
    pairs, P = W.s2m20k_batch(n), W.s2m_params()
elif case == "s2mcap":
    pairs, P = W.s2m20k_batch(n, ground=9728), W.s2m_params()
else:
    raise SystemExit(__doc__)
used = bytes(P.used_feature_type).decode()[:6]
t_gen = time.time() - t0
ctx = lib.Context(0)
b = ctx.batch(pairs)
r = b.run(P); r = b.run(P)
ts = []
for _ in range(reps):
    t = time.time(); r = b.run(P); ts.append(time.time() - t)
ms = float(np.median(ts)) * 1e3
ctx.set_profiling(True); b.run(P); pf = ctx.profile(); ctx.set_profiling(False)
B = W.algorithmic_bytes(r, used)
print("%s x %d pairs: src %s tgt %s used %s | gen %.1f s" % (case, n, [len(c) for c in pairs[0].src], [len(c) for c in pairs[0].tgt], used, t_gen))
print("  %.3f ms per run (min %.3f) = %.1f registrations/s | iters %s codes %s" % (ms, min(ts) * 1e3, n / ms * 1e3, sorted(set(x.iters for x in r)), sorted(set(x.code for x in r))))
print("  kernel ms (profiled run): setup %.3f search %.3f filter %.3f accumulate+step %.3f residual %.3f | launch sets %d" % (pf.ms_setup, pf.ms_nn, pf.ms_filter, pf.ms_accum, pf.ms_residual, pf.launches_nn))
print("  algorithmic bytes (SURVEY 8d) %.2f MB per run -> %.3f TB/s = %.4f of the 8 TB/s HBM peak" % (B / 1e6, B / ms / 1e9, B / ms / 1e9 / 8.0))
if "--check" in sys.argv:
    from oracle import pyoracle
    t = time.time(); ro = pyoracle.icp(pairs[0], P)[0]; t_or = time.time() - t
    dt, dr = synth.pose_error(r[0].T_matrix(), ro.T_matrix())
    same = (ro.code, ro.iters, list(ro.ncorr), list(ro.nsrc0), list(ro.ntgt0)) == (r[0].code, r[0].iters, list(r[0].ncorr), list(r[0].nsrc0), list(r[0].ntgt0))
    print("  oracle %.1f ms | integer outputs equal: %s | dT %.1e m %.1e rad" % (t_or * 1e3, same, dt, dr))
