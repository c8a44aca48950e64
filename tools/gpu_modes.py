"""Device-resident loop (mode 4) vs lock-step path (mode 3) vs auto (mode 0) by batch size on the bench workload: median and minimum of the
resident run time over `reps` runs each.  usage: gpu_modes.py [sizes ...]   (MULLS_* option presets apply, see include/mulls_hip.h)"""
import sys, warnings, time, os
sys.path.insert(0, "."); warnings.filterwarnings("ignore")
import numpy as np
import bench
from mulls_amd import abi, lib
scenes = bench.build_scenes(64, False, 16)
P = bench.bench_params()
for nb in ([int(a) for a in sys.argv[1:]] or [1, 8, 32, 128, 512, 1024, 2048, 4096]):
    pairs = [bench.global_pair(scenes, g) for g in range(nb)]
    row = []
    ctxs = []
    for mode in (4, 3, 0):
        ctx = lib.Context(0); ctx.set_nn_mode(mode)
        b = ctx.batch(pairs); res = abi.make_result_array(nb)
        b.run(P, results=res); b.run(P, results=res)
        ctxs.append((mode, ctx, b, res, []))
    reps = 8 if nb >= 512 else 30
    for _ in range(reps):  # interleaved, so that a slow moment of the box hits every mode alike
        for mode, ctx, b, res, ts in ctxs:
            t = time.perf_counter(); b.run(P, results=res); ts.append(time.perf_counter() - t)
    for mode, ctx, b, res, ts in ctxs:
        md, mn = float(np.median(ts)), min(ts)
        row.append("mode %d: %7.3f ms (min %7.3f) %7.0f reg/s" % (mode, md * 1e3, mn * 1e3, nb / md))
        b.close(); ctx.close()
    print("%5d pairs  " % nb + "   ".join(row), flush=True)
