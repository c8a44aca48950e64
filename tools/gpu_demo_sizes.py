"""The reference's demo pairs (BASELINE configs[0], tests/golden/demo_pair.npz) as resident batches of several sizes: median run time.
usage: gpu_demo_sizes.py [sizes ...]   (MULLS_HIP_LIB / MULLS_* presets apply)"""
import sys, warnings, time
sys.path.insert(0, "."); sys.path.insert(0, "tests"); warnings.filterwarnings("ignore")
import numpy as np
import bench
from mulls_amd import abi, lib
scenes, P = bench.demo_scenes()
ctx = lib.Context(0)
for nb in ([int(a) for a in sys.argv[1:]] or [1, 3, 12, 48, 128, 512]):
    pairs = [bench.global_pair(scenes, g) for g in range(nb)]
    b = ctx.batch(pairs); res = abi.make_result_array(nb)
    b.run(P, results=res); b.run(P, results=res)
    ts = []
    for _ in range(30 if nb < 512 else 10):
        t = time.perf_counter(); b.run(P, results=res); ts.append(time.perf_counter() - t)
    md = float(np.median(ts))
    print("%5d pairs  %8.3f ms (min %8.3f)  %8.0f reg/s   iterations %s" % (nb, md * 1e3, min(ts) * 1e3, nb / md, sorted(set(int(r.iters) for r in res))), flush=True)
    b.close()
