"""Resident loop vs lock-step path by batch size on the bench workload: usage gpu_modes.py [sizes ...]"""
import sys, warnings, time, os
sys.path.insert(0, "."); warnings.filterwarnings("ignore")
import bench
from mulls_amd import abi, lib
scenes = bench.build_scenes(64, False, 16)
P = bench.bench_params()
for nb in ([int(a) for a in sys.argv[1:]] or [1, 8, 32, 128, 512, 1024, 2048, 4096]):
    pairs = [bench.global_pair(scenes, g) for g in range(nb)]
    row = []
    for mode in (4, 3, 0):
        ctx = lib.Context(0); ctx.set_nn_mode(mode)
        b = ctx.batch(pairs); res = abi.make_result_array(nb)
        b.run(P, results=res)
        reps = 3 if nb >= 512 else 10
        t = time.time()
        for _ in range(reps):
            b.run(P, results=res)
        dt = (time.time() - t) / reps
        row.append("mode %d: %8.3f ms %8.0f reg/s" % (mode, dt * 1e3, nb / dt))
        b.close(); ctx.close()
    print("%5d pairs  " % nb + "   ".join(row), flush=True)
