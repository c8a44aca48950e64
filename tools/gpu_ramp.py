"""Step time of the bench batch against the time since the process's first launch (does the device reach its steady clocks only after a while?):
usage gpu_ramp.py [pairs] [seconds] [idle seconds before a second burst]"""
import sys, time, warnings
sys.path.insert(0, "."); warnings.filterwarnings("ignore")
import bench
from mulls_amd import abi, lib
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
idle = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
scenes = bench.build_scenes(64, False, 16)
pairs = [bench.global_pair(scenes, g) for g in range(nb)]
P = bench.bench_params()
ctx = lib.Context(0)
b = ctx.batch(pairs)
res = abi.make_result_array(nb)


def burst(tag, secs):
    t0 = time.perf_counter()
    win_t, win_n, line = t0, 0, []
    while True:
        b.run(P, results=res)
        win_n += 1
        now = time.perf_counter()
        if now - win_t >= 2.0:
            line.append("%5.1fs %.2f" % (now - t0, (now - win_t) / win_n * 1e3))
            win_t, win_n = now, 0
        if now - t0 >= secs:
            break
    print(tag, "ms per step by 2-second window:", " | ".join(line))


burst("first burst ", secs)
if idle > 0:
    time.sleep(idle)
    burst("after %.0f s idle" % idle, min(secs, 12.0))
