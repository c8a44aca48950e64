"""The bench batch through the lock-step path a few times (for profiling): usage gpu_icp_phases_lock.py [pairs] [reps]"""
import sys, warnings
sys.path.insert(0, "."); warnings.filterwarnings("ignore")
import bench
from mulls_amd import abi, lib
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
scenes = bench.build_scenes(64, False, 16)
pairs = [bench.global_pair(scenes, g) for g in range(nb)]
P = bench.bench_params()
ctx = lib.Context(0); ctx.set_nn_mode(3)
b = ctx.batch(pairs)
res = abi.make_result_array(nb)
for _ in range(reps):
    b.run(P, results=res)
print("iters", sorted(set(r.iters for r in res)), "codes", sorted(set(r.code for r in res)))
