"""The bench batch run WITHOUT profiling events as ONE sub-batch (MULLS_SPLIT_MAX_PAIRS=0), for a kernel trace of the plain launch sequence: usage gpu_gaps_noprof.py [pairs] [runs]"""
import sys, warnings
sys.path.insert(0, "."); warnings.filterwarnings("ignore")
import bench
from mulls_amd import abi, lib
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
scenes = bench.build_scenes(64, False, 16)
pairs = [bench.global_pair(scenes, g) for g in range(nb)]
P = bench.bench_params()
ctx = lib.Context(0)
b = ctx.batch(pairs)
res = abi.make_result_array(nb)
for _ in range(runs):
    b.run(P, results=res)
print("done", sorted(set(r.iters for r in res)))
