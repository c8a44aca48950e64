"""Device-resident loop (mode 4) against the lock-step path (mode 3) on the bench workload, bit for bit: usage gpu_icp_equal.py [pairs]"""
import sys, warnings
sys.path.insert(0, "."); warnings.filterwarnings("ignore")
import numpy as np
import bench
from mulls_amd import abi, lib
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
scenes = bench.build_scenes(64, False, 16)
pairs = [bench.global_pair(scenes, g) for g in range(nb)]
for name, P in (("bench", bench.bench_params()), ("all classes", abi.kitti_params(converge_translation=0.0, converge_rotation_d=0.0, used_feature_type="111110")),
                ("converging", abi.kitti_params(used_feature_type="111110", weight_strategy="1101"))):
    out = {}
    for mode in (4, 3):
        ctx = lib.Context(0); ctx.set_nn_mode(mode)
        b = ctx.batch(pairs); res = abi.make_result_array(nb)
        b.run(P, results=res)
        out[mode] = [(r.code, r.iters, tuple(r.T[:]), tuple(r.info[:]), r.sigma, tuple(r.ncorr)) for r in res]
        b.close(); ctx.close()
    same = sum(a == b for a, b in zip(out[4], out[3]))
    print("%-12s %d pairs: %d identical results, codes %s" % (name, nb, same, sorted(set(o[0] for o in out[4]))))
    assert same == nb
