import sys, time, warnings
sys.path.insert(0, "."); warnings.filterwarnings("ignore")
import numpy as np
from mulls_amd import abi, synth, lib
from oracle import pyoracle
ctx = lib.Context(0)
pair, T = synth.make_pair(1)
for name, P in (("kitti s2s (early convergence)", abi.kitti_params(dis_thre_unit=2.4)), ("kitti fixed 20 iters", abi.kitti_params(converge_translation=0.0, converge_rotation_d=0.0))):
    ctx.icp(pair, P)
    ts = []
    for _ in range(20):
        t = time.perf_counter(); r = ctx.icp(pair, P)[0]; ts.append(time.perf_counter() - t)
    b = ctx.batch([pair]); b.run(P)
    tr = []
    for _ in range(20):
        t = time.perf_counter(); b.run(P); tr.append(time.perf_counter() - t)
    to = []
    for _ in range(5):
        t = time.perf_counter(); ro = pyoracle.icp(pair, P)[0]; to.append(time.perf_counter() - t)
    print("%s: iters %d | mulls_icp (alloc+upload+run+free) median %.2f ms | resident run %.2f ms | oracle (3 omp sections) %.2f ms" % (
        name, r.iters, 1e3 * np.median(ts), 1e3 * np.median(tr), 1e3 * np.median(to)))
