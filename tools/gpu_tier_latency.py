import sys, time, warnings
sys.path.insert(0, "."); warnings.filterwarnings("ignore")
import numpy as np
from mulls_amd import abi, synth, lib
pair, T = synth.make_pair(1)
P = abi.kitti_params(converge_translation=0.0, converge_rotation_d=0.0)
for mode in (0, 3, 2):
    ctx = lib.Context(0); ctx.set_nn_mode(mode)
    for nb in (1, 4, 8, 16):
        b = ctx.batch([pair]*nb); b.run(P)
        ts=[]
        for _ in range(15):
            t=time.perf_counter(); b.run(P); ts.append(time.perf_counter()-t)
        print("mode", mode, "pairs", nb, "resident run median %.3f ms" % (1e3*np.median(ts)), flush=True)
        b.close()
    ctx.close()
