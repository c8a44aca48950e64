import sys, os, warnings
sys.path.insert(0, "."); warnings.filterwarnings("ignore")
from mulls_amd import abi, synth, lib
mode, nb = int(sys.argv[1]), int(sys.argv[2])
ctx = lib.Context(0); ctx.set_nn_mode(mode)
base = [synth.make_pair(s + 1)[0] for s in range(4)]
batch = ctx.batch([base[i % 4] for i in range(nb)])
P = abi.kitti_params(converge_translation=0.0, converge_rotation_d=0.0, min_neccessary_corr_ratio=-1.0,
                     max_iter_num=int(sys.argv[3]) if len(sys.argv) > 3 else 20)
batch.run(P)
ctx.set_profiling(True); batch.run(P); pf = ctx.profile()
print("MULLS_DEBUG_STOP=%s mode %d: nn %.3f ms over %d launches -> %.1f us/launch" % (os.environ.get("MULLS_DEBUG_STOP"), mode, pf.ms_nn, pf.launches_nn, 1e3 * pf.ms_nn / max(pf.launches_nn, 1)))
