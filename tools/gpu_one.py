"""Run one resident batch a few times (for profiling). usage: gpu_one.py <nn_mode> <pairs> <reps>"""
import sys, warnings
sys.path.insert(0, "."); warnings.filterwarnings("ignore")
from mulls_amd import abi, synth, lib
mode, nb, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ctx = lib.Context(0); ctx.set_nn_mode(mode)
base = [synth.make_pair(s + 1)[0] for s in range(4)]
batch = ctx.batch([base[i % 4] for i in range(nb)])
P = abi.kitti_params(converge_translation=0.0, converge_rotation_d=0.0)
for _ in range(reps):
    batch.run(P)
