import sys, time, warnings
sys.path.insert(0, "."); warnings.filterwarnings("ignore")
from mulls_amd import abi, synth, lib
ctx = lib.Context(0)
pairs = [synth.make_pair(s + 1)[0] for s in range(4)]
Pb = abi.kitti_params(converge_translation=0.0, converge_rotation_d=0.0)
for mode in (3, 2):
    ctx.set_nn_mode(mode)
    for nb in (1, 2, 4, 8):
        batch = ctx.batch([pairs[i % 4] for i in range(nb)])
        batch.run(Pb)
        t = time.time()
        for _ in range(10): batch.run(Pb)
        print("mode %d batch %d: %.3f ms/run" % (mode, nb, (time.time() - t) / 10 * 1e3))
        batch.close()
