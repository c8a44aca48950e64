"""PCIe-inclusive rate: mulls_icp_batch with the clouds in host memory (staging upload every call) vs the resident batch."""
import sys, time, warnings
sys.path.insert(0, "."); warnings.filterwarnings("ignore")
from mulls_amd import abi, synth, lib
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = lib.Context(0)
pairs = [synth.make_pair(s + 1)[0] for s in range(4)]
plist = [pairs[i % 4] for i in range(nb)]
P = abi.kitti_params(converge_translation=0.0, converge_rotation_d=0.0)
arr = abi.make_pair_array(plist); res = abi.make_result_array(nb)
import ctypes as C
ctx.lib.mulls_icp_batch(ctx.h, arr, nb, C.byref(P), res)
t = time.time()
for _ in range(3):
    ctx.lib.mulls_icp_batch(ctx.h, arr, nb, C.byref(P), res)
dt = (time.time() - t) / 3
mb = sum(len(c) for p in plist for c in p.tgt + p.src) * 48 / 1e6
print("host buffers, %d pairs: %.1f ms per call (%.0f reg/s), %.0f MB staged per call" % (nb, dt * 1e3, nb / dt, mb))
b = ctx.batch(plist); b.run(P)
t = time.time()
for _ in range(3):
    b.run(P)
dt2 = (time.time() - t) / 3
print("resident batch: %.1f ms per run (%.0f reg/s)" % (dt2 * 1e3, nb / dt2))
