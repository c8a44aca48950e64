"""Debug driver of the device-resident loop: usage gpu_icp_debug.py <mode> <pairs> <max_iter>"""
import sys, warnings, time
sys.path.insert(0, "."); warnings.filterwarnings("ignore")
from mulls_amd import abi, synth, lib
mode, nb, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
src = {abi.GROUND: 600, abi.PILLAR: 300, abi.FACADE: 700}
tgt = {abi.GROUND: 2500, abi.PILLAR: 900, abi.FACADE: 3000}
base = [synth.make_pair(11 + s, n_beams=32, n_az=900, src_counts=src, tgt_counts=tgt, vertex_count=0)[0] for s in range(3)]
ctx = lib.Context(0); ctx.set_nn_mode(mode)
P = abi.kitti_params(dis_thre_unit=2.4, max_iter_num=iters)
print("start", mode, nb, iters, flush=True)
t = time.time()
r = ctx.icp_batch([base[i % 3] for i in range(nb)], P, trace_cap=4)
print("done %.3f s" % (time.time() - t), [(x.code, x.iters, list(x.ncorr)[:3], x.T[12]) for x in r[:3]], flush=True)
