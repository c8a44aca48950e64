import sys, warnings, time
sys.path.insert(0, "."); warnings.filterwarnings("ignore")
from mulls_amd import abi, synth, lib
which, mode = int(sys.argv[1]), int(sys.argv[2])
src = {abi.GROUND: 600, abi.PILLAR: 300, abi.FACADE: 700, abi.BEAM: 150, abi.ROOF: 80}
tgt = {abi.GROUND: 2500, abi.PILLAR: 900, abi.FACADE: 3000, abi.BEAM: 400, abi.ROOF: 300}
base = [synth.make_pair(seed, n_beams=32, n_az=900, src_counts=src, tgt_counts=tgt, vertex_count=200)[0] for seed in (11, 12, 13)]
P = [abi.kitti_params(dis_thre_unit=2.4), abi.default_params(), abi.kitti_params(dis_thre_unit=2.4, faithful=0, weight_strategy="1011")][which]
ctx = lib.Context(0); ctx.set_nn_mode(mode)
print("start", which, mode, flush=True)
t = time.time()
r = ctx.icp_batch(base * 4, P, trace_cap=24)
print("done %.3f s" % (time.time() - t), [(x.code, x.iters, list(x.ncorr), x.T[12]) for x in r[:3]], flush=True)
