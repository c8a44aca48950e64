"""Non-finite input coordinates are outside the contract, but they must not take the device down:
a few NaN / inf values sprinkled into otherwise normal clouds, every tier, single and batched."""
import sys, warnings
sys.path.insert(0, "."); warnings.filterwarnings("ignore")
import numpy as np
from mulls_amd import abi, synth, lib
src = {abi.GROUND: 600, abi.PILLAR: 300, abi.FACADE: 700, abi.BEAM: 150, abi.ROOF: 80}
tgt = {abi.GROUND: 2500, abi.PILLAR: 900, abi.FACADE: 3000, abi.BEAM: 400, abi.ROOF: 300}
base, _ = synth.make_pair(11, n_beams=32, n_az=900, src_counts=src, tgt_counts=tgt, vertex_count=200)
rng = np.random.default_rng(5)
def poison(clouds, vals, fields):
    out = []
    for c in clouds:
        c = c.copy()
        if len(c) > 10:
            for v in vals:
                for f in fields:
                    c[f][rng.integers(0, len(c), 3)] = v
        out.append(c)
    return out
cases = {
    "nan_src_xyz": abi.PairData(base.tgt, poison(base.src, [np.nan], "xyz")),
    "nan_tgt_xyz": abi.PairData(poison(base.tgt, [np.nan], "xyz"), base.src),
    "inf_src_xyz": abi.PairData(base.tgt, poison(base.src, [np.inf, -np.inf], "xyz")),
    "inf_tgt_xyz": abi.PairData(poison(base.tgt, [np.inf, -np.inf], "xyz"), base.src, tgt_bound=base.tgt_bound),
    "huge_tgt": abi.PairData(poison(base.tgt, [3e38, -3e38], "xyz"), base.src, tgt_bound=base.tgt_bound),
    "nan_normals": abi.PairData(poison(base.tgt, [np.nan], ["nx", "ny", "nz"]), poison(base.src, [np.nan], ["nx", "ny", "nz"])),
}
for mode in (3, 2, 1):
    ctx = lib.Context(0); ctx.set_nn_mode(mode)
    for name, pair in cases.items():
        for P in (abi.kitti_params(), abi.default_params(used_feature_type="111111", apply_intersection_filter=0), abi.default_params(normal_shooting_on=1)):
            r = ctx.icp(pair, P)[0]
            rb = ctx.icp_batch([pair] * 12, P)
            print("mode %d %-12s -> code %d iters %d | batch codes %s" % (mode, name, r.code, r.iters, sorted(set(x.code for x in rb))), flush=True)
    ctx.close()
print("SURVIVED")
