"""Seeded synthetic workloads of the BASELINE.json configurations other than the headline one (bench.py builds configs[1] / [3] itself):

  configs[2]  scan-to-local-map against a ~1 M-point submap (test/mulls_slam.cpp:679-685 with --local_map_max_pt_num lifted)
  configs[4]  synthetic 128-beam ~240 k-point scans, regime D (every return assigned to a class), all six classes, 40 iterations
  s2m20k      the reference's own scan-to-map default: a 20 000-point local map over five classes (src/map_manager.cpp:68-84,
              script/config/lo_gflag_list_kitti_urban.txt:64) with a 10-12 k ground class — the case that straddles the LDS tier's limit

Data source only (tests, tools/, bench.py); not part of the compute path.
"""
import numpy as np

from . import abi, synth

SEED0 = 20260924


def _guesses(pair, T_gt, n, seed, noise=(0.3, 0.5)):
    """n pairs sharing one scene's clouds, each with its own initial guess around the ground truth (the batch's first pair keeps the scene's)."""
    rng = np.random.default_rng(seed)
    dm, dd = noise
    out = [pair]
    for _ in range(n - 1):
        pert = synth.se3(*(rng.normal(0, dm / np.sqrt(3), 3)), *(np.deg2rad(rng.normal(0, dd / np.sqrt(3), 3))))
        out.append(abi.PairData(pair.tgt, pair.src, init_guess=pert @ T_gt, tgt_bound=pair.tgt_bound))
    return out


def dense_pair(seed=301):
    """configs[4]: one 128-beam pair, ~236 k returns each, nothing down-sampled (tests/test_gpu_large.py's fixture)."""
    none = {c: None for c in range(abi.NCLASS)}
    return synth.make_pair(seed, n_beams=128, n_az=1875, elev_deg=(-25.0, 15.0), src_counts=none, tgt_counts=none, vertex_count=2000)


def dense_params(converge=False):
    """All six classes, 40 iterations (convergence thresholds at 0 so that every registration executes them)."""
    ct, cr = (0.0005, 0.001) if converge else (0.0, 0.0)
    return abi.default_params(used_feature_type="111111", weight_strategy="1111", max_iter_num=40, dis_thre_unit=1.4, dis_thre_min=0.5,
                              converge_translation=ct, converge_rotation_d=cr, normal_bearing=20.0, sigma_thre=0.35)


def dense_batch(n_pairs, n_scenes=None, seed=301):
    """n_pairs dense pairs over n_scenes distinct scenes (default min(n_pairs, 4): a scene takes ~2 s to ray-cast)."""
    n_scenes = max(1, min(n_pairs, 4) if n_scenes is None else n_scenes)
    scenes = [dense_pair(seed + k) for k in range(n_scenes)]
    per = [(n_pairs + n_scenes - 1 - k) // n_scenes for k in range(n_scenes)]
    pairs = []
    for k, (p, T) in enumerate(scenes):
        pairs += _guesses(p, T, per[k], seed * 7 + k) if per[k] else []
    return pairs[:n_pairs]


def submap_pair(seed=7, n_az=7500):
    """configs[2]: a down-sampled 64-beam-sized source (800 / 400 / 1200 / 300 / 200) against a ~1 M-point map (one very dense revolution: 128 x n_az
    rays), the sizes of tools/gpu_configs.py and profiles/r0x_other_configs.txt."""
    none = {c: None for c in range(5)}
    return synth.make_pair(seed, n_beams=128, n_az=n_az, elev_deg=(-25.0, 15.0), src_counts={0: 800, 1: 400, 2: 1200, 3: 300, 4: 200}, tgt_counts=none,
                           vertex_count=2000)


def submap_params(converge=False):
    ct, cr = (0.0005, 0.001) if converge else (0.0, 0.0)
    return abi.kitti_params(converge_translation=ct, converge_rotation_d=cr, used_feature_type="111110")


def submap_batch(n_pairs, seed=7, n_az=7500):
    """n_pairs scans against ONE map: what a localisation server does.  Every pair carries the map's clouds (the batch stages them per pair)."""
    p, T = submap_pair(seed, n_az)
    return _guesses(p, T, n_pairs, seed * 13)


S2M_TGT = {abi.GROUND: 11500, abi.PILLAR: 1500, abi.FACADE: 5500, abi.BEAM: 900, abi.ROOF: 600}  # = 20 000 points over five classes


def s2m20k_batch(n_pairs, ground=11500, n_scenes=8, seed=911):
    """Scan-to-map against the reference's default local map: 20 000 points, `ground` of them in the ground class (capped variant: ground = 9728,
    the difference added to the facade class so that the map keeps its 20 000 points)."""
    tgt = dict(S2M_TGT)
    tgt[abi.FACADE] += tgt[abi.GROUND] - ground
    tgt[abi.GROUND] = ground
    n_scenes = max(1, min(n_scenes, n_pairs))
    pairs = []
    for k in range(n_scenes):
        p, T = synth.make_pair(seed + k, src_counts=synth.R_SOURCE, tgt_counts=tgt, vertex_count=0)
        per = (n_pairs + n_scenes - 1 - k) // n_scenes
        pairs += _guesses(p, T, per, seed * 5 + k) if per else []
    return pairs[:n_pairs]


def s2m_params(converge=False):
    ct, cr = (0.0005, 0.001) if converge else (0.0, 0.0)
    return abi.kitti_params(converge_translation=ct, converge_rotation_d=cr)


def algorithmic_bytes(results, used="111111", iters_run=None):
    """SURVEY 8(d)'s B_reg of a list of results (abi.Result), from the sizes the run reports: per iteration and used class
    64 Ns + 16 Nt + 8 Ns + 72 K, once 32 (Ns + Nt) + 64 Nt, plus 72 K for the residual pass.  Ns / Nt = post-filter class sizes, K = the
    final correspondence count (an upper estimate of the early iterations' K is not attempted: K <= Ns)."""
    total = 0
    for r in results:
        it = r.iters if iters_run is None else iters_run
        for c in range(abi.NCLASS):
            ns, nt, k = int(r.nsrc0[c]), int(r.ntgt0[c]), int(r.ncorr[c])
            if used[c] != "1" or ns == 0 or nt == 0:
                continue
            total += it * (72 * ns + 16 * nt + 72 * k) + 32 * (ns + nt) + 64 * nt + 72 * k
    return total
