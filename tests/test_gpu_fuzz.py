"""Randomised differential test: seeded random points in the 23-dimensional option space of mm_lls_icp, on small pairs with
random initial guesses, HIP (every search tier) against the oracle.  Same comparison as tests/test_gpu_icp.py::compare."""
import numpy as np
import pytest

from mulls_amd import abi, synth
from oracle import pyoracle
from test_gpu_icp import compare

pytestmark = pytest.mark.gpu


def random_params(rng):
    used = "".join(rng.choice(["0", "1"], p=[0.3, 0.7]) for _ in range(6))
    if used[:3].count("1") == 0:
        used = "1" + used[1:]
    kw = dict(
        max_iter_num=int(rng.integers(1, 26)),
        dis_thre_unit=float(rng.uniform(0.6, 3.0)),
        converge_translation=float(rng.choice([0.0, 0.0005, 0.002, 0.01])),
        converge_rotation_d=float(rng.choice([0.0, 0.001, 0.01, 0.05])),
        dis_thre_min=float(rng.uniform(0.2, 0.8)),
        dis_thre_update_rate=float(rng.uniform(1.02, 1.4)),
        used_feature_type=used,
        weight_strategy="".join(rng.choice(["0", "1"]) for _ in range(4)),
        z_xy_balanced_ratio=float(rng.uniform(0.5, 2.0)),
        pt2pt_residual_window=float(rng.uniform(0.02, 0.3)),
        pt2pl_residual_window=float(rng.uniform(0.02, 0.3)),
        pt2li_residual_window=float(rng.uniform(0.02, 0.3)),
        apply_intersection_filter=int(rng.random() < 0.7),
        apply_motion_undistortion=int(rng.random() < 0.15),
        normal_shooting_on=int(rng.random() < 0.15),
        normal_bearing=float(rng.uniform(10.0, 60.0)),
        keep_less_source_points=int(rng.random() < 0.15),
        faithful=int(rng.random() < 0.8),
        sigma_thre=float(rng.choice([0.05, 0.35, 0.5, 5.0])),
        min_neccessary_corr_ratio=float(rng.choice([0.0, 0.03, 0.3])),
        max_bearable_rotation_d=float(rng.choice([0.5, 10.0, 45.0])),
        rng_seed=int(rng.integers(0, 2**31)),
    )
    return abi.default_params(**kw)


@pytest.mark.parametrize("block", range(4))
def test_random_option_points_match_oracle(ctx, pairs_small, block):
    rng = np.random.default_rng(900 + block)
    codes = set()
    for k in range(12):
        base, T_gt = pairs_small[int(rng.integers(0, len(pairs_small)))]
        pert = synth.se3(*rng.normal(0, 0.25, 3), *np.deg2rad(rng.normal(0, 0.6, 3)))
        pair = abi.PairData(base.tgt, base.src, init_guess=pert @ T_gt, tgt_bound=base.tgt_bound)
        P = random_params(rng)
        ro = pyoracle.icp(pair, P, trace_cap=32)[0]
        rg = ctx.icp(pair, P, trace_cap=32)[0]
        compare(ro, rg)
        codes.add(ro.code)
    assert 1 in codes
