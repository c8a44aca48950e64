import os
import sys
import warnings

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

warnings.filterwarnings("ignore", category=RuntimeWarning)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A box without the KFD device node has no AMD GPU at all: a plain `pytest tests` skips the gpu tests there instead of
    erroring in the context fixtures.  Where the node exists nothing is skipped — a library that cannot use the device fails
    the tests loudly (there is no fallback to hide behind)."""
    if os.path.exists("/dev/kfd"):
        return
    skip = pytest.mark.skip(reason="no /dev/kfd: not a GPU box (run with -m gpu on an MI355X)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", params=[4, 3, 2, 1], ids=["resident", "grid_lds", "grid_global", "brute"])
def ctx(request):
    """HIP context on device 0, once per tier (device-resident loop / lock-step uniform grids / LDS-tiled brute force).
    No fallback: if the library or the device is missing the gpu tests fail."""
    from mulls_amd import lib

    c = lib.Context(0)
    c.set_nn_mode(request.param)
    yield c
    c.close()


@pytest.fixture(scope="session")
def ctx_auto():
    """HIP context with the default tier selection (LDS grid for small targets, global-memory grid for large ones)."""
    from mulls_amd import lib

    c = lib.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def pairs_small():
    """A few reduced-size synthetic scan pairs (fast enough for the CPU suite)."""
    from mulls_amd import abi, synth

    out = []
    for seed in (11, 12, 13):
        src = {abi.GROUND: 600, abi.PILLAR: 300, abi.FACADE: 700, abi.BEAM: 150, abi.ROOF: 80}
        tgt = {abi.GROUND: 2500, abi.PILLAR: 900, abi.FACADE: 3000, abi.BEAM: 400, abi.ROOF: 300}
        out.append(synth.make_pair(seed, n_beams=32, n_az=900, src_counts=src, tgt_counts=tgt, vertex_count=200))
    return out


def planes_scene(rng, n_per=400, with_poles=True, noise=0.0):
    """Three mutually orthogonal planes + vertical poles, as six class clouds."""
    from mulls_amd import abi

    def plane(n, axis, off, normal):
        p = rng.uniform(-10, 10, (n, 3))
        p[:, axis] = off + rng.normal(0, noise, n) if noise else off
        return abi.make_points(p, np.tile(normal, (n, 1)), rng.uniform(0, 255, n))

    ground = plane(n_per, 2, -1.7, [0, 0, 1])
    f1 = plane(n_per, 1, 9.0, [0, -1, 0])
    f2 = plane(n_per, 0, 12.0, [-1, 0, 0])
    facade = np.concatenate([f1, f2])
    clouds = [ground, None, facade, None, None, None]
    if with_poles:
        k = n_per // 2
        base = rng.uniform(-8, 8, (12, 2))
        idx = rng.integers(0, 12, k)
        p = np.column_stack([base[idx, 0], base[idx, 1], rng.uniform(-1.7, 3.0, k)])
        clouds[abi.PILLAR] = abi.make_points(p, np.tile([0, 0, 1], (k, 1)), rng.uniform(0, 255, k))
    return clouds


def transformed_copy(clouds, T):
    """Apply a rigid transform (float64 math, float32 store) to every cloud: returns new clouds."""
    from mulls_amd import abi

    out = []
    R, t = T[:3, :3], T[:3, 3]
    for c in clouds:
        if c is None:
            out.append(None)
            continue
        xyz = np.column_stack([c["x"], c["y"], c["z"]]).astype(np.float64) @ R.T + t
        nrm = np.column_stack([c["nx"], c["ny"], c["nz"]]).astype(np.float64) @ R.T
        out.append(abi.make_points(xyz, nrm, c["intensity"], c["curvature"]))
    return out
