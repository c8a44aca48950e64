"""CFilter::apply_motion_compensation / batch_apply_motion_compensation (include/common/cfilter.hpp:470-549) — what test/mulls_slam.cpp:703-712 applies to a
frame's clouds after its registration (on in script/config/lo_gflag_list_32.txt and _128.txt).

CPU: the oracle's restatement against the reference's own lines (oracle/_ref), field for field.  GPU (-m gpu): mulls_motion_compensate on host clouds and on a
device-resident feature block against the oracle.  The device evaluates acos / sin with the GPU's math library where the oracle calls glibc: a last-bit
difference of a double there moves a float coordinate with probability ~1e-9 per point, so the comparison allows one float ulp on at most 1e-4 of the points
(observed: none)."""
import numpy as np
import pytest

from mulls_amd import abi, synth
from oracle import pyoracle, pyref

FIELDS = ("x", "y", "z", "nx", "ny", "nz", "intensity", "curvature")


def scan_points(seed, n_beams=32, n_az=700):
    scene = synth.Scene(seed)
    scan = synth.raycast(scene, synth.se3(0, 0, scene.sensor_height), n_beams, n_az, seed=seed)
    pts = abi.make_points(scan["xyz"], scan["nrm"], scan["intensity"], scan["t"])
    rng = np.random.default_rng(seed)
    k = rng.integers(0, len(pts), 40)
    pts["curvature"][k[:10]] = -0.25  # time stamps outside [0, 1] and exactly on its ends: skipped / kept as the comparison says
    pts["curvature"][k[10:20]] = 1.5
    pts["curvature"][k[20:30]] = 0.0
    pts["curvature"][k[30:]] = 1.0
    return pts


TRANS = [synth.se3(0.9, 0.05, -0.02, 0.002, -0.003, 0.02), synth.se3(-1.4, 0.3, 0.1, np.deg2rad(2.0), np.deg2rad(-1.0), np.deg2rad(170.0)), np.eye(4),
         synth.se3(0.0, 0.0, 0.0, 0.0, 0.0, np.deg2rad(-179.9))]


def fields_equal(a, b):
    return all(np.array_equal(a[f], b[f], equal_nan=True) for f in FIELDS)


@pytest.mark.skipif(not pyref.available(), reason="oracle/_ref/libmulls_ref.so not built")
@pytest.mark.parametrize("thre", [0.0, 0.1, 0.5, 0.6])
def test_oracle_equals_reference_lines(thre):
    for k, T in enumerate(TRANS):
        pts = scan_points(30 + k)
        o, r = pyoracle.motion_compensate(pts, T, thre), pyref.motion_compensate(pts, T, thre)
        assert fields_equal(o, r)
        moved = (o["x"] != pts["x"]) | (o["y"] != pts["y"]) | (o["z"] != pts["z"])
        inside = (pts["curvature"] >= np.float32(thre)) & (pts["curvature"].astype(np.float64) <= 1.0 - np.float32(thre).astype(np.float64))
        assert not (moved & ~inside).any()  # nothing outside the time window moves
        assert all(np.array_equal(o[f], pts[f]) for f in FIELDS[3:])  # directions, intensity, time stamps stay


def test_full_and_zero_time_stamps():
    """t = 1 moves a point by Tran itself, t = 0 leaves it (the slerp's end points)."""
    pts = scan_points(41)
    T = TRANS[0]
    o = pyoracle.motion_compensate(pts, T)
    one, zero = pts["curvature"] == 1.0, pts["curvature"] == 0.0
    assert one.sum() >= 5 and zero.sum() >= 5
    full = pyoracle.transform(pts, T)
    assert np.allclose(o["x"][one], full["x"][one], atol=1e-5) and np.allclose(o["z"][one], full["z"][one], atol=1e-5)
    assert np.array_equal(o["x"][zero], pts["x"][zero]) and np.array_equal(o["y"][zero], pts["y"][zero])


def ulp_close(a, b, frac=1e-4):
    """equal up to one float ulp on at most `frac` of the points"""
    bad = 0
    for f in ("x", "y", "z"):
        ia, ib = a[f].view(np.int32).astype(np.int64), b[f].view(np.int32).astype(np.int64)
        d = np.abs(ia - ib)
        assert d.max(initial=0) <= 1, (f, int(d.max()))
        bad += int((d != 0).sum())
    assert bad <= max(1, int(frac * len(a))), bad
    assert all(np.array_equal(a[f], b[f]) for f in FIELDS[3:])
    return bad


@pytest.mark.gpu
@pytest.mark.parametrize("thre", [0.0, 0.2])
def test_device_equals_oracle_on_host_clouds(ctx_auto, thre):
    for k, T in enumerate(TRANS):
        pts = scan_points(50 + k, n_beams=64, n_az=1200)
        ulp_close(ctx_auto.motion_compensate(pts, T, thre), pyoracle.motion_compensate(pts, T, thre))


@pytest.mark.gpu
def test_resident_block_is_compensated_in_place(ctx_auto):
    """mulls_block_motion_compensate: the block's class clouds and their *_down clouds move as the oracle moves the host copies, the vertex and unground clouds
    stay (undistort_keypoints off, as every caller of the reference leaves it); with it on the vertex cloud moves twice (it rides in both batch calls)."""
    scene = synth.Scene(61)
    scan = synth.raycast(scene, synth.se3(0, 0, scene.sensor_height), 32, 900, seed=61)
    pts = abi.make_points(scan["xyz"], np.zeros_like(scan["xyz"]), scan["intensity"], scan["t"])
    X = abi.extract_params(ground=abi.ground_params(nonground_random_down_rate=1), classify=abi.classify_params(neighbor_k=20))
    T = TRANS[0]
    moving = [abi.EX_GROUND, abi.EX_GROUND_DOWN] + [abi.EX_PILLAR + k for k in range(8)]
    for keypoints in (False, True):
        b = ctx_auto.block().extract(pts, X)
        before = {k: abi.points_of(b.download(k)) for k in range(abi.EX_COUNT) if k not in (abi.EX_RAW, abi.EX_DOWN)}
        assert sum(len(before[k]) for k in moving) > 1000 and len(before[abi.EX_VERTEX]) > 0
        b.motion_compensate(T, undistort_keypoints=keypoints)
        for k, cloud in before.items():
            after = abi.points_of(b.download(k))
            want = cloud
            if k in moving:
                want = pyoracle.motion_compensate(cloud, T)
            elif k == abi.EX_VERTEX and keypoints:
                want = pyoracle.motion_compensate(pyoracle.motion_compensate(cloud, T), T)
            ulp_close(after, want)
        # a device-resident cloud through the general entry point: in place, no copy
        import ctypes as C

        c = b.cloud(abi.EX_GROUND)
        ctx_auto._check(ctx_auto.lib.mulls_motion_compensate(ctx_auto.h, c.pts, c.n, abi.POINT_BYTES, abi.colmajor16(np.linalg.inv(T)), C.c_float(0.0)), "mulls_motion_compensate")
        again = abi.points_of(b.download(abi.EX_GROUND))
        ulp_close(again, pyoracle.motion_compensate(pyoracle.motion_compensate(before[abi.EX_GROUND], T), np.linalg.inv(T)))
        b.close()


@pytest.mark.gpu
def test_callers_own_device_buffer_and_foreign_blocks(ctx_auto):
    """mulls_motion_compensate on a device allocation the library does not own (the caller's own buffer — here a torch tensor): compensated in place like a cloud
    of the library, the same bytes as the host path returns.  mulls_block_motion_compensate refuses a block of another context.  (Advisor findings of round 4.)"""
    import ctypes as C

    from mulls_amd import lib

    hip = C.CDLL("libamdhip64.so")  # the caller's own allocation: straight from the HIP runtime, nothing of the library's
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    T = TRANS[1]
    pts = scan_points(77, n_beams=32, n_az=700)
    want = ctx_auto.motion_compensate(pts, T, 0.1)
    raw = abi.records(pts).copy()
    dev = C.c_void_p()
    assert hip.hipMalloc(C.byref(dev), raw.nbytes) == 0
    assert hip.hipMemcpy(dev, C.c_void_p(raw.ctypes.data), raw.nbytes, 1) == 0  # hipMemcpyHostToDevice
    ctx_auto._check(ctx_auto.lib.mulls_motion_compensate(ctx_auto.h, dev, len(raw), abi.POINT_BYTES, abi.colmajor16(T), C.c_float(0.1)), "mulls_motion_compensate")
    back = np.empty_like(raw)
    assert hip.hipMemcpy(C.c_void_p(back.ctypes.data), dev, raw.nbytes, 2) == 0  # hipMemcpyDeviceToHost
    assert hip.hipFree(dev) == 0
    got = abi.points_of(back)
    assert all(np.array_equal(got[f], want[f]) for f in FIELDS)
    # a block belongs to the context that made it
    other = lib.Context(0)
    scene = synth.Scene(62)
    scan = synth.raycast(scene, synth.se3(0, 0, scene.sensor_height), 16, 400, seed=62)
    spts = abi.make_points(scan["xyz"], np.zeros_like(scan["xyz"]), scan["intensity"], scan["t"])
    X = abi.extract_params(ground=abi.ground_params(nonground_random_down_rate=1), classify=abi.classify_params(neighbor_k=20))
    b = ctx_auto.block().extract(spts, X)
    rc = other.lib.mulls_block_motion_compensate(other.h, b.h, abi.colmajor16(T), 0)
    assert rc == abi.MULLS_E_INVALID
    b.motion_compensate(T)  # its own context: fine
    b.close()
    other.close()
