"""ctypes binding of oracle/_ref/libmulls_ref.so — the reference's own MULLS-ICP function bodies compiled against the
stand-in headers of oracle/ref_shim (TEST INFRASTRUCTURE ONLY; built by oracle/build_ref.sh where /root/reference exists).
"""
import ctypes as C
import os

from mulls_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libmulls_ref.so")
_LIB = None


def available():
    return os.path.exists(LIB_PATH)


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(LIB_PATH)
        _LIB.mulls_ref_icp.argtypes = [C.POINTER(abi.Pair), C.POINTER(abi.Params), C.POINTER(abi.Result)]
    return _LIB


def icp(pair, params, trace_cap=0):
    res = abi.make_result_array(1, 0)
    p = pair.as_pair()
    rc = lib().mulls_ref_icp(C.byref(p), C.byref(params), C.byref(res[0]))
    if rc != 0:
        raise RuntimeError("reference returned %d" % rc)
    return res


def icp_3dof_ground(pair, params):
    res = abi.make_result_array(1, 0)
    p = pair.as_pair()
    rc = lib().mulls_ref_icp_3dof_ground(C.byref(p), C.byref(params), C.byref(res[0]))
    if rc != 0:
        raise RuntimeError("reference returned %d" % rc)
    return res


def icp_4dof_global(pair, heading_step_d, station, max_iter_num=20, dis_thre_unit=1.5, converge_translation=0.005, dis_thre_min=0.5,
                    dis_thre_update_rate=1.05):
    res = abi.make_result_array(1, 0)
    p = pair.as_pair()
    ok = C.c_int(0)
    st = (C.c_double * 3)(*station)
    rc = lib().mulls_ref_icp_4dof_global(C.byref(p), C.c_float(heading_step_d), st, int(max_iter_num), C.c_float(dis_thre_unit),
                                         C.c_float(converge_translation), C.c_float(dis_thre_min), C.c_float(dis_thre_update_rate),
                                         C.byref(res[0]), C.byref(ok))
    if rc != 0:
        raise RuntimeError("reference returned %d" % rc)
    return res, bool(ok.value)


def motion_compensate(pts, Tran, s_ambiguous_thre=0.0):
    """CFilter::apply_motion_compensation(pc_in_out, Tran, s_ambigous_thre), the reference's own lines (cfilter.hpp:470-491)."""
    from oracle import pyoracle

    return pyoracle.motion_compensate(pts, Tran, s_ambiguous_thre, fn=lib().mulls_ref_motion_compensate)


def map_update(map_clouds, map_pose, frame_down, frame_pose, params):
    """MapManager::update_local_map, the reference's own lines (src/map_manager.cpp:18-256)."""
    from oracle import pyoracle

    return pyoracle.map_update(map_clouds, map_pose, frame_down, frame_pose, params, fn=lib().mulls_ref_map_update)


def ground_filter(pts, params):
    """CFilter::fast_ground_filter, the reference's own lines (cfilter.hpp:1658-2036)."""
    from oracle import pyoracle

    return pyoracle._ground_filter(lib().mulls_ref_ground_filter, pts, params)


def classify_nground(pts, params):
    """CFilter::classify_nground_pts, the reference's own lines (cfilter.hpp:2058-2290, with pca.hpp:207-454) over oracle/pcl_restated.h."""
    from oracle import pyoracle

    return pyoracle._classify(lib().mulls_ref_classify_nground, pts, params)


def scanner_filter(pts, self_radius, ghost_radius, z_min_ghost, z_min_global):
    """CFilter::scanner_filter, the reference's own lines (cfilter.hpp:914-929)."""
    from oracle import pyoracle

    return pyoracle.scanner_filter(pts, self_radius, ghost_radius, z_min_ghost, z_min_global, fn=lib().mulls_ref_scanner_filter)


def dist_filter(pts, xy_dist_min, xy_dist_max):
    """CFilter::dist_filter(cloud, xy_dist_min, xy_dist_max), the reference's own lines (cfilter.hpp:806-832)."""
    from oracle import pyoracle

    return pyoracle.dist_filter(pts, xy_dist_min, xy_dist_max, fn=lib().mulls_ref_dist_filter)


def voxel_downsample(pts, voxel_size):
    """CFilter::voxel_downsample, the reference's own lines (cfilter.hpp:83-160)."""
    from oracle import pyoracle

    return pyoracle.voxel_downsample(pts, voxel_size, fn=lib().mulls_ref_voxel_downsample)


def extract_semantic_pts(scan, X):
    """dist_filter + the member CFilter::extract_semantic_pts on a cloudblock_t, the reference's own lines (test/mulls_slam.cpp:359-365,
    cfilter.hpp:2295-2413).  Returns (the clouds of enum mulls_extract_cloud as (n, 48) uint8 records, (gf_down_rate_ground,
    gf_downsample_rate_nonground) after the call)."""
    import ctypes as C

    import numpy as np

    from mulls_amd import abi

    raw_in = abi.records(scan)
    n = len(raw_in)
    outs = [np.zeros((max(n, 1), abi.POINT_BYTES), np.uint8) for _ in range(abi.EX_COUNT)]
    out_p = (C.c_void_p * abi.EX_COUNT)(*[o.ctypes.data for o in outs])
    cap = (C.c_uint32 * abi.EX_COUNT)(*([n] * abi.EX_COUNT))
    nout = (C.c_uint32 * abi.EX_COUNT)()
    rates = (C.c_int32 * 2)()
    f = lib().mulls_ref_extract_semantic_pts
    f.restype = C.c_int
    rc = f(raw_in.ctypes.data_as(C.c_void_p), C.c_uint32(n), C.c_uint32(abi.POINT_BYTES), C.byref(X), out_p, cap, nout, rates)
    if rc != 0:
        raise RuntimeError("mulls_ref_extract_semantic_pts returned %d" % rc)
    return [outs[k][: nout[k]].copy() for k in range(abi.EX_COUNT)], (rates[0], rates[1])
