"""ctypes binding of oracle/liboracle.so — the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; nothing under
mulls_amd/ does.  See the header of oracle/mulls_oracle.cpp for what the oracle restates and how it is pinned.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from mulls_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "mulls_oracle.cpp")
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.mulls_oracle_icp.argtypes = [C.POINTER(abi.Pair), C.POINTER(abi.Params), C.POINTER(abi.Result), C.c_int, C.c_int]
        _LIB.mulls_oracle_icp.restype = C.c_int
        _LIB.mulls_oracle_rotation_angle.restype = C.c_double
    return _LIB


def icp(pair, params, trace_cap=0, nn_mode=0, use_omp=1):
    """Run the oracle on one abi.PairData.  Returns abi.Result (index 0 of a 1-element array)."""
    res = abi.make_result_array(1, trace_cap)
    p = pair.as_pair()
    rc = lib().mulls_oracle_icp(C.byref(p), C.byref(params), C.byref(res[0]), nn_mode, use_omp)
    if rc != 0:
        raise RuntimeError("oracle returned %d" % rc)
    return res


def transform(pts, T):
    pts = np.ascontiguousarray(pts).copy()
    Tc = (C.c_double * 16)(*np.asarray(T, dtype=np.float64).T.reshape(-1))
    lib().mulls_oracle_transform(C.c_void_p(pts.ctypes.data), C.c_uint32(len(pts)), C.c_uint32(abi.POINT_BYTES), Tc)
    return pts


def motion_compensate(pts, Tran, s_ambiguous_thre=0.0, fn=None):
    """CFilter::apply_motion_compensation(pc_in_out, Tran, s_ambigous_thre) (cfilter.hpp:470-491) on a copy of pts."""
    raw = abi.records(pts).copy()  # raw bytes: a copy of a RECORD array would drop the bytes between its fields (data[3], the padding), which the reference keeps
    Tc = (C.c_double * 16)(*np.asarray(Tran, dtype=np.float64).T.reshape(-1))
    f = fn if fn is not None else lib().mulls_oracle_motion_compensate
    rc = f(C.c_void_p(raw.ctypes.data), C.c_uint32(len(raw)), C.c_uint32(abi.POINT_BYTES), Tc, C.c_float(s_ambiguous_thre))
    if rc != 0:
        raise RuntimeError("motion_compensate returned %d" % rc)
    return abi.points_of(raw)


def correspond(src, tgt, dis_thre, normal_check=True, angle_deg=45.0, nn_mode=0):
    n = len(src)
    match = np.zeros(n, np.int32)
    d2 = np.zeros(n, np.float32)
    flags = np.zeros(n, np.uint8)
    cs, ct = abi.as_cloud(src), abi.as_cloud(tgt)
    lib().mulls_oracle_correspond(C.byref(cs), C.byref(ct), C.c_float(dis_thre), int(normal_check), C.c_float(angle_deg),
                                  match.ctypes.data_as(C.c_void_p), d2.ctypes.data_as(C.c_void_p), flags.ctypes.data_as(C.c_void_p), nn_mode)
    return match, d2, flags


def accumulate(metric, src, tgt, corr_src, corr_tgt, corr_d2, iter_num, class_weight, dist_w, resid_w, inten_w, window):
    corr_src = np.ascontiguousarray(corr_src, np.int32)
    corr_tgt = np.ascontiguousarray(corr_tgt, np.int32)
    corr_d2 = np.ascontiguousarray(corr_d2, np.float32)
    out = np.zeros(27, np.float64)
    w = np.zeros(len(corr_src), np.float32)
    cs, ct = abi.as_cloud(src), abi.as_cloud(tgt)
    lib().mulls_oracle_accumulate(int(metric), C.byref(cs), C.byref(ct), corr_src.ctypes.data_as(C.c_void_p),
                                  corr_tgt.ctypes.data_as(C.c_void_p), corr_d2.ctypes.data_as(C.c_void_p), C.c_uint32(len(corr_src)),
                                  int(iter_num), C.c_float(class_weight), int(dist_w), int(resid_w), int(inten_w), C.c_float(window),
                                  out.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p))
    return out, w


def nn(src, tgt, nn_mode=0):
    n = len(src)
    match = np.zeros(n, np.int32)
    d2 = np.zeros(n, np.float32)
    cs, ct = abi.as_cloud(src), abi.as_cloud(tgt)
    lib().mulls_oracle_nn(C.byref(cs), C.byref(ct), match.ctypes.data_as(C.c_void_p), d2.ctypes.data_as(C.c_void_p), nn_mode)
    return match, d2


def construct_trans(x):
    xs = (C.c_double * 6)(*x)
    T = (C.c_double * 16)()
    lib().mulls_oracle_construct_trans(xs, T)
    return np.array(T[:]).reshape(4, 4).T.copy()


def solve(atpa, atpb):
    A = (C.c_double * 36)(*np.asarray(atpa, np.float64).T.reshape(-1))
    b = (C.c_double * 6)(*atpb)
    x = (C.c_double * 6)()
    cof = (C.c_double * 36)()
    rc = lib().mulls_oracle_solve(A, b, x, cof)
    return rc, np.array(x[:]), np.array(cof[:]).reshape(6, 6).T.copy()


def rotation_angle(T):
    Tc = (C.c_double * 16)(*np.asarray(T, dtype=np.float64).T.reshape(-1))
    return float(lib().mulls_oracle_rotation_angle(Tc))


def icp_3dof_ground(pair, params, trace_cap=0, nn_mode=0):
    """lls_icp_3dof_ground (cregistration.hpp:1443-1582); reads the like-named fields of `params`."""
    res = abi.make_result_array(1, trace_cap)
    p = pair.as_pair()
    rc = lib().mulls_oracle_icp_3dof_ground(C.byref(p), C.byref(params), C.byref(res[0]), nn_mode)
    if rc != 0:
        raise RuntimeError("oracle returned %d" % rc)
    return res


def icp_4dof_global(pair, heading_step_d, station, max_iter_num=20, dis_thre_unit=1.5, converge_translation=0.005, dis_thre_min=0.5,
                    dis_thre_update_rate=1.05, nn_mode=0, use_omp=1):
    """mm_lls_icp_4dof_global (cregistration.hpp:1584-1681).  Returns (results, success, best_heading_deg)."""
    res = abi.make_result_array(1, 0)
    p = pair.as_pair()
    ok, best = C.c_int(0), C.c_float(0)
    st = (C.c_double * 3)(*station)
    rc = lib().mulls_oracle_icp_4dof_global(C.byref(p), C.c_float(heading_step_d), st, int(max_iter_num), C.c_float(dis_thre_unit),
                                            C.c_float(converge_translation), C.c_float(dis_thre_min), C.c_float(dis_thre_update_rate),
                                            C.byref(res[0]), C.byref(ok), C.byref(best), nn_mode, use_omp)
    if rc != 0:
        raise RuntimeError("oracle returned %d" % rc)
    return res, bool(ok.value), float(best.value)


def map_update(map_clouds, map_pose, frame_down, frame_pose, params, fn=None):
    """MapManager::update_local_map on host clouds.  Returns (new map clouds[6], frame clouds as appended[6], abi.MapReport).
    `fn`: entry point with the signature of mulls_oracle_map_update (oracle/pyref.py passes the reference-lines build)."""
    mc = [abi.as_points(c) for c in map_clouds]
    fc = [abi.as_points(c) for c in frame_down]
    m_arr, f_arr = (abi.Cloud * 6)(), (abi.Cloud * 6)()
    out_m = [np.zeros(len(mc[c]) + len(fc[c]), abi.POINT_DTYPE) for c in range(6)]
    out_f = [np.zeros(len(fc[c]), abi.POINT_DTYPE) for c in range(6)]
    pm, pf = (C.c_void_p * 6)(), (C.c_void_p * 6)()
    for c in range(6):
        m_arr[c], f_arr[c] = abi.as_cloud(mc[c]), abi.as_cloud(fc[c])
        pm[c] = out_m[c].ctypes.data if len(out_m[c]) else None
        pf[c] = out_f[c].ctypes.data if len(out_f[c]) else None
    nm, nf = (C.c_uint32 * 6)(), (C.c_uint32 * 6)()
    rep = abi.MapReport()
    if fn is None:
        fn = lib().mulls_oracle_map_update
    fn.restype = C.c_int
    rc = fn(m_arr, abi.colmajor16(map_pose), f_arr, abi.colmajor16(frame_pose), C.byref(params), pm, nm, pf, nf, C.byref(rep))
    if rc != 0:
        raise RuntimeError("oracle map_update returned %d" % rc)
    raw = lambda a, n: a[:n].view(np.uint8).copy().view(abi.POINT_DTYPE)  # .copy() of a record array drops the bytes between fields (normal[3])
    return [raw(out_m[c], nm[c]) for c in range(6)], [raw(out_f[c], nf[c]) for c in range(6)], rep


def census(reset=False):
    """Operator census of the runs so far (see g_census in mulls_oracle.cpp): dict of counters."""
    out = (C.c_ulonglong * 8)()
    lib().mulls_oracle_census(out, int(reset))
    names = ("radius_tests", "radius_equal", "rejector_tests", "rejector_equal", "rejector_nan", "brute_queries", "brute_ties")
    d = {n: int(out[i]) for i, n in enumerate(names)}
    d["ransac_tests"], d["ransac_near_boundary"], d["ransac_forms_disagree"] = int(out[7]) & 0xffffffff, (int(out[7]) >> 32) & 0xffff, int(out[7]) >> 48
    return d


def _ground_filter(fn, pts, params):
    pts = abi.as_points(pts)
    n = len(pts)
    outs = [np.zeros(max(n, 1), abi.POINT_DTYPE) for _ in range(3)]
    raw = [np.zeros(max(n, 1) * abi.POINT_BYTES, np.uint8) for _ in range(3)]
    nout = (C.c_uint32 * 3)()
    rc = fn(pts.ctypes.data_as(C.c_void_p), C.c_uint32(n), C.c_uint32(abi.POINT_BYTES), C.byref(params), raw[0].ctypes.data_as(C.c_void_p), C.c_uint32(n),
            raw[1].ctypes.data_as(C.c_void_p), C.c_uint32(n), raw[2].ctypes.data_as(C.c_void_p), C.c_uint32(n), nout)
    if rc != 0:
        raise RuntimeError("ground filter returned %d" % rc)
    # raw 48-byte records (every byte: data[3] carries the height above ground)
    return [raw[k][: nout[k] * abi.POINT_BYTES].reshape(nout[k], abi.POINT_BYTES).copy() for k in range(3)]


def ground_filter(pts, params):
    """CFilter::fast_ground_filter, oracle restatement.  Returns (ground, ground_down, unground) as (n, 48) uint8 record arrays."""
    return _ground_filter(lib().mulls_oracle_ground_filter, pts, params)


def _classify(fn, pts, params):
    raw_in = abi.records(pts)
    n = len(raw_in)
    outs = [np.zeros((max(n, 1), abi.POINT_BYTES), np.uint8) for _ in range(abi.CL_COUNT)]
    after = np.zeros((max(n, 1), abi.POINT_BYTES), np.uint8)
    out_p = (C.c_void_p * abi.CL_COUNT)(*[o.ctypes.data for o in outs])
    cap = (C.c_uint32 * abi.CL_COUNT)(*([n] * abi.CL_COUNT))
    nout = (C.c_uint32 * abi.CL_COUNT)()
    n_after = C.c_uint32(0)
    fn.restype = C.c_int
    rc = fn(raw_in.ctypes.data_as(C.c_void_p), C.c_uint32(n), C.c_uint32(abi.POINT_BYTES), C.byref(params), out_p, cap, nout,
            after.ctypes.data_as(C.c_void_p), C.byref(n_after))
    if rc != 0:
        raise RuntimeError("classify_nground returned %d" % rc)
    return [outs[k][: nout[k]].copy() for k in range(abi.CL_COUNT)], after[: n_after.value].copy()


def classify_nground(pts, params):
    """CFilter::classify_nground_pts, oracle restatement.  Returns ([9 clouds in enum mulls_classify_cloud order], cloud_in afterwards), all as
    raw (n, 48) uint8 records (normal[3] and the other unnamed bytes matter here)."""
    return _classify(lib().mulls_oracle_classify_nground, pts, params)


def search_census(reset=False):
    """Where the restated PCL / FLANN searches had to choose (oracle/pcl_restated.h: search_census): dict of searches, max_nn cuts inside a group of equal
    distances, candidates exactly on the radius, searches returning two neighbours at the same distance."""
    out = (C.c_ulonglong * 4)()
    lib().mulls_oracle_search_census(out, int(reset))
    return dict(searches=int(out[0]), cut_in_tie=int(out[1]), on_radius=int(out[2]), equal_neighbours=int(out[3]))


def classify_census(reset=False):
    """classify_nground's decisions against the accuracy of upstream's float eigen-solver: points with a decision, and those among them where a compared
    quantity lies within 1e-5 / 1e-4 of its threshold."""
    out = (C.c_ulonglong * 3)()
    lib().mulls_oracle_classify_census(out, int(reset))
    return dict(decisions=int(out[0]), within_1e5=int(out[1]), within_1e4=int(out[2]))


def nms_ties(reset=False):
    """How many neighbours in non_max_suppress's visiting order had equal normal[3] so far (the order upstream leaves to std::sort)."""
    f = lib().mulls_oracle_nms_ties
    f.restype = C.c_ulonglong
    return int(f(int(reset)))


def scanner_filter(pts, self_radius, ghost_radius, z_min_ghost, z_min_global, fn=None):
    """CFilter::scanner_filter (cfilter.hpp:914-929).  Returns the kept points as (n, 48) uint8 records."""
    raw_in = abi.records(pts)
    n = len(raw_in)
    out = np.zeros((max(n, 1), abi.POINT_BYTES), np.uint8)
    n_out = C.c_uint32(0)
    f = fn if fn is not None else lib().mulls_oracle_scanner_filter
    f.restype = C.c_int
    rc = f(raw_in.ctypes.data_as(C.c_void_p), C.c_uint32(n), C.c_uint32(abi.POINT_BYTES), C.c_float(self_radius), C.c_float(ghost_radius), C.c_float(z_min_ghost),
           C.c_float(z_min_global), out.ctypes.data_as(C.c_void_p), C.c_uint32(n), C.byref(n_out))
    if rc != 0:
        raise RuntimeError("scanner_filter returned %d" % rc)
    return out[: n_out.value].copy()


def _filter_call(f, pts, *args):
    raw_in = abi.records(pts)
    n = len(raw_in)
    out = np.zeros((max(n, 1), abi.POINT_BYTES), np.uint8)
    n_out = C.c_uint32(0)
    f.restype = C.c_int
    rc = f(raw_in.ctypes.data_as(C.c_void_p), C.c_uint32(n), C.c_uint32(abi.POINT_BYTES), *args, out.ctypes.data_as(C.c_void_p), C.c_uint32(n), C.byref(n_out))
    if rc != 0:
        raise RuntimeError("%s returned %d" % (getattr(f, "__name__", "filter"), rc))
    return out[: n_out.value].copy()


def dist_filter(pts, xy_dist_min, xy_dist_max, fn=None):
    """CFilter::dist_filter(cloud, xy_dist_min, xy_dist_max) (cfilter.hpp:806-832).  Returns the kept points as (n, 48) uint8 records."""
    return _filter_call(fn if fn is not None else lib().mulls_oracle_dist_filter, pts, C.c_double(xy_dist_min), C.c_double(xy_dist_max))


def voxel_downsample(pts, voxel_size, fn=None):
    """CFilter::voxel_downsample (cfilter.hpp:83-160).  Returns pc_down as (n, 48) uint8 records."""
    return _filter_call(fn if fn is not None else lib().mulls_oracle_voxel_downsample, pts, C.c_float(voxel_size))


def extract_features(scan, X):
    """The chain mulls_extract_features runs, stage by stage in the oracle: the clouds of enum mulls_extract_cloud."""
    raw = abi.records(scan)
    if X.apply_dist_filter:
        raw = dist_filter(raw, X.min_dist_used, X.max_dist_used)
    if X.apply_scanner_filter:
        raw = scanner_filter(raw, X.self_ring_radius, X.ghost_radius, X.z_min, X.z_min_min)
    down = voxel_downsample(raw, X.vf_downsample_resolution)
    g, gd, ung = ground_filter(abi.points_of(down), X.ground) if len(down) else (down[:0], down[:0], down[:0])
    c, after = classify_nground(ung, X.classify)
    return [raw, g, gd, after] + c + [down]
