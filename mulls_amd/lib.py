"""Python host-side binding of libmulls_hip.so (the C ABI in include/mulls_hip.h).

This is plumbing for the tests, bench.py and smoke(): it loads the in-tree shared library with ctypes and forwards
calls.  There is deliberately NO fallback: if the HIP library is missing or no gfx950 device is usable, every call
raises.  The product for C++ callers is the library itself plus include/cregistration_hip.hpp.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmulls_hip.so")
LIB_PATH = os.environ.get("MULLS_HIP_LIB", LIB_PATH)  # A/B timing of two builds of the same library on one box (tools/)
_LIB = None


class MullsError(RuntimeError):
    pass


def load():
    """Load libmulls_hip.so (raises if it has not been built: run `python -m mulls_amd.build`)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise MullsError("libmulls_hip.so is not built (python -m mulls_amd.build); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    lib.mulls_default_params.argtypes = [C.POINTER(abi.Params)]
    lib.mulls_default_params.restype = None
    lib.mulls_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.mulls_destroy.argtypes = [vp]
    lib.mulls_destroy.restype = None
    lib.mulls_last_error.argtypes = [vp]
    lib.mulls_last_error.restype = C.c_char_p
    lib.mulls_set_profiling.argtypes = [vp, C.c_int]
    lib.mulls_get_profile.argtypes = [vp, C.POINTER(abi.Profile)]
    lib.mulls_set_nn_mode.argtypes = [vp, C.c_int]
    lib.mulls_set_option.argtypes = [vp, C.c_int, C.c_double]
    lib.mulls_get_option.argtypes = [vp, C.c_int, C.POINTER(C.c_double)]
    lib.mulls_stream.argtypes = [vp]
    lib.mulls_stream.restype = vp
    lib.mulls_icp.argtypes = [vp, C.POINTER(abi.Pair), C.POINTER(abi.Params), C.POINTER(abi.Result)]
    lib.mulls_icp_batch.argtypes = [vp, C.POINTER(abi.Pair), C.c_int, C.POINTER(abi.Params), C.POINTER(abi.Result)]
    lib.mulls_pack_results.argtypes = [C.POINTER(abi.Result), C.c_int, C.c_void_p]
    lib.mulls_pack_results.restype = None
    lib.mulls_icp_batch_sharded.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(abi.Pair), C.c_int, C.POINTER(abi.Params), C.POINTER(abi.Result)]
    lib.mulls_pipe_create.argtypes = [C.c_int, C.c_int, C.POINTER(vp)]
    lib.mulls_pipe_destroy.argtypes = [vp]
    lib.mulls_pipe_destroy.restype = None
    lib.mulls_pipe_depth.argtypes = [vp]
    lib.mulls_pipe_ctx.argtypes = [vp, C.c_int]
    lib.mulls_pipe_ctx.restype = vp
    lib.mulls_pipe_set_option.argtypes = [vp, C.c_int, C.c_double]
    lib.mulls_icp_batch_begin.argtypes = [vp, C.POINTER(abi.Pair), C.c_int, C.POINTER(abi.Params), C.POINTER(abi.Result)]
    lib.mulls_icp_batch_end.argtypes = [vp, C.c_int]
    lib.mulls_batch_create.argtypes = [vp, C.POINTER(abi.Pair), C.c_int, C.POINTER(vp)]
    lib.mulls_batch_run.argtypes = [vp, vp, C.POINTER(abi.Params), C.POINTER(abi.Result)]
    lib.mulls_batch_destroy.argtypes = [vp, vp]
    lib.mulls_batch_destroy.restype = None
    lib.mulls_icp_3dof_ground.argtypes = [vp, C.POINTER(abi.Pair), C.POINTER(abi.Params), C.POINTER(abi.Result)]
    lib.mulls_icp_3dof_ground_batch.argtypes = [vp, C.POINTER(abi.Pair), C.c_int, C.POINTER(abi.Params), C.POINTER(abi.Result)]
    lib.mulls_icp_4dof_global.argtypes = [vp, C.POINTER(abi.Pair), C.c_float, C.POINTER(C.c_double), C.c_int, C.c_float, C.c_float, C.c_float,
                                          C.c_float, C.c_float, C.c_float, C.POINTER(abi.Result), C.POINTER(C.c_int), C.POINTER(C.c_float)]
    lib.mulls_stage_transform.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]
    lib.mulls_stage_correspond.argtypes = [vp, C.POINTER(abi.Cloud), C.POINTER(abi.Cloud), C.c_float, C.c_int, C.c_float, vp, vp, vp]
    lib.mulls_stage_accumulate.argtypes = [vp, C.c_int, C.POINTER(abi.Cloud), C.POINTER(abi.Cloud), vp, vp, vp, C.c_uint32, C.c_int,
                                           C.c_float, C.c_int, C.c_int, C.c_int, C.c_float, vp, vp]
    lib.mulls_map_default_params.argtypes = [C.POINTER(abi.MapParams)]
    lib.mulls_map_default_params.restype = None
    lib.mulls_map_create.argtypes = [vp, C.POINTER(vp)]
    lib.mulls_map_destroy.argtypes = [vp, vp]
    lib.mulls_map_destroy.restype = None
    lib.mulls_map_set.argtypes = [vp, vp, C.POINTER(abi.Cloud), C.POINTER(C.c_double)]
    lib.mulls_map_update.argtypes = [vp, vp, C.POINTER(abi.Cloud), C.POINTER(C.c_double), C.POINTER(abi.MapParams), C.POINTER(abi.MapReport)]
    lib.mulls_map_cloud.argtypes = [vp, vp, C.c_int, C.POINTER(abi.Cloud)]
    lib.mulls_map_pose.argtypes = [vp, vp, C.POINTER(C.c_double)]
    lib.mulls_map_download.argtypes = [vp, vp, C.c_int, vp, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.mulls_map_frame_download.argtypes = [vp, vp, C.c_int, vp, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.mulls_ground_default_params.argtypes = [C.POINTER(abi.GroundParams)]
    lib.mulls_ground_default_params.restype = None
    lib.mulls_ground_filter.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.POINTER(abi.GroundParams), vp, C.c_uint32, vp, C.c_uint32, vp, C.c_uint32,
                                        C.POINTER(C.c_uint32)]
    lib.mulls_classify_default_params.argtypes = [C.POINTER(abi.ClassifyParams)]
    lib.mulls_classify_default_params.restype = None
    lib.mulls_classify_nground.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.POINTER(abi.ClassifyParams), C.POINTER(vp), C.POINTER(C.c_uint32),
                                           C.POINTER(C.c_uint32), vp, C.POINTER(C.c_uint32)]
    lib.mulls_extract_default_params.argtypes = [C.POINTER(abi.ExtractParams)]
    lib.mulls_extract_default_params.restype = None
    lib.mulls_extract_features.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.POINTER(abi.ExtractParams), C.POINTER(vp), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.mulls_block_create.argtypes = [vp, C.POINTER(vp)]
    lib.mulls_block_destroy.argtypes = [vp, vp]
    lib.mulls_block_destroy.restype = None
    lib.mulls_extract_features_resident.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.POINTER(abi.ExtractParams), vp, C.POINTER(C.c_uint32)]
    lib.mulls_block_cloud.argtypes = [vp, vp, C.c_int, C.POINTER(abi.Cloud)]
    lib.mulls_block_download.argtypes = [vp, vp, C.c_int, vp, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.mulls_motion_compensate.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_double), C.c_float]
    lib.mulls_block_motion_compensate.argtypes = [vp, vp, C.POINTER(C.c_double), C.c_int]
    lib.mulls_voxel_downsample.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_float, vp, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.mulls_io_read_kitti_bin.argtypes = [C.c_char_p, vp, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.mulls_io_read_pcd.argtypes = [C.c_char_p, vp, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.mulls_io_write_pcd.argtypes = [C.c_char_p, vp, C.c_uint32, C.c_uint32, C.c_int]
    lib.mulls_io_write_pose.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.c_int]
    _LIB = lib
    return lib


EXPORTS = [
    "mulls_default_params", "mulls_create", "mulls_destroy", "mulls_last_error", "mulls_set_profiling", "mulls_get_profile",
    "mulls_stream", "mulls_set_nn_mode", "mulls_icp", "mulls_icp_batch", "mulls_batch_create", "mulls_batch_run", "mulls_batch_destroy",
    "mulls_stage_transform", "mulls_stage_correspond", "mulls_stage_accumulate", "mulls_icp_3dof_ground", "mulls_icp_3dof_ground_batch",
    "mulls_icp_4dof_global", "mulls_map_default_params", "mulls_map_create", "mulls_map_destroy", "mulls_map_set", "mulls_map_update",
    "mulls_map_cloud", "mulls_map_pose", "mulls_map_download", "mulls_map_frame_download", "mulls_io_read_kitti_bin", "mulls_io_read_pcd",
    "mulls_io_write_pcd", "mulls_io_write_pose", "mulls_ground_default_params", "mulls_ground_filter",
    "mulls_classify_default_params", "mulls_classify_nground", "mulls_extract_default_params", "mulls_extract_features", "mulls_voxel_downsample",
    "mulls_set_option", "mulls_get_option", "mulls_block_create", "mulls_block_destroy", "mulls_extract_features_resident", "mulls_block_cloud", "mulls_block_download",
    "mulls_motion_compensate", "mulls_block_motion_compensate",
    "mulls_pack_results", "mulls_icp_batch_sharded", "mulls_pipe_create", "mulls_pipe_destroy", "mulls_pipe_depth", "mulls_pipe_ctx", "mulls_pipe_set_option", "mulls_icp_batch_begin", "mulls_icp_batch_end",
]


def _read(fn, name, path):
    n = C.c_uint32(0)
    rc = fn(path.encode(), None, 0, C.byref(n))
    if rc != 0:
        raise MullsError("%s(%s) failed with %d" % (name, path, rc))
    out = np.zeros(n.value, abi.POINT_DTYPE)
    if n.value:
        rc = fn(path.encode(), out.ctypes.data_as(C.c_void_p), n.value, C.byref(n))
        if rc != 0:
            raise MullsError("%s(%s) failed with %d" % (name, path, rc))
    return out


def read_kitti_bin(path):
    """DataIo::read_bin_file: KITTI .bin -> POINT_DTYPE array (intensity * 255, trailing zero point like the reference)."""
    return _read(load().mulls_io_read_kitti_bin, "mulls_io_read_kitti_bin", path)


def read_pcd(path):
    """DataIo::read_pcd_file: PCD v0.7 (ascii / binary) of PointXYZINormal -> POINT_DTYPE array."""
    return _read(load().mulls_io_read_pcd, "mulls_io_read_pcd", path)


def write_pcd(path, pts, binary=True):
    pts = abi.as_points(pts)
    rc = load().mulls_io_write_pcd(path.encode(), pts.ctypes.data_as(C.c_void_p), len(pts), abi.POINT_BYTES, int(binary))
    if rc != 0:
        raise MullsError("mulls_io_write_pcd(%s) failed with %d" % (path, rc))


def write_pose(path, T, append=False):
    rc = load().mulls_io_write_pose(path.encode(), abi.colmajor16(T), int(append))
    if rc != 0:
        raise MullsError("mulls_io_write_pose(%s) failed with %d" % (path, rc))


def icp_batch_sharded(contexts, pairs, params, marshalled=None):
    """mulls_icp_batch_sharded: the pairs block-partitioned over `contexts` (Context objects: one per GPU, or several on one), one host thread each."""
    lib = load()
    arr, res = marshalled if marshalled is not None else (abi.make_pair_array(pairs), abi.make_result_array(len(pairs)))
    hs = (C.c_void_p * len(contexts))(*[c.h for c in contexts])
    rc = lib.mulls_icp_batch_sharded(hs, len(contexts), arr, len(pairs), C.byref(params), res)
    if rc != 0:
        raise MullsError("mulls_icp_batch_sharded failed with %d: %s" % (rc, "; ".join((lib.mulls_last_error(c.h) or b"").decode() for c in contexts)))
    return res


class Pipe:
    """mulls_pipe: mulls_icp_batch calls from host buffers in flight on alternating contexts of one device (begin / end by ticket)."""

    def __init__(self, device=0, depth=2):
        self.lib = load()
        self.h = C.c_void_p()
        rc = self.lib.mulls_pipe_create(device, depth, C.byref(self.h))
        if rc != 0:
            raise MullsError("mulls_pipe_create(device=%d, depth=%d) failed with %d" % (device, depth, rc))
        self._keep = {}

    def set_option(self, option, value):
        rc = self.lib.mulls_pipe_set_option(self.h, option, float(value))
        if rc != 0:
            raise MullsError("mulls_pipe_set_option failed with %d" % rc)

    def begin(self, pairs, params, marshalled=None):
        """-> ticket; the marshalled arrays (and through them the clouds) are kept alive until end(ticket)"""
        arr, res = marshalled if marshalled is not None else (abi.make_pair_array(pairs), abi.make_result_array(len(pairs)))
        t = self.lib.mulls_icp_batch_begin(self.h, arr, len(pairs), C.byref(params), res)
        if t < 0:
            raise MullsError("mulls_icp_batch_begin failed with %d" % t)
        self._keep[t] = (arr, res, pairs, params)
        return t

    def end(self, ticket):
        rc = self.lib.mulls_icp_batch_end(self.h, ticket)
        arr, res, _, _ = self._keep.pop(ticket)
        if rc != 0:
            lane = ticket % self.lib.mulls_pipe_depth(self.h)
            raise MullsError("mulls_icp_batch_end(%d) failed with %d: %s" % (ticket, rc, (self.lib.mulls_last_error(self.lib.mulls_pipe_ctx(self.h, lane)) or b"").decode()))
        return res

    def lane_profile(self, lane):
        """mulls_profile of one lane's context (its latest call)"""
        pf = abi.Profile()
        self.lib.mulls_get_profile(self.lib.mulls_pipe_ctx(self.h, lane), C.byref(pf))
        return pf

    def close(self):
        if self.h:
            self.lib.mulls_pipe_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """mulls_ctx: one HIP stream on one device."""

    def __init__(self, device=0):
        self.lib = load()
        self.h = C.c_void_p()
        rc = self.lib.mulls_create(device, C.byref(self.h))
        if rc != 0:
            raise MullsError("mulls_create(device=%d) failed with %d (no usable gfx950 device?)" % (device, rc))

    def close(self):
        if self.h:
            self.lib.mulls_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise MullsError("%s failed with %d: %s" % (what, rc, self.lib.mulls_last_error(self.h).decode()))

    def set_profiling(self, on):
        self._check(self.lib.mulls_set_profiling(self.h, int(on)), "mulls_set_profiling")

    def set_nn_mode(self, mode):
        """0 auto, 1 LDS-tiled brute force, 2 uniform grid in global memory, 3 uniform grid staged in LDS (lock-step), 4 device-resident loop."""
        self._check(self.lib.mulls_set_nn_mode(self.h, int(mode)), "mulls_set_nn_mode")

    def set_option(self, option, value):
        """enum mulls_option (abi.OPT_*): execution options of the context; none of them changes a result."""
        self._check(self.lib.mulls_set_option(self.h, int(option), float(value)), "mulls_set_option")

    def get_option(self, option):
        v = C.c_double(0)
        self._check(self.lib.mulls_get_option(self.h, int(option), C.byref(v)), "mulls_get_option")
        return v.value

    def profile(self):
        p = abi.Profile()
        self._check(self.lib.mulls_get_profile(self.h, C.byref(p)), "mulls_get_profile")
        return p

    def stream(self):
        return self.lib.mulls_stream(self.h)

    # --- mm_lls_icp replacements -------------------------------------------------------------------------------
    def icp(self, pair, params, trace_cap=0):
        res = abi.make_result_array(1, trace_cap)
        p = pair.as_pair()
        self._check(self.lib.mulls_icp(self.h, C.byref(p), C.byref(params), res), "mulls_icp")
        return res

    def icp_batch(self, pairs, params, trace_cap=0, marshalled=None):
        """mulls_icp_batch.  marshalled: (the mulls_pair array of `pairs`, a result array) made earlier — callers timing the library leave the
        Python-side marshalling out that way; last_call_s = seconds spent inside the C call."""
        arr, res = marshalled if marshalled is not None else (abi.make_pair_array(pairs), abi.make_result_array(len(pairs), trace_cap))
        import time

        t0 = time.perf_counter()
        rc = self.lib.mulls_icp_batch(self.h, arr, len(pairs), C.byref(params), res)
        self.last_call_s = time.perf_counter() - t0
        self._check(rc, "mulls_icp_batch")
        return res

    def batch(self, pairs):
        return Batch(self, pairs)

    def block(self):
        """an empty device-resident feature block (Block.extract fills it)"""
        return Block(self)

    def local_map(self, clouds=None, pose=None):
        """Device-resident local map (mulls_map_*), optionally initialised from six host class clouds and a pose_lo."""
        return LocalMap(self, clouds, pose)

    # --- variants ------------------------------------------------------------------------------------
    def icp_3dof_ground(self, pairs, params, trace_cap=0):
        """lls_icp_3dof_ground on one PairData or a list of them."""
        plist = pairs if isinstance(pairs, (list, tuple)) else [pairs]
        arr = abi.make_pair_array(plist)
        res = abi.make_result_array(len(plist), trace_cap)
        self._check(self.lib.mulls_icp_3dof_ground_batch(self.h, arr, len(plist), C.byref(params), res), "mulls_icp_3dof_ground_batch")
        return res

    def icp_4dof_global(self, pair, heading_step_d, station, max_iter_num=20, dis_thre_unit=1.5, converge_translation=0.005,
                        converge_rotation_d=0.05, dis_thre_min=0.5, dis_thre_update_rate=1.05, max_bearable_rotation_d=15.0):
        """mm_lls_icp_4dof_global.  Returns (results, success, best_heading_deg)."""
        res = abi.make_result_array(1, 0)
        p = pair.as_pair()
        ok, best = C.c_int(0), C.c_float(0)
        st = (C.c_double * 3)(*station)
        self._check(self.lib.mulls_icp_4dof_global(self.h, C.byref(p), heading_step_d, st, int(max_iter_num), dis_thre_unit, converge_translation,
                                                   converge_rotation_d, dis_thre_min, dis_thre_update_rate, max_bearable_rotation_d, res,
                                                   C.byref(ok), C.byref(best)), "mulls_icp_4dof_global")
        return res, bool(ok.value), float(best.value)

    # --- feature extraction, first stage -------------------------------------------------------------------
    def ground_filter(self, pts, params):
        """CFilter::fast_ground_filter on the device.  Returns (ground, ground_down, unground) as (n, 48) uint8 record arrays."""
        pts = abi.as_points(pts)
        n = len(pts)
        raw = [np.zeros(max(n, 1) * abi.POINT_BYTES, np.uint8) for _ in range(3)]
        nout = (C.c_uint32 * 3)()
        self._check(self.lib.mulls_ground_filter(self.h, pts.ctypes.data_as(C.c_void_p), n, abi.POINT_BYTES, C.byref(params), raw[0].ctypes.data_as(C.c_void_p), n,
                                                 raw[1].ctypes.data_as(C.c_void_p), n, raw[2].ctypes.data_as(C.c_void_p), n, nout), "mulls_ground_filter")
        return [raw[k][: nout[k] * abi.POINT_BYTES].reshape(nout[k], abi.POINT_BYTES).copy() for k in range(3)]

    # --- feature extraction, second stage -----------------------------------------------------------------------------------
    def classify_nground(self, pts, params, with_cloud_in=False):
        """CFilter::classify_nground_pts on the device.  Returns the nine clouds of enum mulls_classify_cloud as (n, 48) uint8 record arrays
        (and, with_cloud_in, the input cloud as the function leaves it)."""
        raw_in = abi.records(pts)
        n = len(raw_in)
        outs = [np.zeros((max(n, 1), abi.POINT_BYTES), np.uint8) for _ in range(abi.CL_COUNT)]
        out_p = (C.c_void_p * abi.CL_COUNT)(*[o.ctypes.data for o in outs])
        cap = (C.c_uint32 * abi.CL_COUNT)(*([n] * abi.CL_COUNT))
        nout = (C.c_uint32 * abi.CL_COUNT)()
        after = np.zeros((max(n, 1), abi.POINT_BYTES), np.uint8)
        n_after = C.c_uint32(0)
        self._check(self.lib.mulls_classify_nground(self.h, raw_in.ctypes.data_as(C.c_void_p), n, abi.POINT_BYTES, C.byref(params), out_p, cap, nout,
                                                    after.ctypes.data_as(C.c_void_p) if with_cloud_in else None, C.byref(n_after)), "mulls_classify_nground")
        res = [outs[k][: nout[k]].copy() for k in range(abi.CL_COUNT)]
        return (res, after[: n_after.value].copy()) if with_cloud_in else res

    def extract_features(self, scan, params):
        """CFilter::extract_semantic_pts' chain on the device in one call.  Returns the clouds of enum mulls_extract_cloud as (n, 48) uint8 arrays."""
        raw_in = abi.records(scan)
        n = len(raw_in)
        if getattr(self, "_ex_cap", 0) < n:  # receive buffers kept between calls (fresh pages cost more than the transfers)
            self._ex_out = [np.zeros((max(n, 1), abi.POINT_BYTES), np.uint8) for _ in range(abi.EX_COUNT)]
            self._ex_cap = n
        outs = self._ex_out
        out_p = (C.c_void_p * abi.EX_COUNT)(*[o.ctypes.data for o in outs])
        cap = (C.c_uint32 * abi.EX_COUNT)(*([n] * abi.EX_COUNT))
        filtered = bool(params.apply_scanner_filter or params.apply_dist_filter)
        voxels = not (params.vf_downsample_resolution < 0.001)
        if not filtered:
            cap[abi.EX_RAW] = 0  # pc_raw is the scan itself: not sent back
        if not voxels:
            cap[abi.EX_DOWN] = 0  # pc_down is pc_raw
        nout = (C.c_uint32 * abi.EX_COUNT)()
        self._check(self.lib.mulls_extract_features(self.h, raw_in.ctypes.data_as(C.c_void_p), n, abi.POINT_BYTES, C.byref(params), out_p, cap, nout),
                    "mulls_extract_features")
        res = [outs[k][: nout[k]].copy() for k in range(abi.EX_COUNT)]
        if not filtered:
            res[abi.EX_RAW] = raw_in
        if not voxels:
            res[abi.EX_DOWN] = res[abi.EX_RAW]
        return res

    def voxel_downsample(self, pts, voxel_size):
        """CFilter::voxel_downsample (cfilter.hpp:83-160).  Returns pc_down as (n, 48) uint8 records."""
        raw_in = abi.records(pts)
        n = len(raw_in)
        out = np.zeros((max(n, 1), abi.POINT_BYTES), np.uint8)
        n_out = C.c_uint32(0)
        self._check(self.lib.mulls_voxel_downsample(self.h, raw_in.ctypes.data_as(C.c_void_p), n, abi.POINT_BYTES, C.c_float(voxel_size),
                                                    out.ctypes.data_as(C.c_void_p), n, C.byref(n_out)), "mulls_voxel_downsample")
        return out[: n_out.value].copy()

    # --- stage-level entry points --------------------------------------------------------------------------------
    def motion_compensate(self, pts, Tran, s_ambiguous_thre=0.0):
        """CFilter::apply_motion_compensation on a copy of a host cloud (structured array of 48-byte records)."""
        raw = abi.records(pts).copy()  # every byte of the records kept (a copy of a record array would drop the bytes between its fields)
        Tc = abi.colmajor16(Tran)
        self._check(self.lib.mulls_motion_compensate(self.h, C.c_void_p(raw.ctypes.data), len(raw), abi.POINT_BYTES, Tc, C.c_float(s_ambiguous_thre)), "mulls_motion_compensate")
        return abi.points_of(raw)

    def transform(self, pts, T):
        pts = np.ascontiguousarray(pts).copy()
        Tc = (C.c_double * 16)(*np.asarray(T, dtype=np.float64).T.reshape(-1))
        self._check(self.lib.mulls_stage_transform(self.h, pts.ctypes.data, len(pts), abi.POINT_BYTES, Tc), "mulls_stage_transform")
        return pts

    def correspond(self, src, tgt, dis_thre, normal_check=True, angle_deg=45.0):
        n = len(src)
        match = np.zeros(n, np.int32)
        d2 = np.zeros(n, np.float32)
        flags = np.zeros(n, np.uint8)
        cs, ct = abi.as_cloud(src), abi.as_cloud(tgt)
        self._check(self.lib.mulls_stage_correspond(self.h, C.byref(cs), C.byref(ct), dis_thre, int(normal_check), angle_deg,
                                                    match.ctypes.data, d2.ctypes.data, flags.ctypes.data), "mulls_stage_correspond")
        return match, d2, flags

    def accumulate(self, metric, src, tgt, corr_src, corr_tgt, corr_d2, iter_num, class_weight, dist_w, resid_w, inten_w, window):
        corr_src = np.ascontiguousarray(corr_src, np.int32)
        corr_tgt = np.ascontiguousarray(corr_tgt, np.int32)
        corr_d2 = np.ascontiguousarray(corr_d2, np.float32)
        out = np.zeros(27, np.float64)
        w = np.zeros(len(corr_src), np.float32)
        cs, ct = abi.as_cloud(src), abi.as_cloud(tgt)
        self._check(self.lib.mulls_stage_accumulate(self.h, int(metric), C.byref(cs), C.byref(ct), corr_src.ctypes.data, corr_tgt.ctypes.data,
                                                    corr_d2.ctypes.data, len(corr_src), int(iter_num), class_weight, int(dist_w), int(resid_w),
                                                    int(inten_w), window, out.ctypes.data, w.ctypes.data), "mulls_stage_accumulate")
        return out, w


class Batch:
    """mulls_batch: pairs staged in HBM once, registered any number of times."""

    def __init__(self, ctx, pairs):
        self.ctx = ctx
        self.n = len(pairs)
        self._pairs = pairs  # keep numpy storage alive during create
        arr = abi.make_pair_array(pairs)
        self.h = C.c_void_p()
        ctx._check(ctx.lib.mulls_batch_create(ctx.h, arr, self.n, C.byref(self.h)), "mulls_batch_create")

    def run(self, params, trace_cap=0, results=None):
        res = results if results is not None else abi.make_result_array(self.n, trace_cap)
        self.ctx._check(self.ctx.lib.mulls_batch_run(self.ctx.h, self.h, C.byref(params), res), "mulls_batch_run")
        return res

    def close(self):
        if self.h:
            self.ctx.lib.mulls_batch_destroy(self.ctx.h, self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Block:
    """mulls_block: the feature clouds of one scan, device-resident (mulls_extract_features_resident)."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.h = C.c_void_p()
        self.n = [0] * abi.EX_COUNT
        ctx._check(ctx.lib.mulls_block_create(ctx.h, C.byref(self.h)), "mulls_block_create")

    def close(self):
        if self.h and self.ctx.h:
            self.ctx.lib.mulls_block_destroy(self.ctx.h, self.h)
        self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def extract(self, scan, params):
        raw_in = abi.records(scan)
        nout = (C.c_uint32 * abi.EX_COUNT)()
        self.ctx._check(self.ctx.lib.mulls_extract_features_resident(self.ctx.h, raw_in.ctypes.data_as(C.c_void_p), len(raw_in), abi.POINT_BYTES, C.byref(params), self.h, nout),
                        "mulls_extract_features_resident")
        self.n = list(nout)
        return self

    def cloud(self, which):
        """device-resident mulls_cloud of cloud `which` (enum mulls_extract_cloud: abi.EX_*)"""
        c = abi.Cloud()
        self.ctx._check(self.ctx.lib.mulls_block_cloud(self.ctx.h, self.h, which, C.byref(c)), "mulls_block_cloud")
        return c

    def download(self, which):
        n = C.c_uint32(0)
        self.ctx._check(self.ctx.lib.mulls_block_download(self.ctx.h, self.h, which, None, 0, C.byref(n)), "mulls_block_download")
        out = np.zeros((n.value, abi.POINT_BYTES), np.uint8)
        if n.value:
            self.ctx._check(self.ctx.lib.mulls_block_download(self.ctx.h, self.h, which, out.ctypes.data_as(C.c_void_p), n.value, C.byref(n)), "mulls_block_download")
        return out

    def motion_compensate(self, Tran, undistort_keypoints=False):
        """the two batch_apply_motion_compensation calls of test/mulls_slam.cpp:706-710 on the block's clouds, in place on the device"""
        self.ctx._check(self.ctx.lib.mulls_block_motion_compensate(self.ctx.h, self.h, abi.colmajor16(Tran), 1 if undistort_keypoints else 0), "mulls_block_motion_compensate")

    def class_clouds(self, down):
        """the six class clouds in the ABI's order (ground, pillar, facade, beam, roof, vertex) as device clouds: the *_down ones (+ pc_vertex) or the full ones"""
        P = abi.EX_PILLAR
        which = [abi.EX_GROUND_DOWN, P + 4, P + 6, P + 5, P + 7, abi.EX_VERTEX] if down else [abi.EX_GROUND, P, P + 2, P + 1, P + 3, abi.EX_VERTEX]
        return [self.cloud(k) for k in which]


class LocalMap:
    """mulls_map: the six undown class clouds of the reference's local_map cloudblock, kept in HBM between frames."""

    def __init__(self, ctx, clouds=None, pose=None):
        self.ctx = ctx
        self.h = C.c_void_p()
        ctx._check(ctx.lib.mulls_map_create(ctx.h, C.byref(self.h)), "mulls_map_create")
        if clouds is not None:
            self.set(clouds, np.eye(4) if pose is None else pose)

    def close(self):
        if self.h and self.ctx.h:
            self.ctx.lib.mulls_map_destroy(self.ctx.h, self.h)
        self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _clouds(clouds):
        """six clouds, each a point array or an abi.Cloud (a device cloud of a Block / LocalMap)"""
        keep = [c if isinstance(c, abi.Cloud) else abi.as_points(c) for c in clouds]
        arr = (abi.Cloud * abi.NCLASS)()
        for c in range(abi.NCLASS):
            arr[c] = keep[c] if isinstance(keep[c], abi.Cloud) else abi.as_cloud(keep[c])
        return arr, keep

    def set(self, clouds, pose):
        arr, keep = self._clouds(clouds)
        self.ctx._check(self.ctx.lib.mulls_map_set(self.ctx.h, self.h, arr, abi.colmajor16(pose)), "mulls_map_set")

    def update(self, frame_down, frame_pose, params):
        """update_local_map with last_target_cblock = (frame_down[0..4] = pc_*_down, frame_down[5] = pc_vertex, frame_pose)."""
        arr, keep = self._clouds(frame_down)
        rep = abi.MapReport()
        self.ctx._check(self.ctx.lib.mulls_map_update(self.ctx.h, self.h, arr, abi.colmajor16(frame_pose), C.byref(params), C.byref(rep)),
                        "mulls_map_update")
        return rep

    def cloud(self, cls):
        """Device-resident mulls_cloud of class `cls` (usable as a registration target until the next set / update)."""
        c = abi.Cloud()
        self.ctx._check(self.ctx.lib.mulls_map_cloud(self.ctx.h, self.h, cls, C.byref(c)), "mulls_map_cloud")
        return c

    def pose(self):
        m = (C.c_double * 16)()
        self.ctx._check(self.ctx.lib.mulls_map_pose(self.ctx.h, self.h, m), "mulls_map_pose")
        return np.array(m[:]).reshape(4, 4).T.copy()

    def _download(self, fn, name, cls):
        n = C.c_uint32(0)
        self.ctx._check(fn(self.ctx.h, self.h, cls, None, 0, C.byref(n)), name)
        out = np.zeros(n.value, abi.POINT_DTYPE)
        if n.value:
            self.ctx._check(fn(self.ctx.h, self.h, cls, out.ctypes.data_as(C.c_void_p), n.value, C.byref(n)), name)
        return out

    def download(self, cls):
        return self._download(self.ctx.lib.mulls_map_download, "mulls_map_download", cls)

    def frame_download(self, cls):
        return self._download(self.ctx.lib.mulls_map_frame_download, "mulls_map_frame_download", cls)

    def icp(self, src, params, init_guess=None, tgt_bound=None, trace_cap=0):
        """mm_lls_icp with this map as block1 (target clouds stay on the device) and `src` as block2's six source clouds."""
        keep = [s if isinstance(s, abi.Cloud) else abi.as_points(s) for s in src]
        pair = abi.Pair()
        for c in range(abi.NCLASS):
            pair.tgt[c] = self.cloud(c)
            pair.src[c] = keep[c] if isinstance(keep[c], abi.Cloud) else abi.as_cloud(keep[c])
            pair.src_down[c] = abi.as_cloud(None)
        for k in range(6):
            pair.tgt_bound[k] = float(tgt_bound[k])
        g = np.asarray(np.eye(4) if init_guess is None else init_guess, dtype=np.float64).T.reshape(-1)
        for k in range(16):
            pair.init_guess[k] = float(g[k])
        res = abi.make_result_array(1, trace_cap)
        self.ctx._check(self.ctx.lib.mulls_icp(self.ctx.h, C.byref(pair), C.byref(params), res), "mulls_icp")
        return res
