"""ctypes mirror of include/mulls_hip.h (the C ABI of libmulls_hip.so).

Pure declarations: structs, constants and helpers that turn numpy arrays into ``mulls_cloud`` / ``mulls_pair``
records.  No compute happens here.  The struct layout must stay in lock-step with include/mulls_hip.h
(tests/test_abi.py checks sizes against the compiled library).
"""
import ctypes as C

import numpy as np

NCLASS = 6
POINT_BYTES = 48
GROUND, PILLAR, FACADE, BEAM, ROOF, VERTEX = range(6)
CLASS_NAMES = ("ground", "pillar", "facade", "beam", "roof", "vertex")

MULLS_OK = 0
MULLS_E_INVALID = -100
MULLS_E_HIP = -101
MULLS_E_NO_DEVICE = -102
MULLS_E_UNSUPPORTED = -103
MULLS_E_IO = -104
MULLS_E_NOMEM = -105

# enum mulls_option
(OPT_HOST_STEP, OPT_RESIDENT_MIN_PAIRS, OPT_RESIDENT_MAX_PAIRS, OPT_FEW_LAUNCHES_MAX_PAIRS, OPT_SUBBATCHES, OPT_TWO_STREAMS, OPT_CERTIFICATES, OPT_CERT_SLACK_MIN,
 OPT_CERT_SLACK_MAX, OPT_CERT_SLACK_RATE, OPT_LDS_DEDUP, OPT_GRID_H0, OPT_BM_H0, OPT_LEAN_STAGING, OPT_DEBUG_STOP, OPT_DEBUG_TICK, OPT_SPLIT_MIN_PAIRS, OPT_SPLIT_MAX_PAIRS,
 OPT_FUSED_TGT_SETUP, OPT_STAGGER, OPT_STEP_LAUNCH_MAX_PAIRS, OPT_MIXED_TIERS, OPT_BIG_EARLY_SETS, OPT_KCERT, OPT_KCERT_MIN, OPT_ACCUM_WAVE_MIN_TRIPS, OPT_FIRST_DIRECT, OPT_SUM_STEP, OPT_COUNT) = range(29)

# numpy view of pcl::PointXYZINormal (48 B)
POINT_DTYPE = np.dtype(
    {
        "names": ["x", "y", "z", "nx", "ny", "nz", "intensity", "curvature"],
        "formats": [np.float32] * 8,
        "offsets": [0, 4, 8, 16, 20, 24, 32, 36],
        "itemsize": POINT_BYTES,
    }
)


class Cloud(C.Structure):
    _fields_ = [("pts", C.c_void_p), ("n", C.c_uint32), ("stride", C.c_uint32)]


class Pair(C.Structure):
    _fields_ = [
        ("tgt", Cloud * NCLASS),
        ("src", Cloud * NCLASS),
        ("src_down", Cloud * NCLASS),
        ("tgt_bound", C.c_double * 6),
        ("init_guess", C.c_double * 16),
    ]


class Params(C.Structure):
    _fields_ = [
        ("max_iter_num", C.c_int32),
        ("dis_thre_unit", C.c_float),
        ("converge_translation", C.c_float),
        ("converge_rotation_d", C.c_float),
        ("dis_thre_min", C.c_float),
        ("dis_thre_update_rate", C.c_float),
        ("used_feature_type", C.c_char * 8),
        ("weight_strategy", C.c_char * 8),
        ("z_xy_balanced_ratio", C.c_float),
        ("pt2pt_residual_window", C.c_float),
        ("pt2pl_residual_window", C.c_float),
        ("pt2li_residual_window", C.c_float),
        ("apply_intersection_filter", C.c_uint8),
        ("apply_motion_undistortion", C.c_uint8),
        ("normal_shooting_on", C.c_uint8),
        ("use_more_points", C.c_uint8),
        ("normal_bearing", C.c_float),
        ("keep_less_source_points", C.c_uint8),
        ("faithful", C.c_uint8),
        ("rejector_strict", C.c_uint8),
        ("reserved_", C.c_uint8),
        ("sigma_thre", C.c_float),
        ("min_neccessary_corr_ratio", C.c_float),
        ("max_bearable_rotation_d", C.c_float),
        ("rng_seed", C.c_uint64),
    ]


class IterTrace(C.Structure):
    _fields_ = [
        ("iter", C.c_int32),
        ("ncorr", C.c_uint32 * NCLASS),
        ("nsrc", C.c_uint32 * NCLASS),
        ("thr", C.c_float * NCLASS),
        ("atpa", C.c_double * 36),
        ("atpb", C.c_double * 6),
        ("x", C.c_double * 6),
    ]


class Result(C.Structure):
    _fields_ = [
        ("code", C.c_int32),
        ("iters", C.c_int32),
        ("T", C.c_double * 16),
        ("info", C.c_double * 36),
        ("sigma", C.c_float),
        ("confidence", C.c_float),
        ("ncorr", C.c_uint32 * NCLASS),
        ("nsrc0", C.c_uint32 * NCLASS),
        ("ntgt0", C.c_uint32 * NCLASS),
        ("singular", C.c_int32),
        ("cropped", C.c_int32),
        ("crop_box", C.c_double * 6),
        ("ms_total", C.c_float),
        ("trace", C.POINTER(IterTrace)),
        ("trace_cap", C.c_int32),
        ("trace_len", C.c_int32),
    ]

    # convenience views (row-major numpy matrices)
    def T_matrix(self):
        return np.array(self.T[:], dtype=np.float64).reshape(4, 4).T.copy()

    def info_matrix(self):
        return np.array(self.info[:], dtype=np.float64).reshape(6, 6).T.copy()



class MapParams(C.Structure):
    """mulls_map_params: MapManager::update_local_map's positional arguments (map_manager.h:21-31) + seed + tree state."""
    _fields_ = [
        ("local_map_radius", C.c_float),
        ("max_num_pts", C.c_int32),
        ("kept_vertex_num", C.c_int32),
        ("last_frame_reliable_radius", C.c_float),
        ("map_based_dynamic_removal_on", C.c_int32),
        ("used_feature_type", C.c_char * 8),
        ("dynamic_removal_center_radius", C.c_float),
        ("dynamic_dist_thre_min", C.c_float),
        ("dynamic_dist_thre_max", C.c_float),
        ("near_dist_thre", C.c_float),
        ("recalculate_feature_on", C.c_int32),
        ("rng_seed", C.c_uint64),
        ("tree_mode", C.c_int32),
        ("tree_used", C.c_char * 8),
        ("tree_box", C.c_double * 6),
    ]


class MapReport(C.Structure):
    _fields_ = [
        ("n", C.c_uint32 * NCLASS),
        ("frame_n", C.c_uint32 * NCLASS),
        ("feature_point_num", C.c_int32),
        ("dynamic_removal_ran", C.c_int32),
        ("local_bound", C.c_double * 6),
        ("bound", C.c_double * 6),
        ("ms_total", C.c_float),
    ]


def map_params(**kw):
    """update_local_map defaults (map_manager.h:21-31), overridden by keyword."""
    p = MapParams()
    p.local_map_radius, p.max_num_pts, p.kept_vertex_num, p.last_frame_reliable_radius = 80.0, 20000, 800, 60.0
    p.map_based_dynamic_removal_on = 0
    p.used_feature_type = b"111110"
    p.dynamic_removal_center_radius, p.dynamic_dist_thre_min, p.dynamic_dist_thre_max, p.near_dist_thre = 30.0, 0.3, 3.0, 0.03
    p.recalculate_feature_on, p.rng_seed, p.tree_mode, p.tree_used = 0, 0, 0, b"000000"
    for k, v in kw.items():
        if k == "tree_box":
            for i in range(6):
                p.tree_box[i] = float(v[i])
        elif k in ("used_feature_type", "tree_used"):
            setattr(p, k, v.encode() if isinstance(v, str) else v)
        else:
            setattr(p, k, v)
    return p


def colmajor16(M):
    return (C.c_double * 16)(*np.asarray(M, dtype=np.float64).T.reshape(-1))

class GroundParams(C.Structure):
    _fields_ = [
        ("min_grid_pt_num", C.c_int32),
        ("grid_resolution", C.c_float),
        ("max_height_difference", C.c_float),
        ("neighbor_height_diff", C.c_float),
        ("max_ground_height", C.c_float),
        ("ground_random_down_rate", C.c_int32),
        ("ground_random_down_down_rate", C.c_int32),
        ("nonground_random_down_rate", C.c_int32),
        ("reliable_neighbor_grid_num_thre", C.c_int32),
        ("estimate_ground_normal_method", C.c_int32),
        ("distance_weight_downsampling_method", C.c_int32),
        ("standard_distance", C.c_float),
        ("fixed_num_downsampling", C.c_uint8),
        ("apply_grid_wise_outlier_filter", C.c_uint8),
        ("reserved_", C.c_uint8 * 2),
        ("down_ground_fixed_num", C.c_int32),
        ("intensity_thre", C.c_float),
        ("outlier_std_scale", C.c_float),
        ("normal_estimation_radius", C.c_float),
        ("reserved2_", C.c_uint32),
        ("rng_seed", C.c_uint64),
    ]


def ground_params(**overrides):
    """fast_ground_filter's arguments as extract_semantic_pts passes them with script/config/lo_gflag_list_kitti_urban.txt, except
    the ground normal method (0 instead of the RANSAC of method 3) and the distance-inverse sampling (0 instead of 2)."""
    p = GroundParams()
    kw = dict(min_grid_pt_num=6, grid_resolution=2.5, max_height_difference=0.25, neighbor_height_diff=1.5, max_ground_height=2.0,
              ground_random_down_rate=12, ground_random_down_down_rate=3, nonground_random_down_rate=3, reliable_neighbor_grid_num_thre=0,
              estimate_ground_normal_method=0, distance_weight_downsampling_method=0, standard_distance=15.0, fixed_num_downsampling=0,
              apply_grid_wise_outlier_filter=0, down_ground_fixed_num=800, intensity_thre=150.0, outlier_std_scale=3.0, normal_estimation_radius=2.0,
              rng_seed=0)
    kw.update(overrides)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class ClassifyParams(C.Structure):
    _fields_ = [
        ("neighbor_searching_radius", C.c_float),
        ("neighbor_k", C.c_int32),
        ("neigh_k_min", C.c_int32),
        ("pca_down_rate", C.c_int32),
        ("edge_thre", C.c_float),
        ("planar_thre", C.c_float),
        ("edge_thre_down", C.c_float),
        ("planar_thre_down", C.c_float),
        ("extract_vertex_points_method", C.c_int32),
        ("curvature_thre", C.c_float),
        ("vertex_curvature_non_max_radius", C.c_float),
        ("linear_vertical_sin_high_thre", C.c_float),
        ("linear_vertical_sin_low_thre", C.c_float),
        ("planar_vertical_sin_high_thre", C.c_float),
        ("planar_vertical_sin_low_thre", C.c_float),
        ("fixed_num_downsampling", C.c_uint8),
        ("sharpen_with_nms", C.c_uint8),
        ("use_distance_adaptive_pca", C.c_uint8),
        ("reserved_", C.c_uint8),
        ("pillar_down_fixed_num", C.c_int32),
        ("facade_down_fixed_num", C.c_int32),
        ("beam_down_fixed_num", C.c_int32),
        ("roof_down_fixed_num", C.c_int32),
        ("unground_down_fixed_num", C.c_int32),
        ("beam_height_max", C.c_float),
        ("roof_height_min", C.c_float),
        ("feature_pts_ratio_guess", C.c_float),
        ("rng_seed", C.c_uint64),
    ]


CL_PILLAR, CL_BEAM, CL_FACADE, CL_ROOF, CL_PILLAR_DOWN, CL_BEAM_DOWN, CL_FACADE_DOWN, CL_ROOF_DOWN, CL_VERTEX, CL_COUNT = range(10)
CL_NAMES = ("pillar", "beam", "facade", "roof", "pillar_down", "beam_down", "facade_down", "roof_down", "vertex")


def classify_params(**overrides):
    """classify_nground_pts's arguments as extract_semantic_pts passes them for script/run_mulls_reg.sh (mulls_classify_default_params)."""
    p = ClassifyParams()
    kw = dict(neighbor_searching_radius=1.0, neighbor_k=50, neigh_k_min=8, pca_down_rate=1, edge_thre=0.65, planar_thre=0.65, edge_thre_down=0.75,
              planar_thre_down=0.75, extract_vertex_points_method=2, curvature_thre=0.10, vertex_curvature_non_max_radius=1.5,
              linear_vertical_sin_high_thre=0.94, linear_vertical_sin_low_thre=0.17, planar_vertical_sin_high_thre=0.98,
              planar_vertical_sin_low_thre=0.34, fixed_num_downsampling=0, sharpen_with_nms=1, use_distance_adaptive_pca=0, pillar_down_fixed_num=200,
              facade_down_fixed_num=800, beam_down_fixed_num=200, roof_down_fixed_num=200, unground_down_fixed_num=20000,
              beam_height_max=3.4028234663852886e38, roof_height_min=0.0, feature_pts_ratio_guess=0.3, rng_seed=0)
    kw.update(overrides)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class ExtractParams(C.Structure):
    _fields_ = [
        ("ground", GroundParams),
        ("classify", ClassifyParams),
        ("apply_scanner_filter", C.c_uint8),
        ("apply_dist_filter", C.c_uint8),
        ("reserved_", C.c_uint8 * 2),
        ("self_ring_radius", C.c_float),
        ("ghost_radius", C.c_float),
        ("z_min", C.c_float),
        ("z_min_min", C.c_float),
        ("vf_downsample_resolution", C.c_float),
        ("min_dist_used", C.c_double),
        ("max_dist_used", C.c_double),
    ]


EX_RAW, EX_GROUND, EX_GROUND_DOWN, EX_UNGROUND, EX_PILLAR, EX_VERTEX, EX_DOWN, EX_COUNT = 0, 1, 2, 3, 4, 12, 13, 14


def extract_params(ground=None, classify=None, apply_scanner_filter=0, approx_scanner_height=2.0, underground_thre=-7.0, apply_dist_filter=0,
                   min_dist_used=1.0, max_dist_used=120.0, vf_downsample_resolution=0.0):
    """The frame front end: dist_filter (test/mulls_slam.cpp:359-360) -> scanner filter (cfilter.hpp:2338-2346) -> voxel_downsample ->
    fast_ground_filter -> classify_nground_pts."""
    p = ExtractParams()
    p.ground = ground if ground is not None else ground_params()
    p.classify = classify if classify is not None else classify_params()
    p.apply_scanner_filter = apply_scanner_filter
    p.self_ring_radius, p.ghost_radius = 1.75, 20.0
    p.z_min = np.float32(-np.float32(approx_scanner_height) - 4.0)
    p.z_min_min = np.float32(-np.float32(approx_scanner_height) + np.float32(underground_thre))
    p.apply_dist_filter = apply_dist_filter
    p.min_dist_used, p.max_dist_used = min_dist_used, max_dist_used
    p.vf_downsample_resolution = vf_downsample_resolution
    return p


def records(a):
    """Any point array (POINT_DTYPE records or raw (n, 48) bytes) as contiguous raw (n, 48) uint8 records, every byte kept."""
    a = np.asarray(a)
    if a.dtype == np.uint8 and a.ndim == 2 and a.shape[1] == POINT_BYTES:
        return np.ascontiguousarray(a)
    pts = as_points(a)
    return pts.view(np.uint8).reshape(len(pts), POINT_BYTES)


def points_of(raw):
    """Raw (n, 48) records viewed as POINT_DTYPE (no copy)."""
    raw = np.ascontiguousarray(raw)
    return raw.reshape(-1).view(POINT_DTYPE)


class Profile(C.Structure):
    _fields_ = [
        ("ms_setup", C.c_double),
        ("ms_nn", C.c_double),
        ("ms_filter", C.c_double),
        ("ms_accum", C.c_double),
        ("ms_residual", C.c_double),
        ("launches_nn", C.c_int32),
        ("iterations", C.c_int32),
        ("nn_pair_evals", C.c_uint64),
        ("nn_src_pts", C.c_uint64),
        ("nn_tgt_pts", C.c_uint64),
        ("ms_host_step", C.c_double),
        ("ms_host_wait", C.c_double),
        ("ms_host_launch", C.c_double),
        ("nn_tgt_unique", C.c_uint64),
        ("nn_corr_pts", C.c_uint64),
        ("icp_fused_ms", C.c_double * 6),
        ("icp_search_ms", C.c_double * 24),
        ("icp_phase_ms", C.c_double * 6),
        ("ms_stage", C.c_double),
        ("ms_stage_pack", C.c_double),
        ("stage_bytes", C.c_uint64),
    ]


def default_params(**overrides):
    """mm_lls_icp's default arguments (cregistration.hpp:1114-1123)."""
    p = Params()
    p.max_iter_num = 20
    p.dis_thre_unit = 1.5
    p.converge_translation = 0.002
    p.converge_rotation_d = 0.01
    p.dis_thre_min = 0.4
    p.dis_thre_update_rate = 1.1
    p.used_feature_type = b"111110"
    p.weight_strategy = b"1101"
    p.z_xy_balanced_ratio = 1.0
    p.pt2pt_residual_window = 0.1
    p.pt2pl_residual_window = 0.1
    p.pt2li_residual_window = 0.1
    p.apply_intersection_filter = 1
    p.apply_motion_undistortion = 0
    p.normal_shooting_on = 0
    p.use_more_points = 0
    p.normal_bearing = 45.0
    p.keep_less_source_points = 0
    p.faithful = 1
    p.rejector_strict = 1
    p.sigma_thre = 0.5
    p.min_neccessary_corr_ratio = 0.03
    p.max_bearable_rotation_d = 45.0
    p.rng_seed = 0
    for k, v in overrides.items():
        if isinstance(v, str):
            v = v.encode()
        setattr(p, k, v)
    return p


def kitti_params(**overrides):
    """The positional values test/mulls_slam.cpp:642-648 passes with script/config/lo_gflag_list_kitti_urban.txt
    (column s2s/s2m)."""
    kw = dict(
        max_iter_num=20,
        dis_thre_unit=1.4,
        converge_translation=0.0005,
        converge_rotation_d=0.001,
        dis_thre_min=0.5,
        dis_thre_update_rate=1.1,
        used_feature_type="111000",
        weight_strategy="1111",
        z_xy_balanced_ratio=1.0,
        pt2pt_residual_window=0.05,
        pt2pl_residual_window=0.05,
        pt2li_residual_window=0.05,
        apply_intersection_filter=1,
        normal_bearing=20.0,
        sigma_thre=0.35,
        min_neccessary_corr_ratio=0.03,
        max_bearable_rotation_d=45.0,
    )
    kw.update(overrides)
    return default_params(**kw)


def make_points(xyz, normals=None, intensity=None, curvature=None):
    """Build a (n,) POINT_DTYPE array from column arrays."""
    xyz = np.asarray(xyz, dtype=np.float32).reshape(-1, 3)
    n = xyz.shape[0]
    pts = np.zeros(n, dtype=POINT_DTYPE)
    pts["x"], pts["y"], pts["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    if normals is not None:
        normals = np.asarray(normals, dtype=np.float32).reshape(-1, 3)
        pts["nx"], pts["ny"], pts["nz"] = normals[:, 0], normals[:, 1], normals[:, 2]
    if intensity is not None:
        pts["intensity"] = np.asarray(intensity, dtype=np.float32)
    if curvature is not None:
        pts["curvature"] = np.asarray(curvature, dtype=np.float32)
    return pts


def normal3(pts):
    """normal[3] of every record (offset 28; not a named field: the registration never reads it, the map's PCA refresh writes it)."""
    pts = np.ascontiguousarray(pts)
    return pts.view(np.uint8).reshape(len(pts), POINT_BYTES)[:, 28:32].copy().view(np.float32).reshape(len(pts))


def as_points(a):
    """Coerce any structured array carrying the eight point fields to a contiguous POINT_DTYPE array."""
    if a is None:
        return np.zeros(0, dtype=POINT_DTYPE)
    a = np.asarray(a)
    if a.dtype == POINT_DTYPE and a.flags["C_CONTIGUOUS"]:
        return a
    out = np.zeros(len(a), dtype=POINT_DTYPE)
    for k in POINT_DTYPE.names:
        out[k] = a[k]
    return out


def as_cloud(pts):
    """mulls_cloud borrowing a POINT_DTYPE numpy array (caller keeps the array alive)."""
    c = Cloud()
    if pts is None or len(pts) == 0:
        c.pts, c.n, c.stride = None, 0, POINT_BYTES
        return c
    assert pts.dtype == POINT_DTYPE and pts.flags["C_CONTIGUOUS"]
    c.pts = pts.ctypes.data
    c.n = len(pts)
    c.stride = POINT_BYTES
    return c


class PairData:
    """Owns the numpy arrays of one registration problem and exposes the ctypes ``Pair`` borrowing them."""

    def __init__(self, tgt, src, init_guess=None, tgt_bound=None, src_down=None):
        self.tgt = [as_points(t) for t in tgt]
        self.src = [as_points(s) for s in src]
        self.src_down = None if src_down is None else [as_points(s) for s in src_down]
        self.init_guess = np.eye(4) if init_guess is None else np.asarray(init_guess, dtype=np.float64)
        if tgt_bound is None:
            allp = [t for t in self.tgt if len(t)]
            if allp:
                mn = [min(float(t[k].min()) for t in allp) for k in ("x", "y", "z")]
                mx = [max(float(t[k].max()) for t in allp) for k in ("x", "y", "z")]
            else:
                mn, mx = [0.0] * 3, [0.0] * 3
            tgt_bound = mn + mx
        self.tgt_bound = [float(v) for v in tgt_bound]

    def fill(self, pair):
        for c in range(NCLASS):
            pair.tgt[c] = as_cloud(self.tgt[c])
            pair.src[c] = as_cloud(self.src[c])
            pair.src_down[c] = as_cloud(self.src_down[c]) if self.src_down is not None else as_cloud(None)
        for k in range(6):
            pair.tgt_bound[k] = self.tgt_bound[k]
        g = np.asarray(self.init_guess, dtype=np.float64).T.reshape(-1)  # column-major
        for k in range(16):
            pair.init_guess[k] = float(g[k])
        return pair

    def as_pair(self):
        return self.fill(Pair())


def make_pair_array(pairs):
    arr = (Pair * len(pairs))()
    for i, p in enumerate(pairs):
        p.fill(arr[i])
    return arr


def make_result_array(n, trace_cap=0):
    res = (Result * n)()
    traces = []
    if trace_cap:
        for i in range(n):
            t = (IterTrace * trace_cap)()
            traces.append(t)
            res[i].trace = C.cast(t, C.POINTER(IterTrace))
            res[i].trace_cap = trace_cap
    res._traces = traces  # keep alive
    return res
