"""Seeded synthetic spinning-LiDAR scenes that emit MULLS feature-class clouds directly.

The hot path never sees raw scans: it consumes six feature-class clouds per block (ground / pillar / facade / beam /
roof / vertex) carrying unit normals (planar classes) or unit principal directions (linear classes) in the normal
fields (reference: include/common/pca.hpp:437-454).  Feature extraction is out of scope, so this generator ray-casts
an urban-canyon scene and labels every return by the primitive it hit, with the analytic normal / direction.

Used by the tests, bench.py and __graft_entry__.smoke() as the data source; it is not part of the compute path.
"""
import numpy as np

from . import abi


def _rot_z(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


def _rot_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]])


def _rot_x(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[1.0, 0.0, 0.0], [0.0, c, -s], [0.0, s, c]])


def se3(tx=0.0, ty=0.0, tz=0.0, roll=0.0, pitch=0.0, yaw=0.0):
    T = np.eye(4)
    T[:3, :3] = _rot_z(yaw) @ _rot_y(pitch) @ _rot_x(roll)
    T[:3, 3] = (tx, ty, tz)
    return T


class Scene:
    """World frame: ground plane z = 0, street along x."""

    def __init__(self, seed, n_poles=60, n_rails=30, n_facades=4, n_roofs=6, extent=60.0):
        rng = np.random.default_rng(seed)
        self.sensor_height = 1.73
        # facades: vertical rectangles parallel to the street (|y| in [8,15]) and a few across it (|x| in [20,40])
        fac = []
        for k in range(n_facades):
            if k % 2 == 0:
                y = rng.uniform(8.0, 15.0) * (1 if (k // 2) % 2 == 0 else -1)
                fac.append(dict(p0=np.array([rng.uniform(-10, 10), y, 0.0]), n=np.array([0.0, -np.sign(y), 0.0]),
                                t=np.array([1.0, 0.0, 0.0]), half_len=rng.uniform(25, extent), height=rng.uniform(8, 20),
                                base=rng.uniform(40, 200)))
            else:
                x = rng.uniform(20.0, 40.0) * (1 if (k // 2) % 2 == 0 else -1)
                fac.append(dict(p0=np.array([x, rng.uniform(-3, 3), 0.0]), n=np.array([-np.sign(x), 0.0, 0.0]),
                                t=np.array([0.0, 1.0, 0.0]), half_len=rng.uniform(6, 14), height=rng.uniform(8, 20),
                                base=rng.uniform(40, 200)))
        self.facades = fac
        # roofs / canopies: horizontal rectangles
        self.roofs = [dict(c=np.array([rng.uniform(-extent, extent), rng.uniform(-7, 7), rng.uniform(4.0, 8.0)]),
                           hx=rng.uniform(3, 10), hy=rng.uniform(2, 5), base=rng.uniform(40, 200)) for _ in range(n_roofs)]
        # poles: vertical cylinders
        self.poles = [dict(c=np.array([rng.uniform(-extent, extent), rng.uniform(3.5, 7.5) * rng.choice([-1, 1])]), r=0.15,
                           h=rng.uniform(3.5, 6.0), base=rng.uniform(40, 200)) for _ in range(n_poles)]
        # rails: horizontal cylinders, direction in the xy-plane
        rails = []
        for _ in range(n_rails):
            a = rng.uniform(-0.3, 0.3) + (np.pi / 2 if rng.random() < 0.3 else 0.0)
            rails.append(dict(c=np.array([rng.uniform(-extent, extent), rng.uniform(3.0, 7.0) * rng.choice([-1, 1]), rng.uniform(0.6, 3.5)]),
                              u=np.array([np.cos(a), np.sin(a), 0.0]), r=0.08, half_len=rng.uniform(2.0, 8.0), base=rng.uniform(40, 200)))
        self.rails = rails
        self.ground_base = rng.uniform(20, 80)


def raycast(scene, pose, n_beams=64, n_az=1900, elev_deg=(-24.8, 2.0), max_range=120.0, range_noise=0.02, seed=0):
    with np.errstate(all="ignore"):
        return _raycast(scene, pose, n_beams, n_az, elev_deg, max_range, range_noise, seed)


def _raycast(scene, pose, n_beams, n_az, elev_deg, max_range, range_noise, seed):
    """Cast one revolution from the sensor at world pose `pose` (4x4, sensor->world).

    Returns a dict of per-return arrays in the SENSOR frame: xyz (n,3), nrm (n,3), cls (n,), intensity (n,), t (n,) in [0,1].
    """
    rng = np.random.default_rng(seed)
    el = np.deg2rad(np.linspace(elev_deg[0], elev_deg[1], n_beams))
    az = np.linspace(0.0, 2 * np.pi, n_az, endpoint=False)
    EL, AZ = np.meshgrid(el, az, indexing="ij")
    d_s = np.stack([np.cos(EL) * np.cos(AZ), np.cos(EL) * np.sin(AZ), np.sin(EL)], -1).reshape(-1, 3)
    tfrac = (AZ / (2 * np.pi)).reshape(-1)
    R, o = pose[:3, :3], pose[:3, 3]
    d = d_s @ R.T
    n = d.shape[0]
    best_t = np.full(n, max_range)
    cls = np.full(n, -1, np.int32)
    nrm = np.zeros((n, 3))
    inten = np.zeros(n)

    def commit(mask, t, c, nv, base):
        upd = mask & (t < best_t) & (t > 0.5)
        best_t[upd] = t[upd]
        cls[upd] = c
        nrm[upd] = nv[upd] if nv.ndim == 2 else nv
        inten[upd] = base

    # ground z = 0
    with np.errstate(divide="ignore", invalid="ignore"):
        t = -o[2] / d[:, 2]
    commit(d[:, 2] < -1e-6, np.where(d[:, 2] < -1e-6, t, np.inf), abi.GROUND, np.array([0.0, 0.0, 1.0]), scene.ground_base)
    # facades
    for f in scene.facades:
        dn = d @ f["n"]
        with np.errstate(divide="ignore", invalid="ignore"):
            t = ((f["p0"] - o) @ f["n"]) / dn
        hit = o + d * t[:, None]
        u = (hit - f["p0"]) @ f["t"]
        ok = (np.abs(dn) > 1e-9) & (np.abs(u) < f["half_len"]) & (hit[:, 2] > 0.0) & (hit[:, 2] < f["height"])
        commit(ok, np.where(ok, t, np.inf), abi.FACADE, f["n"], f["base"])
    # roofs
    for r in scene.roofs:
        with np.errstate(divide="ignore", invalid="ignore"):
            t = (r["c"][2] - o[2]) / d[:, 2]
        hit = o + d * t[:, None]
        ok = (np.abs(d[:, 2]) > 1e-9) & (np.abs(hit[:, 0] - r["c"][0]) < r["hx"]) & (np.abs(hit[:, 1] - r["c"][1]) < r["hy"])
        commit(ok, np.where(ok, t, np.inf), abi.ROOF, np.array([0.0, 0.0, -1.0]), r["base"])
    # poles (vertical cylinders): quadratic in xy
    dxy2 = d[:, 0] ** 2 + d[:, 1] ** 2
    for p in scene.poles:
        oc = o[:2] - p["c"]
        b = d[:, 0] * oc[0] + d[:, 1] * oc[1]
        c = oc @ oc - p["r"] ** 2
        disc = b * b - dxy2 * c
        with np.errstate(divide="ignore", invalid="ignore"):
            t = (-b - np.sqrt(np.maximum(disc, 0.0))) / dxy2
        z = o[2] + d[:, 2] * t
        ok = (disc > 0) & (z > 0.0) & (z < p["h"])
        commit(ok, np.where(ok, t, np.inf), abi.PILLAR, np.array([0.0, 0.0, 1.0]), p["base"])
    # rails (horizontal cylinders along u)
    for r in scene.rails:
        u = r["u"]
        oc = o - r["c"]
        dp = d - np.outer(d @ u, u)
        ocp = oc - (oc @ u) * u
        a = np.einsum("ij,ij->i", dp, dp)
        b = dp @ ocp
        c = ocp @ ocp - r["r"] ** 2
        disc = b * b - a * c
        with np.errstate(divide="ignore", invalid="ignore"):
            t = (-b - np.sqrt(np.maximum(disc, 0.0))) / a
        s = (oc @ u) + (d @ u) * t
        ok = (disc > 0) & (np.abs(s) < r["half_len"])
        commit(ok, np.where(ok, t, np.inf), abi.BEAM, u, r["base"])

    keep = cls >= 0
    t_hit = best_t[keep] + rng.normal(0.0, range_noise, keep.sum())
    xyz = d_s[keep] * t_hit[:, None]
    nv = nrm[keep] @ R  # world -> sensor frame (R^T n)
    it = np.clip(inten[keep] + rng.normal(0.0, 5.0, keep.sum()), 0.0, 255.0)
    return dict(xyz=xyz.astype(np.float32), nrm=nv.astype(np.float32), cls=cls[keep], intensity=it.astype(np.float32),
                t=tfrac[keep].astype(np.float32))


def _pick(rng, idx, k):
    if k is None or len(idx) <= k:
        return np.sort(idx)
    return np.sort(rng.choice(idx, size=k, replace=False))


def class_clouds(scan, counts, seed=0, vertex_count=0):
    """Split a scan into the six class clouds, randomly thinned to `counts` (dict class->max points or None)."""
    rng = np.random.default_rng(seed)
    out = []
    for c in range(abi.NCLASS):
        if c == abi.VERTEX:
            idx = np.nonzero((scan["cls"] == abi.PILLAR) | (scan["cls"] == abi.BEAM))[0]
            idx = _pick(rng, idx, vertex_count) if vertex_count else idx[:0]
        else:
            idx = _pick(rng, np.nonzero(scan["cls"] == c)[0], counts.get(c))
        out.append(abi.make_points(scan["xyz"][idx], scan["nrm"][idx], scan["intensity"][idx], scan["t"][idx]))
    return out


# reference-default ("R") sizes: script/config/lo_gflag_list_kitti_urban.txt:39-42 fixed-number source down-sampling
# (ground 800 / pillar 400 / facade 1200 / beam 200) and un-down-sampled previous-frame features as the target.
R_SOURCE = {abi.GROUND: 800, abi.PILLAR: 400, abi.FACADE: 1200, abi.BEAM: 200, abi.ROOF: 100}
R_TARGET = {abi.GROUND: 5000, abi.PILLAR: 1500, abi.FACADE: 6000, abi.BEAM: 600, abi.ROOF: 400}


def make_pair(seed, n_beams=64, n_az=1900, elev_deg=(-24.8, 2.0), src_counts=None, tgt_counts=None, vertex_count=300,
              guess_noise=(0.3, 0.5), motion=None, scene=None):
    """One scan pair (target = scan A at the origin, source = scan B after a known motion T_gt).

    Returns (abi.PairData, T_gt) where T_gt maps source-frame points into the target frame and the pair's
    init_guess is T_gt perturbed by guess_noise = (metres, degrees).
    """
    rng = np.random.default_rng(seed)
    scene = scene or Scene(seed)
    src_counts = R_SOURCE if src_counts is None else src_counts
    tgt_counts = R_TARGET if tgt_counts is None else tgt_counts
    h = scene.sensor_height
    pose_a = se3(0, 0, h)
    if motion is None:
        motion = se3(rng.uniform(0.5, 1.5), rng.normal(0, 0.05), rng.normal(0, 0.02), np.deg2rad(rng.normal(0, 0.2)),
                     np.deg2rad(rng.normal(0, 0.2)), np.deg2rad(rng.normal(0, 1.0)))
    pose_b = pose_a @ motion
    scan_a = raycast(scene, pose_a, n_beams, n_az, elev_deg, seed=seed * 2 + 1)
    scan_b = raycast(scene, pose_b, n_beams, n_az, elev_deg, seed=seed * 2 + 2)
    tgt = class_clouds(scan_a, tgt_counts, seed=seed * 3 + 1, vertex_count=vertex_count)
    src = class_clouds(scan_b, src_counts, seed=seed * 3 + 2, vertex_count=vertex_count)
    T_gt = motion
    dm, dd = guess_noise
    pert = se3(*(rng.normal(0, dm / np.sqrt(3), 3)), *(np.deg2rad(rng.normal(0, dd / np.sqrt(3), 3))))
    guess = pert @ T_gt
    pair = abi.PairData(tgt, src, init_guess=guess)
    pair.n_raw = (len(scan_a["xyz"]), len(scan_b["xyz"]))
    return pair, T_gt


def drive(seed, n_frames, n_beams=32, n_az=700, elev_deg=(-24.8, 2.0), counts=None, vertex_count=150, step=0.8):
    """A short drive through ONE scene: per frame the six class clouds in the sensor frame and the ground-truth pose
    (frame -> world, world = frame 0).  Consecutive frames differ by about `step` metres and a degree of heading."""
    rng = np.random.default_rng(seed)
    scene = Scene(seed)
    counts = R_SOURCE if counts is None else counts
    world0 = se3(0, 0, scene.sensor_height)
    pose = np.eye(4)
    frames = []
    for k in range(n_frames):
        scan = raycast(scene, world0 @ pose, n_beams, n_az, elev_deg, seed=seed * 7 + k)
        frames.append((class_clouds(scan, counts, seed=seed * 11 + k, vertex_count=vertex_count), pose.copy()))
        pose = pose @ se3(rng.uniform(0.7, 1.3) * step, rng.normal(0, 0.03), rng.normal(0, 0.01), np.deg2rad(rng.normal(0, 0.1)),
                          np.deg2rad(rng.normal(0, 0.1)), np.deg2rad(rng.normal(0, 1.0)))
    return frames


def pose_error(T, T_ref):
    """(translation error in m, rotation geodesic in rad) between two 4x4 transforms."""
    dT = np.linalg.inv(T_ref) @ T
    c = (np.trace(dT[:3, :3]) - 1.0) / 2.0
    return float(np.linalg.norm(dT[:3, 3])), float(np.arccos(np.clip(c, -1.0, 1.0)))
