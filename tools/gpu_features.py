"""Feature extraction on the device vs the CPU oracle, one 64-beam scan: fast_ground_filter + classify_nground_pts.  usage gpu_features.py"""
import sys, time, warnings
sys.path.insert(0, "."); warnings.filterwarnings("ignore")
import numpy as np
from mulls_amd import abi, lib, synth
from oracle import pyoracle
scene = synth.Scene(3)
s = synth.raycast(scene, synth.se3(0, 0, scene.sensor_height), 64, 1900, seed=3)
pts = abi.make_points(s["xyz"], np.zeros_like(s["xyz"]), s["intensity"], s["t"])
ctx = lib.Context(0)
GP = abi.ground_params()
for name, CP, rate in (("run_mulls_reg.sh (r 1.0, k 50)", abi.classify_params(), 3), ("kitti flags (r 0.7, k 25)", abi.classify_params(neighbor_searching_radius=0.7, neighbor_k=25, neigh_k_min=7, curvature_thre=0.08), 3),
                       ("every unground point (rate 1)", abi.classify_params(), 1)):
    GPr = abi.ground_params(nonground_random_down_rate=rate)
    ung = ctx.ground_filter(pts, GPr)[2]
    a = ctx.classify_nground(ung, CP)
    t = time.time()
    for _ in range(10):
        a = ctx.classify_nground(ung, CP)
    dt = (time.time() - t) / 10
    t = time.time()
    g = ctx.ground_filter(pts, GPr)
    a = ctx.classify_nground(g[2], CP)
    dboth = time.time() - t
    t = time.time()
    b, _ = pyoracle.classify_nground(ung, CP)
    do = time.time() - t
    ok = all(np.array_equal(x, y) for x, y in zip(a, b))
    print("%-32s %d unground points -> %s: device %.2f ms (upload + kernels + host sort + download), scan -> features %.2f ms, oracle %.1f ms (one core), identical %s"
          % (name, len(ung), [len(x) for x in a], dt * 1e3, dboth * 1e3, do * 1e3, ok))
X = abi.extract_params(ground=abi.ground_params(), classify=abi.classify_params())
a = ctx.extract_features(pts, X)
t = time.time()
for _ in range(10):
    a = ctx.extract_features(pts, X)
dx = (time.time() - t) / 10
t = time.time()
for _ in range(10):
    g = ctx.ground_filter(pts, abi.ground_params())
    c = ctx.classify_nground(g[2], abi.classify_params())
d2 = (time.time() - t) / 10
print("mulls_extract_features (one call, clouds stay on the device): %.2f ms; the two calls: %.2f ms; %d points -> %s" % (dx * 1e3, d2 * 1e3, len(pts), [len(x) for x in a]))
