"""Ground filter (mulls_ground_filter) on the device vs the CPU oracle, one 64-beam scan: usage gpu_ground.py"""
import sys, time, warnings
sys.path.insert(0, "."); warnings.filterwarnings("ignore")
import numpy as np
from mulls_amd import abi, lib, synth
from oracle import pyoracle
scene = synth.Scene(3)
s = synth.raycast(scene, synth.se3(0, 0, scene.sensor_height), 64, 1900, seed=3)
pts = abi.make_points(s["xyz"], np.zeros_like(s["xyz"]), s["intensity"], s["t"])
ctx = lib.Context(0)
for name, P in (("kitti (dist 0)", abi.ground_params()), ("dist 2", abi.ground_params(distance_weight_downsampling_method=2)), ("outlier filter", abi.ground_params(apply_grid_wise_outlier_filter=1)),
                ("normals: RANSAC (3)", abi.ground_params(estimate_ground_normal_method=3, distance_weight_downsampling_method=2)),
                ("normals: radius (1)", abi.ground_params(estimate_ground_normal_method=1)), ("normals: k-NN (2)", abi.ground_params(estimate_ground_normal_method=2))):
    a = ctx.ground_filter(pts, P)
    t = time.time()
    for _ in range(10):
        a = ctx.ground_filter(pts, P)
    dt = (time.time() - t) / 10
    t = time.time()
    for _ in range(3):
        b = pyoracle.ground_filter(pts, P)
    do = (time.time() - t) / 3
    ok = all(np.array_equal(x, y) for x, y in zip(a, b))
    print("%-20s %d points -> ground %d / %d, unground %d: device %.2f ms (upload + kernel + download), oracle %.2f ms, identical %s"
          % (name, len(pts), len(a[0]), len(a[1]), len(a[2]), dt * 1e3, do * 1e3, ok))
