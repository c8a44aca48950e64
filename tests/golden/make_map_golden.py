"""Generate tests/golden/map_update_small.npz: a three-frame local-map sequence (update_local_map with map-based dynamic
removal, no thinning) written by the reference's own lines (oracle/_ref, src/map_manager.cpp:18-256) when that build
exists, else by the oracle; the `producer` field records which.

    python tests/golden/make_map_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from mulls_amd import abi, synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
FIELDS = abi.POINT_DTYPE.names
PATH = os.path.join(HERE, "map_update_small.npz")
PARAMS = dict(max_num_pts=3000, kept_vertex_num=100000, local_map_radius=50.0, map_based_dynamic_removal_on=1,
              dynamic_removal_center_radius=25.0, dynamic_dist_thre_min=0.25, dynamic_dist_thre_max=1.0, near_dist_thre=0.05, tree_mode=2,
              tree_used="111110", tree_box=[-30.0, -15.0, -3.0, 32.0, 15.0, 8.0])


def table(pts):
    return np.column_stack([pts[k] for k in FIELDS]).astype(np.float32)


def from_table(t):
    out = np.zeros(len(t), abi.POINT_DTYPE)
    for i, k in enumerate(FIELDS):
        out[k] = t[:, i]
    return out


def frames():
    counts = {abi.GROUND: 260, abi.PILLAR: 110, abi.FACADE: 300, abi.BEAM: 60, abi.ROOF: 30}
    return synth.drive(31, 3, n_beams=24, n_az=500, counts=counts, vertex_count=60)


def main():
    from oracle import pyoracle, pyref

    producer = "ref" if pyref.available() else "oracle"
    update = pyref.map_update if producer == "ref" else pyoracle.map_update
    fr = frames()
    P = abi.map_params(**PARAMS)
    out = dict(producer=np.array(producer), params=np.array(json.dumps(PARAMS)))
    clouds, pose = [c.copy() for c in fr[0][0]], fr[0][1]
    for k, (fc, fp) in enumerate(fr):
        out["pose%d" % k] = fp
        for c in range(6):
            out["frame%d_%d" % (k, c)] = table(fc[c])
    for k in range(1, len(fr)):
        clouds, appended, rep = update(clouds, pose, fr[k][0], fr[k][1], P)
        pose = fr[k][1]
        for c in range(6):
            out["map%d_%d" % (k, c)] = table(clouds[c])
            out["app%d_%d" % (k, c)] = table(appended[c])
        out["bounds%d" % k] = np.array(list(rep.local_bound) + list(rep.bound))
        out["removed%d" % k] = np.int32(sum(len(fr[k][0][c]) - rep.frame_n[c] for c in range(5)))
    np.savez_compressed(PATH, **out)
    print("wrote", PATH, "producer", producer, "removed", [int(out["removed%d" % k]) for k in range(1, len(fr))])


def load():
    z = np.load(PATH)
    n = len([k for k in z.files if k.startswith("pose")])
    fr = [([from_table(z["frame%d_%d" % (k, c)]) for c in range(6)], z["pose%d" % k]) for k in range(n)]
    return z, fr, abi.map_params(**json.loads(str(z["params"])))


if __name__ == "__main__":
    main()
