"""Golden fixture of the frame front end's pre-stages (tests/test_classify.py::test_front_end_golden_fixture): CFilter::dist_filter
(cfilter.hpp:806-832) and CFilter::voxel_downsample (:83-160) run BY THE REFERENCE'S OWN LINES (oracle/_ref, pyref) on the golden scan (every 4th
point of the reference's demo_data/pcd/000000.pcd, tests/golden/ground_filter_demo.npz): sizes and checksums of the clouds they return.  The
voxel grid's choice of point per voxel is std::sort's (the comparison sees the voxel only), so these checksums pin that choice as libstdc++
makes it.  Run where /root/reference exists:  python tests/golden/make_front_end_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from mulls_amd import abi  # noqa: E402
from oracle import pyoracle, pyref  # noqa: E402

DIST = ((1.5, 120.0), (5.0, 40.0))
VOXELS = (0.05, 0.2, 0.5, 2.0)


def checksum(raw):
    return int(np.frombuffer(raw.tobytes(), np.uint32).astype(np.uint64).sum() & 0xffffffff)


def main():
    scan = np.load(os.path.join(HERE, "ground_filter_demo.npz"))["scan"].view(abi.POINT_DTYPE).reshape(-1)
    rows = []
    for lo, hi in DIST:
        r = pyref.dist_filter(scan, lo, hi)
        assert np.array_equal(r, pyoracle.dist_filter(scan, lo, hi)), "oracle != reference lines"
        rows.append((len(r), checksum(r)))
    for v in VOXELS:
        r = pyref.voxel_downsample(scan, v)
        assert np.array_equal(r, pyoracle.voxel_downsample(scan, v)), "oracle != reference lines"
        rows.append((len(r), checksum(r)))
    path = os.path.join(HERE, "front_end_demo.npz")
    np.savez_compressed(path, dist=np.array(DIST, np.float64), voxels=np.array(VOXELS, np.float32), sizes=np.array([r[0] for r in rows], np.int64),
                        checksums=np.array([r[1] for r in rows], np.int64))
    print(len(scan), "points ->", rows, os.path.getsize(path), "B")


if __name__ == "__main__":
    main()
