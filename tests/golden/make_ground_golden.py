"""Golden fixture of the ground filter (tests/test_ground_filter.py::test_golden_fixture): every 4th point of the reference's
demo scan demo_data/pcd/000000.pcd (x, y, z, intensity as float32) with the sizes and checksums of the three clouds the oracle
returns for it — the oracle that test_oracle_equals_reference_lines ties byte for byte to the reference's own lines.
Run where /root/reference exists:  python tests/golden/make_ground_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from mulls_amd import abi, lib  # noqa: E402
from oracle import pyoracle, pyref  # noqa: E402


def main():
    pts = lib.read_pcd("/root/reference/demo_data/pcd/000000.pcd")[::4].copy()
    scan = np.zeros(len(pts), abi.POINT_DTYPE)
    for k in ("x", "y", "z", "intensity"):
        scan[k] = pts[k]
    P = abi.ground_params()
    a = pyoracle.ground_filter(scan, P)
    b = pyref.ground_filter(scan, P)
    assert all(np.array_equal(x, y) for x, y in zip(a, b)), "oracle != reference lines"
    sizes = np.array([len(x) for x in a], np.int64)
    sums = np.array([int(np.frombuffer(x.tobytes(), np.uint32).astype(np.uint64).sum() & 0xffffffff) for x in a], np.int64)
    path = os.path.join(HERE, "ground_filter_demo.npz")
    np.savez_compressed(path, scan=np.frombuffer(scan.tobytes(), np.uint8), sizes=sizes, checksums=sums)
    print(len(scan), "points ->", sizes, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
