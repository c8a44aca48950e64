"""Golden fixture of classify_nground_pts (tests/test_classify.py::test_golden_fixture): the non-ground cloud the ground filter leaves of the
golden scan (every 4th point of the reference's demo_data/pcd/000000.pcd, tests/golden/ground_filter_demo.npz) with the sizes and checksums
of the nine clouds the oracle returns for it — the oracle that test_oracle_equals_reference_lines ties byte for byte to the reference's own
lines (checked again here).  Run where /root/reference exists:  python tests/golden/make_classify_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from mulls_amd import abi  # noqa: E402
from oracle import pyoracle, pyref  # noqa: E402

GROUND = abi.ground_params(nonground_random_down_rate=1)
CLASSIFY = abi.classify_params(neighbor_searching_radius=1.5, neighbor_k=30)  # a quarter of the scan's points: a wider neighbourhood


def checksum(raw):
    return int(np.frombuffer(raw.tobytes(), np.uint32).astype(np.uint64).sum() & 0xffffffff)


def main():
    scan = np.load(os.path.join(HERE, "ground_filter_demo.npz"))["scan"].view(abi.POINT_DTYPE).reshape(-1)
    ung = pyoracle.ground_filter(scan, GROUND)[2]
    a, a_in = pyoracle.classify_nground(ung, CLASSIFY)
    b, b_in = pyref.classify_nground(ung, CLASSIFY)
    assert all(np.array_equal(x, y) for x, y in zip(a, b)) and np.array_equal(a_in, b_in), "oracle != reference lines"
    sizes = np.array([len(x) for x in a] + [len(a_in)], np.int64)
    sums = np.array([checksum(x) for x in a] + [checksum(a_in)], np.int64)
    path = os.path.join(HERE, "classify_demo.npz")
    np.savez_compressed(path, sizes=sizes, checksums=sums, unground_size=len(ung), unground_checksum=checksum(ung))
    print(len(scan), "points ->", len(ung), "non-ground ->", sizes, os.path.getsize(path), "B")


if __name__ == "__main__":
    main()
