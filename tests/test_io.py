"""On-disk formats: KITTI .bin, PCD v0.7 of PointXYZINormal, odometry pose lines — host code of the
library, checked against byte-level expectations derived from the reference's readers/writers (dataio.hpp:279-312,
:357-378, :1896-1926) and, where /root/reference is mounted, against its demo_data files."""
import os
import struct

import numpy as np
import pytest

from mulls_amd import abi, lib

DEMO = "/root/reference/demo_data/pcd/000000.pcd"


def random_points(n, seed=0):
    rng = np.random.default_rng(seed)
    p = np.zeros(n, abi.POINT_DTYPE)
    for k in abi.POINT_DTYPE.names:
        p[k] = rng.normal(0, 10, n).astype(np.float32)
    return p


def test_kitti_bin_reader(tmp_path):
    rng = np.random.default_rng(1)
    raw = rng.normal(0, 20, (1000, 4)).astype(np.float32)
    raw[:, 3] = rng.random(1000).astype(np.float32)
    path = str(tmp_path / "000000.bin")
    raw.tofile(path)
    pts = lib.read_kitti_bin(path)
    # the reference's loop appends one default-constructed point after the last record (dataio.hpp:368-375)
    assert len(pts) == 1001
    assert np.array_equal(pts["x"][:1000], raw[:, 0]) and np.array_equal(pts["y"][:1000], raw[:, 1]) and np.array_equal(pts["z"][:1000], raw[:, 2])
    assert np.array_equal(pts["intensity"][:1000], raw[:, 3] * np.float32(255))
    for k in abi.POINT_DTYPE.names:
        assert pts[k][1000] == 0
    assert not pts["nx"].any() and not pts["curvature"].any()
    with pytest.raises(lib.MullsError):
        lib.read_kitti_bin(str(tmp_path / "missing.bin"))


@pytest.mark.parametrize("binary", [True, False])
def test_pcd_round_trip(tmp_path, binary):
    pts = random_points(257, 2)
    pts["curvature"][3] = np.nan
    path = str(tmp_path / "c.pcd")
    lib.write_pcd(path, pts, binary=binary)
    head = open(path, "rb").read(400).decode("latin1")
    assert head.startswith("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity normal_x normal_y normal_z curvature\n")
    assert "WIDTH 1\nHEIGHT 257\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS 257\nDATA %s\n" % ("binary" if binary else "ascii") in head
    back = lib.read_pcd(path)
    assert len(back) == 257
    for k in abi.POINT_DTYPE.names:
        if binary:
            assert np.array_equal(back[k], pts[k], equal_nan=True), k
        else:  # eight significant digits are not always enough to round-trip a float32: 1 ulp
            assert np.allclose(back[k], pts[k], rtol=1e-7, atol=0, equal_nan=True), k
    if binary:
        size = os.path.getsize(path)
        assert size == head.index("DATA binary\n") + len("DATA binary\n") + 257 * 32  # eight float32 per point on disk


def test_pcd_reader_maps_fields_by_name(tmp_path):
    """A file with another field order, an extra field and a missing one: by-name mapping, absent fields stay 0."""
    n = 5
    path = str(tmp_path / "odd.pcd")
    rows = np.arange(n * 5, dtype=np.float32).reshape(n, 5)
    with open(path, "wb") as f:
        f.write(("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS intensity z ring y x\nSIZE 4 4 4 4 4\nTYPE F F F F F\n"
                 "COUNT 1 1 1 1 1\nWIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA binary\n" % (n, n)).encode())
        f.write(rows.tobytes())
    p = lib.read_pcd(path)
    assert np.array_equal(p["intensity"], rows[:, 0]) and np.array_equal(p["z"], rows[:, 1]) and np.array_equal(p["y"], rows[:, 3])
    assert np.array_equal(p["x"], rows[:, 4]) and not p["nx"].any() and not p["curvature"].any()


def test_pose_lines(tmp_path):
    T = np.array([[0.99999987, -1.234567891e-4, 0.5, 12.3456789012], [1e-9, 1.0, -0.25, -1000.5], [3.0, 2.0, 1.0, 1e-12], [0, 0, 0, 1]])
    path = str(tmp_path / "pose.txt")
    lib.write_pose(path, T, append=False)
    lib.write_pose(path, np.eye(4), append=True)
    lines = open(path).read().split("\n")
    assert lines[0] == "0.99999987 -0.00012345679 0.5 12.345679 1e-09 1 -0.25 -1000.5 3 2 1 1e-12"  # std::setprecision(8), default notation
    assert lines[1] == "1 0 0 0 0 1 0 0 0 0 1 0" and lines[2] == ""
    lib.write_pose(path, np.eye(4), append=False)  # overwrite
    assert open(path).read().count("\n") == 1


@pytest.mark.skipif(not os.path.exists(DEMO), reason="reference demo data not mounted")
def test_reference_demo_pcd():
    """demo_data/pcd/000000.pcd (binary, 124 668 points, 32 B per point on disk)."""
    p = lib.read_pcd(DEMO)
    assert len(p) == 124668
    raw = open(DEMO, "rb").read()
    body = raw[raw.index(b"DATA binary\n") + len(b"DATA binary\n"):]
    first = struct.unpack("<8f", body[:32])
    assert (p["x"][0], p["y"][0], p["z"][0], p["intensity"][0], p["nx"][0], p["ny"][0], p["nz"][0]) == first[:7]
    rng = np.sqrt(p["x"].astype(np.float64) ** 2 + p["y"] ** 2 + p["z"] ** 2)
    assert 9.0 < np.median(rng) < 11.5 and p["intensity"].max() <= 255  # median range 10.1 m, intensity 0-252


def _pcd(tmp_path, header, payload=b""):
    p = tmp_path / "bad.pcd"
    p.write_bytes(header.encode() + payload)
    return str(p)


def test_pcd_reader_rejects_hostile_headers(tmp_path):
    """The header is untrusted: negative sizes / counts, point counts the file cannot hold and overflowing products are
    refused with an error code — no out-of-bounds read, no exception across the ABI."""
    import ctypes as C

    from mulls_amd import lib

    L = lib.load()
    n = C.c_uint32(0)
    base = "# .PCD v0.7\nVERSION 0.7\nFIELDS x y z intensity\n%s\nTYPE F F F F\n%s\nWIDTH %s\nHEIGHT 1\nPOINTS %s\nDATA binary\n"
    cases = [
        base % ("SIZE 4 -4 4 4", "COUNT 1 1 1 1", "4", "4"),            # step would wrap to 8 -> reads past the blob
        base % ("SIZE 4 4 4 3", "COUNT 1 1 1 1", "4", "4"),             # not a PCD field size
        base % ("SIZE 4 4 4 4", "COUNT 1 0 1 1", "4", "4"),             # zero count
        base % ("SIZE 4 4 4 4", "COUNT 1 -1 1 1", "4", "4"),            # negative count
        base % ("SIZE 4 4 4 4", "COUNT 1 1 1 1", "4", "4000000000000"),  # more points than the file (or memory) can hold
        base % ("SIZE 4 4 4 4", "COUNT 1 1 1 1", "4", "18446744073709551615"),  # points * step overflows size_t
    ]
    for hdr in cases:
        path = _pcd(tmp_path, hdr, b"\0" * 64)
        rc = L.mulls_io_read_pcd(path.encode(), None, 0, C.byref(n))
        assert rc in (abi.MULLS_E_IO, abi.MULLS_E_NOMEM), (rc, hdr)
    # WIDTH * HEIGHT overflow without a POINTS line
    hdr = "VERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 4294967296\nHEIGHT 4294967296\nDATA ascii\n"
    assert L.mulls_io_read_pcd(_pcd(tmp_path, hdr).encode(), None, 0, C.byref(n)) == abi.MULLS_E_IO
    # a well-formed file of the same shape still reads
    good = base % ("SIZE 4 4 4 4", "COUNT 1 1 1 1", "2", "2")
    path = _pcd(tmp_path, good, np.arange(8, dtype=np.float32).tobytes())
    pts = lib.read_pcd(path)
    assert len(pts) == 2 and pts["x"][1] == 4.0 and pts["intensity"][1] == 7.0
