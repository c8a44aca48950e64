"""CPU tests of the drop-in boundary: the C-ABI library loads here (no GPU) and exports every symbol that
include/mulls_hip.h declares; the ctypes mirror has the same struct layout as the C header; without a device the
product fails loudly instead of falling back to anything."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

from mulls_amd import abi, build, lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mulls_hip.h")


@pytest.fixture(scope="module")
def so():
    return build.build()


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mulls_[a-z0-9_]+)\s*\(", text)))


def test_header_is_plain_c():
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write('#include "mulls_hip.h"\nint main(void){return (int)sizeof(mulls_pair) == 0;}\n')
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", src, "-o", os.path.join(d, "t.o")])


def test_library_exports_every_declared_symbol(so):
    handle = C.CDLL(so)
    names = declared_functions()
    assert len(names) >= 15
    for name in names:
        assert hasattr(handle, name), "libmulls_hip.so does not export " + name
    assert sorted(lib.EXPORTS) == names


def test_ctypes_layout_matches_header():
    fields = {
        "mulls_cloud": (abi.Cloud, ["pts", "n", "stride"]),
        "mulls_pair": (abi.Pair, ["tgt", "src", "src_down", "tgt_bound", "init_guess"]),
        "mulls_params": (abi.Params, [f[0] for f in abi.Params._fields_]),
        "mulls_iter_trace": (abi.IterTrace, [f[0] for f in abi.IterTrace._fields_]),
        "mulls_result": (abi.Result, [f[0] for f in abi.Result._fields_]),
        "mulls_profile": (abi.Profile, [f[0] for f in abi.Profile._fields_]),
        "mulls_map_params": (abi.MapParams, [f[0] for f in abi.MapParams._fields_]),
        "mulls_map_report": (abi.MapReport, [f[0] for f in abi.MapReport._fields_]),
    }
    prog = ['#include <stdio.h>', '#include <stddef.h>', '#include "mulls_hip.h"', "int main(void){"]
    for cname, (_, names) in fields.items():
        prog.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for f in names:
            prog.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, f, cname, f))
    prog.append("return 0;}")
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "t.c"), os.path.join(d, "t")
        open(src, "w").write("\n".join(prog))
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        out = subprocess.check_output([exe]).decode().split("\n")
    got = dict(line.split() for line in out if line)
    for cname, (ct, names) in fields.items():
        assert int(got[cname]) == C.sizeof(ct), cname
        for f in names:
            assert int(got["%s.%s" % (cname, f)]) == getattr(ct, f).offset, (cname, f)
    assert abi.POINT_DTYPE.itemsize == 48


def test_default_params_match_reference_defaults(so):
    p = abi.Params()
    lib.load().mulls_default_params(C.byref(p))
    q = abi.default_params()
    for name, _ in abi.Params._fields_:
        if name == "reserved_":
            continue
        assert getattr(p, name) == getattr(q, name), name
    assert p.max_iter_num == 20 and p.used_feature_type == b"111110" and p.weight_strategy == b"1101"


def test_no_device_fails_loudly(so):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    with pytest.raises(lib.MullsError):
        lib.Context(0)


def test_product_does_not_reference_the_oracle():
    """The package under mulls_amd/ must never import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "mulls_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pyoracle" not in text and "liboracle" not in text and "mulls_oracle_" not in text, f
    out = subprocess.check_output(["ldd", os.path.join(pkg, "libmulls_hip.so")]).decode()
    assert "oracle" not in out
