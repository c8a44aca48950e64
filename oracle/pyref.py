"""ctypes binding of oracle/_ref/libmulls_ref.so — the reference's own MULLS-ICP function bodies compiled against the
stand-in headers of oracle/ref_shim (TEST INFRASTRUCTURE ONLY; built by oracle/build_ref.sh where /root/reference exists).
"""
import ctypes as C
import os

from mulls_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libmulls_ref.so")
_LIB = None


def available():
    return os.path.exists(LIB_PATH)


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(LIB_PATH)
        _LIB.mulls_ref_icp.argtypes = [C.POINTER(abi.Pair), C.POINTER(abi.Params), C.POINTER(abi.Result)]
    return _LIB


def icp(pair, params, trace_cap=0):
    res = abi.make_result_array(1, 0)
    p = pair.as_pair()
    rc = lib().mulls_ref_icp(C.byref(p), C.byref(params), C.byref(res[0]))
    if rc != 0:
        raise RuntimeError("reference returned %d" % rc)
    return res
