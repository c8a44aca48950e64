"""Committed golden fixtures (tests/golden/*.npz, produced by tests/golden/make_golden.py).

CPU: the oracle must reproduce every fixture exactly (guards the oracle against accidental edits).
GPU: the HIP path must reproduce every fixture: integer outputs identical, transform within 1e-7 (contract: 1e-4)."""
import importlib.util
import os

import numpy as np
import pytest

from mulls_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
mg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mg)


def check(r, z, exact):
    assert r.code == int(z["code"]) and r.iters == int(z["iters"])
    assert list(r.ncorr) == list(z["ncorr"]) and list(r.nsrc0) == list(z["nsrc0"]) and list(r.ntgt0) == list(z["ntgt0"])
    assert r.trace_len == len(z["trace_ncorr"])
    for k in range(r.trace_len):
        assert list(r.trace[k].ncorr) == list(z["trace_ncorr"][k]), k
        assert list(r.trace[k].nsrc) == list(z["trace_nsrc"][k]), k
    if exact:
        assert np.array_equal(r.T_matrix(), z["T"]) and np.array_equal(r.info_matrix(), z["info"]) and r.sigma == float(z["sigma"])
    else:
        dt, dr = synth.pose_error(r.T_matrix(), z["T"])
        assert dt <= 1e-7 and dr <= 1e-7
        assert np.abs(r.info_matrix() - z["info"]).max() <= 1e-6 * np.abs(z["info"]).max()
        assert abs(r.sigma - float(z["sigma"])) <= 1e-6
    assert r.confidence == float(z["confidence"]) or (np.isnan(r.confidence) and np.isnan(float(z["confidence"])))


@pytest.mark.parametrize("name", sorted(mg.CASES))
def test_oracle_reproduces_golden(name):
    from oracle import pyoracle

    pair, P, z = mg.load(name)
    r = pyoracle.icp(pair, P, trace_cap=64)[0]
    check(r, z, exact=str(z["producer"]) == "oracle")


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(mg.CASES))
def test_hip_reproduces_golden(ctx, name):
    pair, P, z = mg.load(name)
    r = ctx.icp(pair, P, trace_cap=64)[0]
    check(r, z, exact=False)
