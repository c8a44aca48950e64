"""The code that is compiled out of the default build must keep compiling: the LDS tier's k-candidate records and look (`-DMULLS_LDS_KCERT=1`, lds_tier.h;
profiles/r05_experiments.txt item 6 says why the default leaves them out).  hipcc cross-compiles gfx950 without a GPU; the bit-identity of such a build is a GPU check
(`MULLS_HIP_LIB=tools/_bin/libmulls_ldskc.so pytest tests/test_gpu_icp.py`, tools/build_variant.sh)."""
import os
import shutil
import subprocess

import pytest

from mulls_amd import build

CSRC = os.path.join(os.path.dirname(os.path.abspath(build.__file__)), "csrc")


@pytest.mark.parametrize("flags", [["-DMULLS_LDS_KCERT=1"]])
def test_variant_translation_unit_compiles(tmp_path, flags):
    hipcc = build.hipcc()
    if not (os.path.isabs(hipcc) and os.path.exists(hipcc)) and shutil.which(hipcc) is None:
        pytest.skip("no hipcc here")
    obj = tmp_path / "k_search.variant.o"
    cmd = [hipcc] + build.FLAGS + ["-Werror"] + flags + ["-c", os.path.join(CSRC, "k_search.hip"), "-o", str(obj)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    assert obj.stat().st_size > 100000
