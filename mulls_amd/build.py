"""Build libmulls_hip.so (hand-written HIP kernels + host driver) for gfx950 with hipcc, in-tree.

    python -m mulls_amd.build            # incremental
    python -m mulls_amd.build --force

hipcc cross-compiles without a GPU.  The flags matter for parity: -ffp-contract=off keeps every float/double
expression free of FMA contraction (the reference is built -O3 without -march, CMakeLists.txt:43).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmulls_hip.so")
SOURCES = ["k_setup.hip", "k_grid.hip", "k_search.hip", "k_reduce.hip", "k_icp.hip", "k_ground.hip", "k_ground_normals.hip", "k_classify.hip", "map_kernels.hip", "driver.cpp", "batch.cpp", "loop.cpp", "variants.cpp", "stage.cpp", "shard.cpp", "map.cpp", "ground.cpp", "classify.cpp", "io.cpp"]
DEPS = ["device_types.h", "device_util.h", "lds_tier.h", "big_tier.h", "crop_grid.h", "solve_wave.h", "detmath.h", "accum.h", "icp_step.h", "launch.h", "map_launch.h", "classify_launch.h", "ground_launch.h", "pca_device.h", "ctx.h", "batch.h", "hostmath.h", os.path.join("..", "..", "include", "mulls_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fopenmp", "-Wall", "-Wno-unused-function"]


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for f in SOURCES + DEPS + [os.path.join("..", "build.py")]:
        if os.path.getmtime(os.path.join(CSRC, f)) > t:
            return True
    return False


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    objs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src + ".o")
        cmd = [hipcc()] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-fopenmp", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
