"""Build librsqp_hip.so in-tree with hipcc for gfx950 (no GPU needed to compile)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB = os.path.join(LIB_DIR, "librsqp_hip.so")
SOURCES = ["rsqp_api.hip", "qp_small.hip", "qp_large.hip", "sparse.hip", "dense_la.hip"]
HEADERS = ["rsqp_internal.h", "rsqp_sparse.h", "rsqp_large.h", "rsqp_dense.h", os.path.join("..", "..", "include", "rsqp_hip.h")]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build_lib(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result",
           "-o", LIB] + [os.path.join(CSRC, f) for f in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
