"""Build librsqp_hip.so in-tree with hipcc for gfx950 (no GPU needed to compile).

One object per translation unit (device code never crosses a TU), compiled in parallel and only
when the source or a header is newer than the object; then one link step."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB = os.path.join(LIB_DIR, "librsqp_hip.so")
SOURCES = ["rsqp_api.hip", "qp_small.hip", "qp_large.hip", "sparse.hip", "dense_la.hip", "qp_dump.cpp"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wno-unused-result"]


def _headers():
    """every header of csrc/ (incl. qp_small_x.h, which qp_small.hip includes) + the public C ABI"""
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")] + \
           [os.path.join(_HERE, "..", "include", "rsqp_hip.h")]


def _obj(src):
    return os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".o")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build():
    hdr = _headers()
    return _stale(LIB, [os.path.join(CSRC, f) for f in SOURCES] + hdr)


def build_lib(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdr = _headers()
    todo = [f for f in SOURCES if force or _stale(_obj(f), [os.path.join(CSRC, f)] + hdr)]

    def compile_one(f):
        cmd = [hipcc] + FLAGS + (["-x", "hip"] if f.endswith(".hip") else []) + ["-c", os.path.join(CSRC, f), "-o", _obj(f)]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(todo)))) as ex:
        list(ex.map(compile_one, todo))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + [_obj(f) for f in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    import sys
    print(build_lib(force="--force" in sys.argv, verbose=True))
