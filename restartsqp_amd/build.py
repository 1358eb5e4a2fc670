"""Build librsqp_hip.so in-tree with hipcc for gfx950 (no GPU needed to compile).

One object per translation unit (device code never crosses a TU), compiled in parallel; then one link step.
What is stale is decided by CONTENT, not by mtime: the sha256 of every source, header and flag is stamped into the
library (`rsqp_build_hash()`, from build_stamp.cpp) and kept beside every object (`obj/<name>.hash`). A prebuilt `.so`
that travelled to another box with an older or newer tree is therefore recognised and rebuilt; `build_lib()` returns the
library path and `LAST_ACTION` says what it did ("up to date" / "rebuilt: <objects>")."""
import hashlib
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB = os.path.join(LIB_DIR, "librsqp_hip.so")
SOURCES = ["rsqp_api.hip", "qp_small.hip", "qp_tiny.hip", "qp_lane.hip", "qp_large.hip", "sparse.hip", "dense_la.hip", "qp_dump.cpp", "rsqp_rccl.cpp"]
STAMP = "build_stamp.cpp"
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wno-unused-result"]
LINK_LIBS = []          # librccl is bound at run time (dlopen in rsqp_rccl.cpp): a single-GPU host needs no RCCL
LAST_ACTION = None


def _headers():
    """every header of csrc/ (incl. qp_small_x.h, which qp_small.hip includes) + the public C ABI"""
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")] + \
           [os.path.join(_HERE, "..", "include", "rsqp_hip.h")]


def _obj(src):
    return os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".o")


def _sha(paths, extra=""):
    h = hashlib.sha256(extra.encode())
    for p in paths:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _sources():
    return [f for f in SOURCES if os.path.exists(os.path.join(CSRC, f))]


def source_hash():
    """content hash of everything the library is built from"""
    return _sha([os.path.join(CSRC, f) for f in _sources() + [STAMP]] + _headers(), " ".join(FLAGS + LINK_LIBS))


def _includes(path, seen=None):
    """the quoted includes of a source, transitively (paths relative to the including file)"""
    import re
    seen = seen if seen is not None else []
    try:
        txt = open(path).read()
    except OSError:
        return seen
    for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', txt, flags=re.M):
        q = os.path.normpath(os.path.join(os.path.dirname(path), inc))
        if q not in seen and os.path.exists(q):
            seen.append(q)
            _includes(q, seen)
    return seen


def _tu_hash(f):
    """content hash of one translation unit: the source, every header it includes (transitively), the flags"""
    src = os.path.join(CSRC, f)
    return _sha([src] + sorted(_includes(src)), " ".join(FLAGS))


def stamped_hash(path=None):
    """the hash stamped into a built library, or None (no library / a library from before the stamp existed). Read from the
    file's bytes: dlopen-ing a library that is about to be replaced would pin the OLD image under that path for this process"""
    path = path or LIB
    if not os.path.exists(path):
        return None
    with open(path, "rb") as f:
        data = f.read()
    k = data.find(b"RSQP_SRC_HASH=")
    if k < 0:
        return None
    h = data[k + 14:k + 14 + 64]
    try:
        h = h.decode("ascii")
    except UnicodeDecodeError:
        return None
    return h if len(h) == 64 and all(c in "0123456789abcdef" for c in h) else None


def needs_build():
    return stamped_hash() != source_hash()


def build_lib(force=False, verbose=False):
    global LAST_ACTION
    want = source_hash()
    if not force and stamped_hash() == want:
        LAST_ACTION = "up to date (content hash %s)" % want[:12]
        return LIB
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = _sources()

    def stale(f):
        hp = _obj(f)[:-2] + ".hash"
        return force or not os.path.exists(_obj(f)) or not os.path.exists(hp) or open(hp).read().strip() != _tu_hash(f)

    todo = [f for f in srcs if stale(f)]

    def compile_one(f):
        cmd = [hipcc] + FLAGS + (["-x", "hip"] if f.endswith(".hip") else []) + ["-c", os.path.join(CSRC, f), "-o", _obj(f)]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        with open(_obj(f)[:-2] + ".hash", "w") as fh:
            fh.write(_tu_hash(f))

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(todo)))) as ex:
        list(ex.map(compile_one, todo))
    # the stamp: one tiny TU that carries the content hash of the whole tree
    stamp_o = _obj(STAMP)
    subprocess.check_call([hipcc, "-O1", "-fPIC", "-std=c++17", '-DRSQP_SRC_HASH="%s"' % want, "-c", os.path.join(CSRC, STAMP), "-o", stamp_o])
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + [_obj(f) for f in srcs] + [stamp_o] + LINK_LIBS
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    LAST_ACTION = "rebuilt: %s (content hash %s)" % (", ".join(todo) if todo else "link only", want[:12])
    return LIB


if __name__ == "__main__":
    import sys
    print(build_lib(force="--force" in sys.argv, verbose=True))
    print(LAST_ACTION)
