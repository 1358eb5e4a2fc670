"""CPU-side checks: the C-ABI library builds, loads and exports every symbol the header
declares (no compute without a GPU), fails loudly without a device, and the host-side
helpers (dump reader/writer, generators, image sizing) behave."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT, dump_paths
from restartsqp_amd import problems
from restartsqp_amd.qpdump import read_qore_dump, write_qore_dump


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "rsqp_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rsqp_[A-Za-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(capi):
    L = capi.lib()
    names = header_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(L, n), "librsqp_hip.so does not export %s" % n
    assert set(names) == set(capi.SYMBOLS), set(names) ^ set(capi.SYMBOLS)
    assert b"gfx950" in L.rsqp_version()


def test_no_silent_cpu_fallback(capi):
    """Without a GPU every entry point that computes must fail loudly."""
    if capi.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(capi.RsqpError) as e:
        capi.Solver(4, 2)
    assert e.value.code == capi.ERR_DEVICE
    with pytest.raises(capi.RsqpError):
        capi.Batch([problems.hs071_first_qp()])
    jc, ir, _ = problems.sparse_pattern(30, 50, 100, seed=1)
    with pytest.raises(capi.RsqpError):
        capi.SpmvPlan(50, 30, jc, ir, 1)


def test_product_does_not_import_oracle():
    import subprocess, sys
    code = ("import sys; sys.path.insert(0, %r); import restartsqp_amd.capi, restartsqp_amd.interface, "
            "restartsqp_amd.handler, restartsqp_amd.problems, restartsqp_amd.parallel; "
            "assert 'oracle' not in sys.modules" % ROOT)
    subprocess.check_call([sys.executable, "-c", code])
    for dirpath, _, files in os.walk(os.path.join(ROOT, "restartsqp_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                assert "oracle" not in open(os.path.join(dirpath, f)).read().lower().replace("no cpu fallback", ""), f


def test_qore_dump_round_trip(tmp_path):
    for p in dump_paths():
        q = read_qore_dump(p)
        out = tmp_path / os.path.basename(p)
        write_qore_dump(str(out), q)
        assert open(p).read().split() == open(out).read().split() or _numerically_equal(p, str(out))
        q2 = read_qore_dump(str(out))
        for name in ("H_jc", "H_ir", "H_val", "A_jc", "A_ir", "A_val", "g", "lb", "ub", "lbA", "ubA"):
            assert np.array_equal(getattr(q, name), getattr(q2, name))


def _numerically_equal(a, b):
    ta, tb = open(a).read().split(), open(b).read().split()
    return len(ta) == len(tb) and all(float(x) == float(y) for x, y in zip(ta, tb))


def test_dump_headers_match_survey():
    shapes = [(read_qore_dump(p).nV, read_qore_dump(p).nC) for p in dump_paths()]
    assert shapes == problems.HS_DUMP_SHAPES


def test_generators_are_deterministic():
    a, b = problems.hs_batch(40), problems.hs_batch(40)
    assert all(np.array_equal(p.g, q.g) and np.array_equal(p.A_val, q.A_val) for p, q in zip(a, b))
    jc, ir, _ = problems.sparse_pattern(100, 200, 500, seed=3)
    assert jc[-1] == 500 and len(set(zip(np.repeat(np.arange(100), np.diff(jc)).tolist(), ir.tolist()))) == 500
    q = problems.hs071_first_qp()
    assert (q.nV, q.nC, len(q.A_val), len(q.H_val)) == (8, 2, 12, 11)   # triplet path: 16 CSC slots incl. 5 explicit zeros


def test_sparse_config_shape():
    q = problems.sparse_qp(n=1000, m=2000, nnz=20000, seed=5)
    assert q.A_jc[-1] == 20000 and q.nV == 1000 and q.nC == 2000
    seq = list(problems.sparse_sequence(q, nsteps=4))
    assert [c for _, c in seq] == [False, True, False, True]
    assert not np.array_equal(seq[1][0].A_val, seq[0][0].A_val) and np.array_equal(seq[0][0].A_val, q.A_val)


def test_mirrors_declare_every_pure_virtual_of_the_reference_interface():
    """The plug-in contract: every `= 0` method of the reference's QPSolverInterface
    (include/sqphot/QPsolverInterface.hpp:43-194; fixture made by tests/golden/make_interface_fixture.py)
    is declared pure in the C++ mirror with the same arity and constness, overridden by HipQPInterface
    (so the subclass is concrete against the real header), and present in the Python mirror."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("mk", os.path.join(ROOT, "tests", "golden", "make_interface_fixture.py"))
    mk = importlib.util.module_from_spec(spec); spec.loader.exec_module(mk)
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "qpsolver_interface_pure_virtuals.json")))["pure_virtuals"]
    assert len(want) == 31
    hpp = open(os.path.join(ROOT, "restartsqp_amd", "csrc", "host", "HipQPInterface.hpp")).read()
    base = hpp[hpp.index("class QPSolverInterface {"):hpp.index("class HipQPInterface : public")]
    have = mk.pure_virtuals(base)
    key = lambda p: (p["name"], p["nparams"], p["const"])
    assert sorted(map(key, have)) == sorted(map(key, want))
    derived = hpp[hpp.index("class HipQPInterface : public"):]
    for p in want:
        assert re.search(r"\b%s\s*\([^;{}]*\)\s*(const\s*)?override" % p["name"], derived), p["name"]
    from restartsqp_amd.interface import HipQPInterface
    for p in want:
        assert callable(getattr(HipQPInterface, p["name"], None)), p["name"]


def test_c_abi_sharding_matches_the_python_partitions():
    """rsqp_shard_range / rsqp_balanced_shard (host-only entries of the C ABI, for C++ hosts that run one process per GPU)
    give the partitions of restartsqp_amd/parallel.py: 512 hs0xx QPs on 1 / 2 / 3 / 8 ranks."""
    import ctypes as C
    from restartsqp_amd import capi, parallel, problems
    L = capi.lib()
    ps = problems.hs_batch(512)
    nV = np.array([p.nV for p in ps], np.int32); nC = np.array([p.nC for p in ps], np.int32)
    ip = C.POINTER(C.c_int)
    for world in (1, 2, 3, 8):
        shards = parallel.balanced_shards(ps, world)
        for rank in range(world):
            lo, hi = C.c_int(), C.c_int()
            assert L.rsqp_shard_range(512, rank, world, C.byref(lo), C.byref(hi)) == 0
            assert (lo.value, hi.value) == parallel.shard_range(512, rank, world)
            idx = np.zeros(512, np.int32); cnt = C.c_int()
            assert L.rsqp_balanced_shard(512, nV.ctypes.data_as(ip), nC.ctypes.data_as(ip), rank, world, idx.ctypes.data_as(ip), C.byref(cnt)) == 0
            assert idx[:cnt.value].tolist() == shards[rank].tolist()
    assert L.rsqp_shard_range(10, 3, 3, C.byref(lo), C.byref(hi)) != 0


def test_rccl_entries_fail_loudly_when_the_library_is_missing():
    """on a host without librccl every rsqp_rccl_* entry must return RSQP_ERR_DEVICE with the loader's message (the
    first version called dlerror() twice and built a std::string from NULL). RSQP_RCCL_LIBRARY points the loader at a file that
    does not exist; a child process, because the library is bound once per process."""
    import subprocess, sys
    code = ("import sys, ctypes as C; sys.path.insert(0, %r)\n"
            "from restartsqp_amd import capi\n"
            "L = capi.lib(); buf = C.create_string_buffer(128)\n"
            "rc = L.rsqp_rccl_unique_id(buf)\n"
            "msg = L.rsqp_last_error().decode()\n"
            "assert rc == capi.ERR_DEVICE, rc\n"
            "assert 'not loadable' in msg and 'no_such_rccl' in msg, msg\n"
            "assert L.rsqp_rccl_comm_create(buf, 0, 1, 0, C.byref(C.c_void_p())) == capi.ERR_DEVICE\n"
            "print('ok')\n" % ROOT)
    env = dict(os.environ, RSQP_RCCL_LIBRARY="/nonexistent/no_such_rccl.so")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr + out.stdout
