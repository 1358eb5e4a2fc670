"""On-disk QP formats at the C ABI (host-only entry points of librsqp_hip.so, no GPU needed).

WriteQPDataToFile layouts: reference src/qpOASESInterface.cpp:791-814 + src/SpHbMat.cpp:568-578 (qpOASES),
src/QOREInterface.cpp:582-598 + src/SpHbMat.cpp:556-567 (QORE); reader: test/QPsolvers_testers.cpp:48-150.
Fixtures: the 18 dumps of reference test/unsolved_QP_data (tests/golden/qore_dumps)."""
import os

import numpy as np
import pytest

from conftest import dump_paths
from restartsqp_amd import capi, qpdump


def tokens(path):
    with open(path) as f:
        return f.read().split()


@pytest.mark.parametrize("path", dump_paths(), ids=os.path.basename)
def test_qore_dump_round_trip_is_token_identical(path, tmp_path):
    """read a reference dump (C reader), write it back in the QORE layout (C writer): same tokens"""
    d = capi.read_qore_dump(path)
    q = qpdump.QPData(d["nV"], d["nC"], d["H_jc"], d["H_ir"], d["H_val"], d["A_jc"], d["A_ir"], d["A_val"], d["g"],
                      d["lb"], d["ub"], d["lbA"], d["ubA"])
    out = str(tmp_path / "rt.log")
    capi.write_qp_dump(out, q, capi.DUMP_QORE)
    assert tokens(out) == tokens(path)


@pytest.mark.parametrize("path", dump_paths()[:6], ids=os.path.basename)
def test_c_reader_matches_python_reader(path):
    d, q = capi.read_qore_dump(path), qpdump.read_qore_dump(path)
    assert (d["nV"], d["nC"]) == (q.nV, q.nC)
    for k in ("lb", "ub", "lbA", "ubA", "g", "A_jc", "A_ir", "A_val", "H_jc", "H_ir", "H_val"):
        assert np.array_equal(d[k], getattr(q, k)), k


def test_qpoases_layout(tmp_path):
    """order of the sections (qpOASESInterface.cpp:802-809): lb, lbA, ub, ubA, g, then per matrix
    ir[nnz], jc[nV+1], val[nnz] (SpHbMat.cpp:568-578); "%23.16e" / "%d", one number per line"""
    q = qpdump.read_qore_dump(dump_paths()[0])
    out = str(tmp_path / "qpoases.log")
    capi.write_qp_dump(out, q, capi.DUMP_QPOASES)
    lines = open(out).read().split("\n")
    assert lines[-1] == "" and all(len(ln.split()) == 1 for ln in lines[:-1])
    exp = []
    for vec in (q.lb, q.lbA, q.ub, q.ubA, q.g):
        exp += ["%23.16e" % v for v in vec]
    for jc, ir, val in ((q.A_jc, q.A_ir, q.A_val), (q.H_jc, q.H_ir, q.H_val)):
        exp += ["%d" % v for v in ir] + ["%d" % v for v in jc] + ["%23.16e" % v for v in val]
    assert lines[:-1] == exp
    # and the Python writer of the same layout agrees byte for byte
    out2 = str(tmp_path / "qpoases_py.log")
    qpdump.write_qpoases_dump(out2, q)
    assert open(out).read() == open(out2).read()


def test_writer_rejects_bad_arguments(tmp_path):
    q = qpdump.read_qore_dump(dump_paths()[0])
    with pytest.raises(capi.RsqpError):
        capi.write_qp_dump(str(tmp_path / "x.log"), q, layout=7)
    with pytest.raises(capi.RsqpError):
        capi.write_qp_dump(str(tmp_path / "no_such_dir" / "x.log"), q)
    with pytest.raises(capi.RsqpError):
        capi.read_qore_dump(str(tmp_path / "missing.log"))
