"""Reader / writer for the on-disk QP dump formats of the reference.

QORE layout (writer: reference ``src/QOREInterface.cpp:589-597`` +
``src/SpHbMat.cpp:556-567``; reader: ``test/QPsolvers_testers.cpp:48-150``), one number
per line::

    nV, nC, nnz(A), nnz(H)
    lb[nV+nC]  ub[nV+nC]  g[nV]
    A as CSR: rowptr[nC+1] col[nnzA] val[nnzA]
    H as CSR: rowptr[nV+1] col[nnzH] val[nnzH]      (full symmetric)

qpOASES layout (``src/qpOASESInterface.cpp:802-809`` + ``src/SpHbMat.cpp:568-578``)::

    lb[nV] lbA[nC] ub[nV] ubA[nC] g[nV]
    A as CSC: ir[nnzA] jc[nV+1] val[nnzA]
    H as CSC: ir[nnzH] jc[nV+1] val[nnzH]
"""
from dataclasses import dataclass

import numpy as np


@dataclass
class QPData:
    """Plain QP  min 1/2 x'Hx + g'x  s.t. lbA <= Ax <= ubA, lb <= x <= ub  (CSC matrices)."""
    nV: int
    nC: int
    H_jc: np.ndarray
    H_ir: np.ndarray
    H_val: np.ndarray
    A_jc: np.ndarray
    A_ir: np.ndarray
    A_val: np.ndarray
    g: np.ndarray
    lb: np.ndarray
    ub: np.ndarray
    lbA: np.ndarray
    ubA: np.ndarray
    name: str = ""

    def dense_A(self):
        return csc_to_dense(self.nC, self.nV, self.A_jc, self.A_ir, self.A_val)

    def dense_H(self):
        return csc_to_dense(self.nV, self.nV, self.H_jc, self.H_ir, self.H_val)


def csc_to_dense(nrow, ncol, jc, ir, val):
    M = np.zeros((nrow, ncol))
    for c in range(ncol):
        for k in range(jc[c], jc[c + 1]):
            M[ir[k], c] = val[k]
    return M


def csr_to_csc(nrow, ncol, rp, ci, val):
    """Same conversion as ``convert_csr_to_csc`` in test/QPsolvers_testers.cpp:18-29 (via a
    dense copy there; done directly here): column-major entry order, rows ascending."""
    nnz = len(val)
    jc = np.zeros(ncol + 1, np.int32)
    for k in range(nnz):
        jc[ci[k] + 1] += 1
    jc = np.cumsum(jc).astype(np.int32)
    ir = np.zeros(nnz, np.int32)
    out = np.zeros(nnz)
    fill = np.zeros(ncol, np.int64)
    for r in range(nrow):
        for k in range(rp[r], rp[r + 1]):
            c = ci[k]
            p = jc[c] + fill[c]
            fill[c] += 1
            ir[p] = r
            out[p] = val[k]
    return jc, ir, out


def dense_to_csc(M, tol=1.0e-16):
    """SpHbMat dense constructor rule: keep |entry| > m_eps (src/SpHbMat.cpp:100-110)."""
    M = np.asarray(M, dtype=np.float64)
    nrow, ncol = M.shape
    mask = np.abs(M.T) > tol              # [col, row]
    counts = mask.sum(axis=1)
    jc = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    cols, rows = np.nonzero(mask)
    return jc, rows.astype(np.int32), M.T[mask].astype(np.float64)


def read_qore_dump(path):
    with open(path) as f:
        tok = f.read().split()
    pos = 0

    def ints(n):
        nonlocal pos
        out = np.array([int(t) for t in tok[pos:pos + n]], dtype=np.int32)
        pos += n
        return out

    def dbls(n):
        nonlocal pos
        out = np.array([float(t) for t in tok[pos:pos + n]], dtype=np.float64)
        pos += n
        return out

    nV, nC, nnzA, nnzH = (int(v) for v in ints(4))
    lb_all, ub_all, g = dbls(nV + nC), dbls(nV + nC), dbls(nV)
    A_rp, A_ci, A_v = ints(nC + 1), ints(nnzA), dbls(nnzA)
    H_rp, H_ci, H_v = ints(nV + 1), ints(nnzH), dbls(nnzH)
    if pos != len(tok):
        raise ValueError("%s: %d trailing tokens" % (path, len(tok) - pos))
    A_jc, A_ir, A_val = csr_to_csc(nC, nV, A_rp, A_ci, A_v)
    H_jc, H_ir, H_val = csr_to_csc(nV, nV, H_rp, H_ci, H_v)
    import os
    return QPData(nV, nC, H_jc, H_ir, H_val, A_jc, A_ir, A_val, g, lb_all[:nV].copy(), ub_all[:nV].copy(),
                  lb_all[nV:].copy(), ub_all[nV:].copy(), name=os.path.basename(path))


def _csc_to_csr(nrow, ncol, jc, ir, val):
    rp, ci, v = csr_to_csc(ncol, nrow, jc, ir, val)  # transpose trick
    return rp, ci, v


def write_qore_dump(path, qp):
    A_rp, A_ci, A_v = _csc_to_csr(qp.nC, qp.nV, qp.A_jc, qp.A_ir, qp.A_val)
    H_rp, H_ci, H_v = _csc_to_csr(qp.nV, qp.nV, qp.H_jc, qp.H_ir, qp.H_val)
    with open(path, "w") as f:
        for n in (qp.nV, qp.nC, len(A_v), len(H_v)):
            f.write("%d\n" % n)
        for vec in (np.concatenate([qp.lb, qp.lbA]), np.concatenate([qp.ub, qp.ubA]), qp.g):
            for v in vec:
                f.write("%23.16e\n" % v)
        for ptr, idx, val in ((A_rp, A_ci, A_v), (H_rp, H_ci, H_v)):
            for v in ptr:
                f.write("%d\n" % v)
            for v in idx:
                f.write("%d\n" % v)
            for v in val:
                f.write("%23.16e\n" % v)


def write_qpoases_dump(path, qp):
    with open(path, "w") as f:
        for vec in (qp.lb, qp.lbA, qp.ub, qp.ubA, qp.g):
            for v in vec:
                f.write("%23.16e\n" % v)
        for jc, ir, val in ((qp.A_jc, qp.A_ir, qp.A_val), (qp.H_jc, qp.H_ir, qp.H_val)):
            for v in ir:
                f.write("%d\n" % v)
            for v in jc:
                f.write("%d\n" % v)
            for v in val:
                f.write("%23.16e\n" % v)
