"""ctypes binding of ``librsqp_hip.so`` (the C ABI declared in ``include/rsqp_hip.h``).

This is the only way Python reaches the engine: no CPU fallback exists. Loading fails loudly
when the library has not been built (``__graft_entry__.build()`` / ``restartsqp_amd.build``),
and every call that needs a GPU returns ``RSQP_ERR_DEVICE`` -> ``RsqpError`` without one.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RSQP_LIB") or os.path.join(_HERE, "lib", "librsqp_hip.so")   # RSQP_LIB: tuning builds (tools/)

OK = 0
ERR_ARG, ERR_DEVICE, ERR_TOO_LARGE, ERR_WORKING_SET = -1, -2, -3, -4
VEC_G, VEC_LB, VEC_UB, VEC_LBA, VEC_UBA = range(5)
MODE_COLD, MODE_HOT_VECTORS, MODE_HOT_MATRICES, MODE_WARM_REINIT = range(4)
QP_OPTIMAL, QPERROR_INFEASIBLE, QPERROR_UNBOUNDED = 20, 22, 23
ACTIVE_ABOVE, ACTIVE_BELOW, ACTIVE_BOTH_SIDE, INACTIVE = 1, -1, -99, 0

ip = C.POINTER(C.c_int)
dp = C.POINTER(C.c_double)

# every exported symbol of include/rsqp_hip.h: name -> (restype, argtypes)
fp = C.POINTER(C.c_float)
SYMBOLS = {
    "rsqp_dense_gemm": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_double), C.c_int,
                                  C.POINTER(C.c_double), C.c_int, C.c_double, C.POINTER(C.c_double), C.c_int, C.c_int, fp]),
    "rsqp_dense_qr": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                C.c_double, C.POINTER(C.c_int), fp]),
    "rsqp_dense_chol_inverse": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.c_double,
                                          C.POINTER(C.c_int), fp]),
    "rsqp_version": (C.c_char_p, []),
    "rsqp_build_hash": (C.c_char_p, []),
    "rsqp_device_count": (C.c_int, []),
    "rsqp_last_error": (C.c_char_p, []),
    "rsqp_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "rsqp_destroy": (None, [C.c_void_p]),
    "rsqp_set_options": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "rsqp_set_reinit_guess": (C.c_int, [C.c_void_p, C.c_int]),
    "rsqp_get_last_mode": (C.c_int, [C.c_void_p]),
    "rsqp_get_large_path": (C.c_int, [C.c_void_p]),
    "rsqp_get_nV": (C.c_int, [C.c_void_p]),
    "rsqp_get_nC": (C.c_int, [C.c_void_p]),
    "rsqp_write_qp_dump": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, dp, dp, dp, dp, dp, ip, ip, dp, ip, ip, dp]),
    "rsqp_write_qp_data": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "rsqp_read_qore_dump_sizes": (C.c_int, [C.c_char_p, ip, ip, ip, ip]),
    "rsqp_read_qore_dump": (C.c_int, [C.c_char_p, dp, dp, dp, dp, dp, ip, ip, dp, ip, ip, dp]),
    "rsqp_set_engine_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "rsqp_get_engine_profile": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "rsqp_engine_profile_names": (C.c_int, [C.c_void_p, C.c_int]),
    "rsqp_get_setup_profile": (C.c_int, [C.c_void_p, dp]),
    "rsqp_get_structure_seconds": (C.c_double, [C.c_void_p, C.c_int]),
    "rsqp_set_engine": (C.c_int, [C.c_void_p, C.c_int]),
    "rsqp_get_engine": (C.c_int, [C.c_void_p]),
    "rsqp_set_A_triplet": (C.c_int, [C.c_void_p, C.c_int, ip, ip, dp, C.c_int, ip, ip, ip, dp]),
    "rsqp_set_H_triplet": (C.c_int, [C.c_void_p, C.c_int, ip, ip, dp, C.c_int]),
    "rsqp_set_A_csc": (C.c_int, [C.c_void_p, ip, ip, dp]),
    "rsqp_set_H_csc": (C.c_int, [C.c_void_p, ip, ip, dp]),
    "rsqp_get_A_nnz": (C.c_int, [C.c_void_p]),
    "rsqp_get_H_nnz": (C.c_int, [C.c_void_p]),
    "rsqp_get_A_csc": (C.c_int, [C.c_void_p, ip, ip, dp, ip]),
    "rsqp_get_H_csc": (C.c_int, [C.c_void_p, ip, ip, dp, ip]),
    "rsqp_set_vector": (C.c_int, [C.c_void_p, C.c_int, dp]),
    "rsqp_set_entry": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double]),
    "rsqp_get_vector": (C.c_int, [C.c_void_p, C.c_int, dp]),
    "rsqp_reset_constraints": (C.c_int, [C.c_void_p]),
    "rsqp_optimize_qp": (C.c_int, [C.c_void_p, ip]),
    "rsqp_optimize_lp": (C.c_int, [C.c_void_p, ip]),
    "rsqp_solve": (C.c_int, [C.c_void_p, C.c_int, ip, dp, dp, ip]),
    "rsqp_get_primal": (C.c_int, [C.c_void_p, dp]),
    "rsqp_get_dual": (C.c_int, [C.c_void_p, dp]),
    "rsqp_get_objective": (C.c_double, [C.c_void_p]),
    "rsqp_get_status": (C.c_int, [C.c_void_p]),
    "rsqp_is_solved": (C.c_int, [C.c_void_p]),
    "rsqp_get_working_set_raw": (C.c_int, [C.c_void_p, ip, ip]),
    "rsqp_get_working_set": (C.c_int, [C.c_void_p, ip, ip]),
    "rsqp_test_optimality": (C.c_int, [C.c_void_p, ip, ip, C.c_void_p]),
    "rsqp_A_times": (C.c_int, [C.c_void_p, dp, dp]),
    "rsqp_A_transposed_times": (C.c_int, [C.c_void_p, dp, dp]),
    "rsqp_H_times": (C.c_int, [C.c_void_p, dp, dp]),
    "rsqp_batch_create": (C.c_int, [C.c_int, ip, ip, ip, ip, dp, ip, ip, dp, C.c_int, C.POINTER(C.c_void_p)]),
    "rsqp_batch_destroy": (None, [C.c_void_p]),
    "rsqp_batch_set_vectors": (C.c_int, [C.c_void_p, dp, dp, dp, dp, dp]),
    "rsqp_batch_set_matrix_values": (C.c_int, [C.c_void_p, dp, dp]),
    "rsqp_batch_solve": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "rsqp_batch_sync": (C.c_int, [C.c_void_p]),
    "rsqp_batch_set_keep_state": (C.c_int, [C.c_void_p, C.c_int]),
    "rsqp_batch_get_last_kernel": (C.c_int, [C.c_void_p]),
    "rsqp_batch_last_solve_ms": (C.c_float, [C.c_void_p]),
    "rsqp_batch_timer_start": (C.c_int, [C.c_void_p]),
    "rsqp_batch_timer_stop_ms": (C.c_float, [C.c_void_p]),
    "rsqp_batch_get_results": (C.c_int, [C.c_void_p, dp, dp, ip, ip, ip, ip, dp]),
    "rsqp_batch_test_optimality": (C.c_int, [C.c_void_p, C.c_void_p, ip]),
    "rsqp_batch_record_stride": (C.c_int, [C.c_void_p]),
    "rsqp_batch_pack_records_dev": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rsqp_batch_pack_records_host": (C.c_int, [C.c_void_p, dp]),
    "rsqp_shard_range": (C.c_int, [C.c_int, C.c_int, C.c_int, ip, ip]),
    "rsqp_balanced_shard": (C.c_int, [C.c_int, ip, ip, C.c_int, C.c_int, ip, ip]),
    "rsqp_rccl_unique_id": (C.c_int, [C.c_char_p]),
    "rsqp_rccl_comm_create": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "rsqp_rccl_comm_destroy": (C.c_int, [C.c_void_p]),
    "rsqp_rccl_broadcast_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p]),
    "rsqp_batch_allgather_records": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "rsqp_time_value_refresh": (C.c_int, [C.c_void_p, C.c_int, fp, fp]),
    "rsqp_time_value_refresh_fused": (C.c_int, [C.c_void_p, C.c_int, fp]),
    "rsqp_time_large_kernel": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp]),
    "rsqp_spmv_plan_create": (C.c_int, [C.c_int, C.c_int, ip, ip, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "rsqp_spmv_plan_destroy": (None, [C.c_void_p]),
    "rsqp_spmv_plan_upload": (C.c_int, [C.c_void_p, dp, dp, C.c_int]),
    "rsqp_spmv_plan_run": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "rsqp_spmv_plan_download": (C.c_int, [C.c_void_p, dp, C.c_int]),
    "rsqp_spmv_plan_variant": (C.c_int, [C.c_void_p, C.c_int, ip]),
}


class OptimalityStatus(C.Structure):
    _fields_ = [("primal_violation", C.c_double), ("dual_violation", C.c_double),
                ("compl_violation", C.c_double), ("stationarity_violation", C.c_double),
                ("KKT_error", C.c_double)]


class RsqpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("rsqp error %d: %s" % (code, msg))
        self.code = code


_LIB = None


def lib():
    """Load librsqp_hip.so; raises if it is missing (there is no fallback path)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _LIB = L
    return _LIB


def check(rc):
    if rc < 0:
        raise RsqpError(rc, lib().rsqp_last_error().decode())
    return rc


def _d(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def _i(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int32)


def _dp(a):
    return None if a is None else a.ctypes.data_as(dp)


def _ip(a):
    return None if a is None else a.ctypes.data_as(ip)


def device_count():
    return lib().rsqp_device_count()


DUMP_QPOASES, DUMP_QORE = 0, 1


def write_qp_dump(path, qp, layout=DUMP_QORE):
    """WriteQPDataToFile in the reference's layouts (host only): `qp` has the QPData fields."""
    a = [_d(qp.lb), _d(qp.ub), _d(qp.lbA), _d(qp.ubA), _d(qp.g)]
    Ajc, Air, Aval = _i(qp.A_jc), _i(qp.A_ir), _d(qp.A_val)
    Hjc, Hir, Hval = _i(qp.H_jc), _i(qp.H_ir), _d(qp.H_val)
    check(lib().rsqp_write_qp_dump(os.fsencode(path), layout, qp.nV, qp.nC, *[_dp(v) for v in a], _ip(Ajc), _ip(Air),
                                   _dp(Aval), _ip(Hjc), _ip(Hir), _dp(Hval)))


def read_qore_dump(path):
    """C reader of the QORE dump layout -> dict of arrays (CSC matrices)."""
    n = [C.c_int(0) for _ in range(4)]
    check(lib().rsqp_read_qore_dump_sizes(os.fsencode(path), *[C.byref(v) for v in n]))
    nV, nC, nnzA, nnzH = (v.value for v in n)
    d = dict(nV=nV, nC=nC, lb=np.zeros(nV), ub=np.zeros(nV), lbA=np.zeros(nC), ubA=np.zeros(nC), g=np.zeros(nV),
             A_jc=np.zeros(nV + 1, np.int32), A_ir=np.zeros(nnzA, np.int32), A_val=np.zeros(nnzA),
             H_jc=np.zeros(nV + 1, np.int32), H_ir=np.zeros(nnzH, np.int32), H_val=np.zeros(nnzH))
    check(lib().rsqp_read_qore_dump(os.fsencode(path), _dp(d["lb"]), _dp(d["ub"]), _dp(d["lbA"]), _dp(d["ubA"]), _dp(d["g"]),
                                    _ip(d["A_jc"]), _ip(d["A_ir"]), _dp(d["A_val"]), _ip(d["H_jc"]), _ip(d["H_ir"]),
                                    _dp(d["H_val"])))
    return d


class Solver:
    """Thin RAII wrapper over one ``rsqp_solver`` handle."""

    def __init__(self, nV, nC, device=-1):
        self.nV, self.nC = nV, nC
        h = C.c_void_p()
        check(lib().rsqp_create(nV, nC, device, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().rsqp_destroy(self._h)
            self._h = None

    __del__ = close

    def write_qp_data(self, path, layout=DUMP_QPOASES):
        """WriteQPDataToFile for the data held by this handle."""
        check(lib().rsqp_write_qp_data(self._h, os.fsencode(path), layout))

    def set_engine_profiling(self, on=True):
        check(lib().rsqp_set_engine_profiling(self._h, int(bool(on))))

    def engine_profile(self):
        """per-kernel-class device time / algorithmic bytes of the HBM-resident engine since its creation
        (rsqp_get_engine_profile), or None when the handle runs the LDS-resident engine"""
        n = lib().rsqp_get_engine_profile(self._h, None, 0)
        if n <= 0:
            return None
        buf = (C.c_double * (4 * n))()
        names = (C.c_char_p * n)()
        lib().rsqp_get_engine_profile(self._h, C.cast(buf, C.c_void_p), n)
        lib().rsqp_engine_profile_names(C.cast(names, C.c_void_p), n)
        out = {}
        for k in range(n):
            calls, ms, byts, peak = buf[4 * k], buf[4 * k + 1], buf[4 * k + 2], buf[4 * k + 3]
            if calls > 0:
                out[names[k].decode()] = {"calls": int(calls), "ms_total": ms, "us_per_call": 1e3 * ms / calls,
                                          "algorithmic_bytes": byts, "achieved": byts / (ms * 1e-3) / 1e9 if ms > 0 else None,
                                          "unit": "GB/s", "peak": 8000.0, "bound": "hbm",
                                          "frac": byts / (ms * 1e-3) / 1e9 / 8000.0 if ms > 0 else None}
        return out

    def setup_profile(self):
        """the last blocked (matrix-core) set-up on the HBM-resident engine (rsqp_get_setup_profile), or None"""
        buf = np.zeros(8)
        if lib().rsqp_get_setup_profile(self._h, buf.ctypes.data_as(dp)) != 1:
            return None
        return {"nFR": int(buf[0]), "nAC": int(buf[1]), "nZ": int(buf[2]), "ms_qr_q_rinv": buf[3], "ms_zhz_chol_inv": buf[4],
                "flops_qr_q_rinv": buf[5], "flops_zhz_chol_inv": buf[6], "range_space": bool(buf[7])}

    def structure_seconds(self, which=0):
        return lib().rsqp_get_structure_seconds(self._h, which)

    def time_value_refresh(self, repeats=50):
        a, b = C.c_float(0), C.c_float(0)
        check(lib().rsqp_time_value_refresh(self._h, repeats, C.byref(a), C.byref(b)))
        return a.value, b.value

    def time_value_refresh_fused(self, repeats=50):
        a = C.c_float(0)
        check(lib().rsqp_time_value_refresh_fused(self._h, repeats, C.byref(a)))
        return a.value

    def set_engine(self, engine):
        """0 automatic, 1 LDS-resident kernel, 2 HBM-resident engine."""
        check(lib().rsqp_set_engine(self._h, engine))

    @property
    def engine(self):
        return lib().rsqp_get_engine(self._h)

    def set_reinit_guess(self, from_y0):
        """warm re-initialisation without guessed constraints: from_y0=False as qpOASES, from A x0 (the library default, the
        reference's behaviour) or -- opt-in, from_y0=True -- sides from sign(y0). No default argument: a bare call must not
        silently leave the reference's rule"""
        check(lib().rsqp_set_reinit_guess(self._h, int(bool(from_y0))))

    MODE_NAMES = {-1: "none", 0: "init (cold)", 1: "hotstart(g, lb, ub, lbA, ubA)", 2: "hotstart(H, g, A, ..) -- new matrices",
                  3: "init(.., x_qp, y_qp, &bounds) -- status flip"}

    def last_mode(self):
        """RSQP_MODE_* the dispatch of the last optimize_qp / optimize_lp chose"""
        return lib().rsqp_get_last_mode(self._h)

    LARGE_PATHS = {-1: "none", 0: "null-space", 1: "range-space (diagonal H)", 2: "general range-space (banded H^-1)",
                   3: "general range-space (dense H^-1)", 4: "general range-space (dense H^-1 + static tableau [I; A] H^-1 [I A'])"}

    def large_path(self):
        """which formulation of the HBM-resident engine holds this handle's factors (rsqp_get_large_path)"""
        return lib().rsqp_get_large_path(self._h)

    def set_options(self, qp_maxiter=1000, lp_maxiter=100):
        check(lib().rsqp_set_options(self._h, qp_maxiter, lp_maxiter))

    def set_A_triplet(self, irow, jcol, val, ident=()):
        irow, jcol, val = _i(irow), _i(jcol), _d(val)
        ii = _i([b[0] for b in ident]); ij = _i([b[1] for b in ident])
        isz = _i([b[2] for b in ident]); iv = _d([b[3] for b in ident])
        check(lib().rsqp_set_A_triplet(self._h, len(val), _ip(irow), _ip(jcol), _dp(val), len(ident), _ip(ii), _ip(ij),
                                       _ip(isz), _dp(iv)))

    def set_H_triplet(self, irow, jcol, val, is_symmetric=True):
        irow, jcol, val = _i(irow), _i(jcol), _d(val)
        check(lib().rsqp_set_H_triplet(self._h, len(val), _ip(irow), _ip(jcol), _dp(val), int(is_symmetric)))

    def set_A_csc(self, jc, ir, val):
        jc, ir, val = _i(jc), _i(ir), _d(val)
        check(lib().rsqp_set_A_csc(self._h, _ip(jc), _ip(ir), _dp(val)))

    def set_H_csc(self, jc, ir, val):
        jc, ir, val = _i(jc), _i(ir), _d(val)
        check(lib().rsqp_set_H_csc(self._h, _ip(jc), _ip(ir), _dp(val)))

    def _get_csc(self, which):
        nnz = check(getattr(lib(), "rsqp_get_%s_nnz" % which)(self._h))
        ncol = self.nV
        jc = np.zeros(ncol + 1, np.int32); ir = np.zeros(nnz, np.int32); val = np.zeros(nnz)
        order = np.zeros(nnz, np.int32)
        check(getattr(lib(), "rsqp_get_%s_csc" % which)(self._h, _ip(jc), _ip(ir), _dp(val), _ip(order)))
        return jc, ir, val, order

    def get_A_csc(self):
        return self._get_csc("A")

    def get_H_csc(self):
        return self._get_csc("H")

    def set_vector(self, which, v):
        v = _d(v)
        n = self.nV if which <= VEC_UB else self.nC
        if len(v) < n:
            raise ValueError("vector too short")
        check(lib().rsqp_set_vector(self._h, which, _dp(v)))

    def set_entry(self, which, loc, value):
        check(lib().rsqp_set_entry(self._h, which, loc, float(value)))

    def get_vector(self, which):
        out = np.zeros(self.nV if which <= VEC_UB else self.nC)
        check(lib().rsqp_get_vector(self._h, which, _dp(out)))
        return out

    def reset_constraints(self):
        check(lib().rsqp_reset_constraints(self._h))

    def optimize_qp(self):
        n = C.c_int(0)
        check(lib().rsqp_optimize_qp(self._h, C.byref(n)))
        return n.value

    def optimize_lp(self):
        n = C.c_int(0)
        check(lib().rsqp_optimize_lp(self._h, C.byref(n)))
        return n.value

    def solve(self, mode, nWSR, x0=None, y0=None, guess_b=None):
        n = C.c_int(nWSR)
        x0, y0, gb = _d(x0), _d(y0), _i(guess_b)
        check(lib().rsqp_solve(self._h, mode, C.byref(n), _dp(x0), _dp(y0), _ip(gb)))
        return n.value

    @property
    def x(self):
        out = np.zeros(self.nV)
        check(lib().rsqp_get_primal(self._h, _dp(out)))
        return out

    @property
    def y(self):
        out = np.zeros(self.nV + self.nC)
        check(lib().rsqp_get_dual(self._h, _dp(out)))
        return out

    @property
    def objective(self):
        return lib().rsqp_get_objective(self._h)

    @property
    def status(self):
        return lib().rsqp_get_status(self._h)

    def is_solved(self):
        return bool(lib().rsqp_is_solved(self._h))

    def working_set_raw(self):
        wb = np.zeros(self.nV, np.int32); wc = np.zeros(self.nC, np.int32)
        check(lib().rsqp_get_working_set_raw(self._h, _ip(wb), _ip(wc)))
        return wb, wc

    def working_set(self):
        Wc = np.zeros(self.nC, np.int32); Wb = np.zeros(self.nV, np.int32)
        check(lib().rsqp_get_working_set(self._h, _ip(Wc), _ip(Wb)))
        return Wc, Wb

    def test_optimality(self):
        Wc = np.zeros(self.nC, np.int32); Wb = np.zeros(self.nV, np.int32)
        st = OptimalityStatus()
        rc = check(lib().rsqp_test_optimality(self._h, _ip(Wc), _ip(Wb), C.byref(st)))
        return bool(rc), st, Wc, Wb

    def A_times(self, p):
        p = _d(p); out = np.zeros(self.nC)
        check(lib().rsqp_A_times(self._h, _dp(p), _dp(out)))
        return out

    def A_transposed_times(self, p):
        p = _d(p); out = np.zeros(self.nV)
        check(lib().rsqp_A_transposed_times(self._h, _dp(p), _dp(out)))
        return out

    def H_times(self, p):
        p = _d(p); out = np.zeros(self.nV)
        check(lib().rsqp_H_times(self._h, _dp(p), _dp(out)))
        return out


class Batch:
    """A batch of independent QPs (``rsqp_batch``): problems given as a list of QPData-like
    objects with fields nV, nC, A_jc/A_ir/A_val, H_jc/H_ir/H_val, g, lb, ub, lbA, ubA."""

    def __init__(self, problems, device=-1):
        self.problems = problems
        self.nq = len(problems)
        self.nV = _i([p.nV for p in problems]); self.nC = _i([p.nC for p in problems])
        Ajc = _i(np.concatenate([p.A_jc for p in problems]))
        Air = _i(np.concatenate([p.A_ir for p in problems] + [np.zeros(0, np.int32)]))
        Aval = _d(np.concatenate([p.A_val for p in problems] + [np.zeros(0)]))
        haveH = all(p.H_jc is not None for p in problems)
        Hjc = _i(np.concatenate([p.H_jc for p in problems])) if haveH else None
        Hir = _i(np.concatenate([p.H_ir for p in problems] + [np.zeros(0, np.int32)])) if haveH else None
        Hval = _d(np.concatenate([p.H_val for p in problems] + [np.zeros(0)])) if haveH else None
        h = C.c_void_p()
        check(lib().rsqp_batch_create(self.nq, _ip(self.nV), _ip(self.nC), _ip(Ajc), _ip(Air), _dp(Aval), _ip(Hjc),
                                      _ip(Hir), _dp(Hval), device, C.byref(h)))
        self._h = h
        self.offV = np.concatenate([[0], np.cumsum(self.nV)]).astype(np.int64)
        self.offC = np.concatenate([[0], np.cumsum(self.nC)]).astype(np.int64)
        self.set_vectors_from(problems)

    def close(self):
        if getattr(self, "_h", None):
            lib().rsqp_batch_destroy(self._h)
            self._h = None

    __del__ = close

    def set_vectors(self, g, lb, ub, lbA, ubA):
        g, lb, ub, lbA, ubA = _d(g), _d(lb), _d(ub), _d(lbA), _d(ubA)
        check(lib().rsqp_batch_set_vectors(self._h, _dp(g), _dp(lb), _dp(ub), _dp(lbA), _dp(ubA)))

    def set_vectors_from(self, problems):
        cat = lambda name: np.concatenate([getattr(p, name) for p in problems] + [np.zeros(0)])
        self.set_vectors(cat("g"), cat("lb"), cat("ub"), cat("lbA"), cat("ubA"))

    def set_matrix_values(self, Aval=None, Hval=None):
        Aval, Hval = _d(Aval), _d(Hval)
        check(lib().rsqp_batch_set_matrix_values(self._h, _dp(Aval), _dp(Hval)))

    def solve(self, mode=MODE_COLD, max_nWSR=1000, sync=True):
        check(lib().rsqp_batch_solve(self._h, mode, max_nWSR))
        if sync:
            check(lib().rsqp_batch_sync(self._h))

    def set_keep_state(self, keep):
        check(lib().rsqp_batch_set_keep_state(self._h, int(bool(keep))))

    def last_kernel(self):
        """0 LDS null-space kernels, 1 tableau kernel with 8 lanes per problem, 2 lane-per-problem kernel (rsqp_batch_get_last_kernel)"""
        return lib().rsqp_batch_get_last_kernel(self._h)

    def last_solve_ms(self):
        return lib().rsqp_batch_last_solve_ms(self._h)

    def timer_start(self):
        check(lib().rsqp_batch_timer_start(self._h))

    def timer_stop_ms(self):
        return lib().rsqp_batch_timer_stop_ms(self._h)

    def results(self):
        sV, sC = int(self.offV[-1]), int(self.offC[-1])
        x = np.zeros(sV); y = np.zeros(sV + sC); wb = np.zeros(sV, np.int32); wc = np.zeros(sC, np.int32)
        st = np.zeros(self.nq, np.int32); nw = np.zeros(self.nq, np.int32); obj = np.zeros(self.nq)
        check(lib().rsqp_batch_get_results(self._h, _dp(x), _dp(y), _ip(wb), _ip(wc), _ip(st), _ip(nw), _dp(obj)))
        out = []
        for q in range(self.nq):
            v0, v1, c0, c1 = self.offV[q], self.offV[q + 1], self.offC[q], self.offC[q + 1]
            yo = v0 + c0
            out.append(dict(x=x[v0:v1], y=y[yo:yo + (v1 - v0) + (c1 - c0)], ws_b=wb[v0:v1], ws_c=wc[c0:c1],
                            status=int(st[q]), nWSR=int(nw[q]), obj=float(obj[q])))
        return out

    @property
    def record_stride(self):
        return lib().rsqp_batch_record_stride(self._h)

    def pack_records_dev(self, dev_ptr):
        """fixed-stride result records (parallel.pack_records layout) into device memory at `dev_ptr`
        (nq * record_stride doubles, e.g. a torch tensor's data_ptr()); asynchronous on the batch's stream"""
        check(lib().rsqp_batch_pack_records_dev(self._h, C.c_void_p(dev_ptr)))

    def pack_records(self):
        """the same records, packed on the device and copied to the host: array [nq, record_stride]"""
        out = np.zeros(self.nq * self.record_stride)
        check(lib().rsqp_batch_pack_records_host(self._h, _dp(out)))
        return out.reshape(self.nq, self.record_stride)

    def test_optimality(self):
        st = (OptimalityStatus * self.nq)()
        ok = np.zeros(self.nq, np.int32)
        check(lib().rsqp_batch_test_optimality(self._h, C.cast(st, C.c_void_p), _ip(ok)))
        return ok, [st[q].KKT_error for q in range(self.nq)]

    def allgather_records(self, comm, count_per_rank, all_dev_ptr):
        """native RCCL: pack this rank's records into its slot of the DEVICE buffer at `all_dev_ptr`
        (world * count_per_rank * record_stride doubles) and all-gather in place; returns when done"""
        check(lib().rsqp_batch_allgather_records(self._h, comm._c, int(count_per_rank), C.c_void_p(all_dev_ptr)))


def rccl_unique_id():
    """ncclGetUniqueId through the C ABI: 128 bytes that rank 0 ships to every other rank"""
    buf = C.create_string_buffer(128)
    check(lib().rsqp_rccl_unique_id(buf))
    return buf.raw


class RcclComm:
    """an ncclComm_t created through the C ABI (rsqp_rccl_comm_create): one per rank, bound to `device`"""

    def __init__(self, uid, rank, world, device):
        c = C.c_void_p()
        check(lib().rsqp_rccl_comm_create(C.create_string_buffer(bytes(uid), 128), rank, world, device, C.byref(c)))
        self._c, self.rank, self.world = c, rank, world

    def broadcast_dev(self, dev_ptr, nbytes, root=0, stream=None):
        check(lib().rsqp_rccl_broadcast_dev(self._c, C.c_void_p(dev_ptr), int(nbytes), root, C.c_void_p(stream or 0)))

    def close(self):
        if getattr(self, "_c", None):
            lib().rsqp_rccl_comm_destroy(self._c)
            self._c = None


class SpmvPlan:
    def __init__(self, nrow, ncol, jc, ir, nbatch, device=-1):
        jc, ir = _i(jc), _i(ir)
        self.nrow, self.ncol, self.nnz, self.nbatch = nrow, ncol, int(jc[ncol]), nbatch
        h = C.c_void_p()
        check(lib().rsqp_spmv_plan_create(nrow, ncol, _ip(jc), _ip(ir), nbatch, device, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().rsqp_spmv_plan_destroy(self._h)
            self._h = None

    __del__ = close

    def upload(self, vals=None, xin=None, transposed=False):
        vals, xin = _d(vals), _d(xin)
        check(lib().rsqp_spmv_plan_upload(self._h, _dp(vals), _dp(xin), int(transposed)))

    def run(self, transposed=False, repeats=1):
        ms = C.c_float(0)
        check(lib().rsqp_spmv_plan_run(self._h, int(transposed), repeats, C.byref(ms)))
        return ms.value

    def variant(self, transposed=False):
        """(kernel code, uses 16-bit indices) of the product -- see rsqp_spmv_plan_variant."""
        i16 = C.c_int(0)
        v = check(lib().rsqp_spmv_plan_variant(self._h, int(transposed), C.byref(i16)))
        return v, bool(i16.value)

    def download(self, transposed=False):
        out = np.zeros(self.nbatch * (self.ncol if transposed else self.nrow))
        check(lib().rsqp_spmv_plan_download(self._h, _dp(out), int(transposed)))
        return out.reshape(self.nbatch, -1)
