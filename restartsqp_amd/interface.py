"""Host-side mirror of the reference's QP-solver plug-in interface for the HIP engine.

``HipQPInterface`` has the method names, argument meaning and error behaviour of
``qpOASESInterface`` (reference ``include/sqphot/qpOASESInterface.hpp:37-262``, pure virtuals
of ``include/sqphot/QPsolverInterface.hpp:47-184``) and forwards every call to the C ABI
(``include/rsqp_hip.h``). Nothing is computed here: assembly, products, the QP solve and the
KKT certificate all run in librsqp_hip.so on the GPU. The C++ twin a RestartSQP maintainer
would compile into the reference lives in ``restartsqp_amd/csrc/host/``.
"""
import numpy as np

from . import capi
from .sqptypes import (INVALID_WORKING_SET, LP_NOT_OPTIMAL, QP, QP_NOT_OPTIMAL, IdentityInfo, NLPInfo, OptimalityStatus, Options,
                    SpTripletMat)


class HipQPInterface:
    def __init__(self, nlp_info=None, qptype=QP, options=None, jnlst=None, *, H=None, A=None, g=None, lb=None,
                 ub=None, lbA=None, ubA=None, device=-1):
        """Either ``(NLPInfo, QPType, Options, Journalist)`` -- qpOASESInterface.cpp:35-50, QP of
        size nVar+2*nCon by nCon -- or the plain-QP form ``H=, A=, g=, ...`` with CSC triples
        (qpOASESInterface.cpp:54-94)."""
        self.options_ = options or Options()
        self.jnlst_ = jnlst
        self.qpOptimalStatus_ = OptimalityStatus()
        if nlp_info is not None:
            self.nConstr_QP_ = nlp_info.nCon
            self.nVar_QP_ = nlp_info.nVar + 2 * nlp_info.nCon
            self._s = capi.Solver(self.nVar_QP_, self.nConstr_QP_, device)
        else:
            A_jc, A_ir, A_val = A
            self.nVar_QP_ = len(A_jc) - 1
            self.nConstr_QP_ = len(lbA)
            self._s = capi.Solver(self.nVar_QP_, self.nConstr_QP_, device)
            self._s.set_A_csc(A_jc, A_ir, A_val)
            if H is not None:
                self._s.set_H_csc(*H)
            for which, v in ((capi.VEC_G, g), (capi.VEC_LB, lb), (capi.VEC_UB, ub), (capi.VEC_LBA, lbA),
                             (capi.VEC_UBA, ubA)):
                self._s.set_vector(which, v)   # lb/ub may be longer (QORE length): first nV are read
        self._s.set_options(self.options_.qp_maxiter, self.options_.lp_maxiter)

    # ---- data getters (QPsolverInterface.hpp:47-59)
    def getLb(self):
        return self._s.get_vector(capi.VEC_LB)

    def getUb(self):
        return self._s.get_vector(capi.VEC_UB)

    def getLbA(self):
        return self._s.get_vector(capi.VEC_LBA)

    def getUbA(self):
        return self._s.get_vector(capi.VEC_UBA)

    def getG(self):
        return self._s.get_vector(capi.VEC_G)

    def getH(self):
        return self._s.get_H_csc()

    def getA(self):
        return self._s.get_A_csc()

    # ---- solve
    def optimizeQP(self, stats=None):
        """qpOASESInterface::optimizeQP (:137-224); raises QP_NOT_OPTIMAL like handle_error (:754-756)."""
        nWSR = self._s.optimize_qp()
        if stats is not None:
            stats.qp_iter_addValue(nWSR)
        if not self._s.is_solved():
            raise QP_NOT_OPTIMAL("QP solver did not reach optimality (status %d)" % self._s.status)

    def optimizeLP(self, stats=None):
        """qpOASESInterface::optimizeLP (:227-284); raises LP_NOT_OPTIMAL like handle_error (:714-716)."""
        nWSR = self._s.optimize_lp()
        if stats is not None:
            stats.qp_iter_addValue(nWSR)
        if not self._s.is_solved():
            raise LP_NOT_OPTIMAL("LP solver did not reach optimality (status %d)" % self._s.status)

    # ---- result getters (:290-357)
    def get_optimal_solution(self):
        return self._s.x

    def get_obj_value(self):
        return self._s.objective

    def get_multipliers_bounds(self):
        return self._s.y[:self.nVar_QP_]

    def get_multipliers_constr(self):
        return self._s.y[self.nVar_QP_:]

    def get_working_set(self, W_constr, W_bounds):
        try:
            Wc, Wb = self._s.working_set()
        except capi.RsqpError as e:
            if e.code == capi.ERR_WORKING_SET:
                raise INVALID_WORKING_SET(str(e))
            raise
        W_constr[:] = Wc
        W_bounds[:] = Wb

    def get_status(self):
        return self._s.status

    def test_optimality(self, W_c=None, W_b=None):
        try:
            ok, st, Wc, Wb = self._s.test_optimality()
        except capi.RsqpError as e:
            if e.code == capi.ERR_WORKING_SET:
                raise INVALID_WORKING_SET(str(e))
            raise
        if W_c is not None:
            W_c[:] = Wc
        if W_b is not None:
            W_b[:] = Wb
        o = self.qpOptimalStatus_
        o.compl_violation, o.stationarity_violation = st.compl_violation, st.stationarity_violation
        o.dual_violation, o.primal_violation, o.KKT_error = st.dual_violation, st.primal_violation, st.KKT_error
        return ok

    def get_optimality_status(self):
        return self.qpOptimalStatus_

    # ---- setters (:361-484): (location, value) or a whole vector
    def _set(self, which, a, b=None):
        if b is None:
            self._s.set_vector(which, a)
        else:
            self._s.set_entry(which, int(a), b)

    def set_lb(self, a, b=None):
        self._set(capi.VEC_LB, a, b)

    def set_ub(self, a, b=None):
        self._set(capi.VEC_UB, a, b)

    def set_lbA(self, a, b=None):
        self._set(capi.VEC_LBA, a, b)

    def set_ubA(self, a, b=None):
        self._set(capi.VEC_UBA, a, b)

    def set_g(self, a, b=None):
        self._set(capi.VEC_G, a, b)

    def set_H(self, rhs: SpTripletMat):
        self._s.set_H_triplet(rhs.RowIndex, rhs.ColIndex, rhs.MatVal, rhs.isSymmetric)

    def set_A(self, rhs: SpTripletMat, I_info: IdentityInfo):
        self._s.set_A_triplet(rhs.RowIndex, rhs.ColIndex, rhs.MatVal, I_info.blocks())

    def reset_constraints(self):
        self._s.reset_constraints()

    def WriteQPDataToFile(self, level, category, filename):
        """qpOASES-layout dump "qpOASES" + filename (:791-814), written by the C ABI (rsqp_write_qp_data)."""
        self._s.write_qp_data("qpOASES" + filename, capi.DUMP_QPOASES)

    def WriteQPDataToFileQORE(self, filename):
        """the QORE layout of the same data (src/QOREInterface.cpp:582-598): "QORE_" + filename"""
        self._s.write_qp_data("QORE_" + filename, capi.DUMP_QORE)
