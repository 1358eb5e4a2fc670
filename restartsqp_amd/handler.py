"""Mirror of the caller side of the boundary: ``QPhandler`` (reference ``src/QPhandler.cpp``).

In RestartSQP this class stays as it is; it is restated here only so that tests and the
replay harness can drive ``HipQPInterface`` with exactly the call sequence
``Algorithm::setupQP`` produces (``src/Algorithm.cpp:645-697``): per-element virtual setters,
``set_A`` with the ``[J I -I]`` identity descriptor, ``solveQP`` followed by the mandatory
certificate. NEW_FORMULATION=false, non-QORE branch.
"""
import numpy as np

from .interface import HipQPInterface
from .sqptypes import INF, QP, QP_NOT_OPTIMAL, IdentityInfo, NLPInfo, Options


class QPhandler:
    def __init__(self, nlp_info: NLPInfo, qptype=QP, jnlst=None, options=None, device=-1):
        self.nlp_info_ = nlp_info
        self.nConstr_QP_ = nlp_info.nCon                       # QPhandler.cpp:39
        self.nVar_QP_ = nlp_info.nVar + 2 * nlp_info.nCon      # :40
        n, m = nlp_info.nVar, nlp_info.nCon
        self.I_info_A_ = IdentityInfo([1, 1], [n + 1, n + m + 1], [m, m], [1.0, -1.0])  # :41-51
        self.W_b_ = np.zeros(self.nVar_QP_, np.int32)
        self.W_c_ = np.zeros(self.nConstr_QP_, np.int32)
        self.solverInterface_ = HipQPInterface(nlp_info, qptype, options or Options(), jnlst, device=device)
        self.qpOptimalStatus_ = None

    def set_bounds(self, delta, x_l, x_u, x_k, c_l, c_u, c_k):  # :167-201
        s = self.solverInterface_
        n, m = self.nlp_info_.nVar, self.nlp_info_.nCon
        for i in range(m):
            s.set_lbA(i, c_l[i] - c_k[i])
            s.set_ubA(i, c_u[i] - c_k[i])
        for i in range(n):
            s.set_lb(i, max(x_l[i] - x_k[i], -delta))
            s.set_ub(i, min(x_u[i] - x_k[i], delta))
        for i in range(2 * m):
            s.set_ub(n + i, INF)

    def update_bounds(self, delta, x_l, x_u, x_k, c_l, c_u, c_k, refresh_ubA=False):  # :342-368 (ubA is NOT refreshed)
        """refresh_ubA=True is NOT the reference's behaviour: its qpOASES branch leaves ubA stale (QPhandler.cpp:358-360),
        which turns its own run infeasible after the first accepted step when a constraint is an equality; the whole-
        trajectory replay (tests/test_sqp_trajectory.py) needs the correct value and says so."""
        s = self.solverInterface_
        n, m = self.nlp_info_.nVar, self.nlp_info_.nCon
        for i in range(m):
            s.set_lbA(i, c_l[i] - c_k[i])
            if refresh_ubA:
                s.set_ubA(i, c_u[i] - c_k[i])
        for i in range(n):
            s.set_lb(i, max(x_l[i] - x_k[i], -delta))
            s.set_ub(i, min(x_u[i] - x_k[i], delta))

    def update_delta(self, delta, x_l, x_u, x_k):  # :533-567
        s = self.solverInterface_
        for i in range(self.nlp_info_.nVar):
            s.set_lb(i, max(x_l[i] - x_k[i], -delta))
            s.set_ub(i, min(x_u[i] - x_k[i], delta))

    def set_g(self, grad, rho):  # :272-297
        s = self.solverInterface_
        for i in range(self.nVar_QP_):
            s.set_g(i, grad[i] if i < self.nlp_info_.nVar else rho)

    def update_penalty(self, rho):  # :430-441
        for i in range(self.nlp_info_.nVar, self.nVar_QP_):
            self.solverInterface_.set_g(i, rho)

    def update_grad(self, grad):  # :450-463
        for i in range(self.nlp_info_.nVar):
            self.solverInterface_.set_g(i, grad[i])

    def set_H(self, hessian):  # :310-318
        self.solverInterface_.set_H(hessian)

    update_H = set_H           # :508-517

    def set_A(self, jacobian):  # :326-334
        self.solverInterface_.set_A(jacobian, self.I_info_A_)

    update_A = set_A           # :520-530

    def solveQP(self, stats=None, options=None):  # :470-499
        self.solverInterface_.optimizeQP(stats)
        if not self.test_optimality():
            raise QP_NOT_OPTIMAL("KKT certificate failed: %g" % self.solverInterface_.get_optimality_status().KKT_error)

    def solveLP(self, stats=None):  # include/sqphot/QPhandler.hpp:66-68
        self.solverInterface_.optimizeLP(stats)

    def test_optimality(self):  # :580-587
        self.qpOptimalStatus_ = self.solverInterface_.get_optimality_status()
        return self.solverInterface_.test_optimality(self.W_c_, self.W_b_)

    def get_optimal_solution(self):
        return self.solverInterface_.get_optimal_solution()

    def get_multipliers_bounds(self):
        return self.solverInterface_.get_multipliers_bounds()

    def get_multipliers_constr(self):
        return self.solverInterface_.get_multipliers_constr()

    def get_objective(self):
        return self.solverInterface_.get_obj_value()

    def get_status(self):
        return self.solverInterface_.get_status()

    def get_infea_measure_model(self):  # :592-594, oneNorm of the slack part
        x = self.solverInterface_.get_optimal_solution()
        return float(np.abs(x[self.nlp_info_.nVar:]).sum())
