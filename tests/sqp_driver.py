"""Minimal trust-region S-l1-QP driver -- TEST INFRASTRUCTURE, not a re-implementation of the reference's Algorithm.cpp.

It walks an analytic NLP (restartsqp_amd.problems.hs071_nlp / hs065_nlp) from its starting point to a KKT point with the
reference's loop structure and default options, so that the QP boundary sees a WHOLE SQP trajectory: the same sequence of
set / update / solve calls that Algorithm::setupQP issues (reference src/Algorithm.cpp:645-697), with the dirty-flag
pattern of ratio_test (:722-797: an accepted step refreshes A, H, bounds and g) and update_radius (:820-866: a rejected
or very successful step changes only delta), trial point and merit as get_trial_point_info / ratio_test (:414-437,
:722-731), infeasibility measure as cal_infea (:577-602). Simplified on purpose: the penalty update multiplies rho by
increase_parm while the QP model stays infeasible (the reference also solves an LP there, :886-1028), no second-order
correction (off by default, Options.cpp:26), convergence = KKT residual of the NLP with the QP multipliers.

One deviation is forced: QPhandler::update_bounds never refreshes ubA on the qpOASES branch (reference
src/QPhandler.cpp:358-360); with an equality constraint the reference's own run turns infeasible after the first accepted
step (tests/test_gpu_parity.py::test_stale_ubA_quirk_reports_infeasible). The driver hands the QP the CORRECT ubA.

A backend is anything with  solve(qp: QPData, flags: dict) -> dict(x, y, obj, nWSR, status)."""
import numpy as np

from restartsqp_amd import problems

OPT = dict(eta_c=0.25, eta_s=1.0e-8, eta_e=0.75, gamma_c=0.5, gamma_e=2.0, delta=1.0, delta_min=1.0e-16, delta_max=1.0e8,
           tol=1.0e-8, rho=1.0, increase_parm=10.0, rho_max=1.0e6, iter_max=100, opt_tol=1.0e-6)   # reference src/Options.cpp:20-52


def cal_infea(c, c_l, c_u):
    return float(np.sum(np.maximum(c_l - c, 0.0)) + np.sum(np.maximum(c - c_u, 0.0)))


def kkt_residual(nlp, lam_c, lam_b):
    """stationarity + complementarity-free feasibility of the NLP at nlp['x'] with the QP's multipliers"""
    n = nlp["info"].nVar
    J = np.zeros((nlp["info"].nCon, n))
    for r, c, v in zip(nlp["J"].RowIndex, nlp["J"].ColIndex, nlp["J"].MatVal):
        J[r - 1, c - 1] = v
    stat = np.abs(nlp["grad"] - J.T @ lam_c - lam_b).sum()
    return stat + cal_infea(nlp["c"], nlp["c_l"], nlp["c_u"])


def run_sqp(nlp_fn, backend, name="hs071", record=None):
    """Returns (x, f, iterations, trace). trace = one entry per QP solve: the iterate the QP was built at, delta, rho, the
    dirty flags handed to the boundary and the backend's answer."""
    o = OPT
    nlp = nlp_fn(None, None)
    n, m = nlp["info"].nVar, nlp["info"].nCon
    delta, rho = o["delta"], o["rho"]
    lam_c = np.zeros(m)
    flags = dict(first=True, A=False, H=False, bounds=False, delta=False, penalty=False, g=False)
    trace = []
    infea = cal_infea(nlp["c"], nlp["c_l"], nlp["c_u"])
    it = 0
    while it < o["iter_max"]:
        qp = problems.handler_qp(nlp, delta=delta, rho=rho, name="%s_sqp_it%d" % (name, it))
        ans = backend.solve(qp, dict(flags))
        trace.append(dict(it=it, x=nlp["x"].tolist(), lam=lam_c.tolist(), delta=delta, rho=rho, flags=dict(flags), **{k: ans[k] for k in ("nWSR", "status")},
                          x_qp=np.asarray(ans["x"]).tolist(), y_qp=np.asarray(ans["y"]).tolist()))
        flags = dict(first=False, A=False, H=False, bounds=False, delta=False, penalty=False, g=False)
        if ans["status"] != 20:
            break
        xq, yq = np.asarray(ans["x"]), np.asarray(ans["y"])
        p = xq[:n]
        infea_model = float(np.abs(xq[n:]).sum())                # QPhandler::get_infea_measure_model
        if infea_model > o["tol"] and rho < o["rho_max"] and infea > o["tol"]:
            # penalty update (simplified): a larger rho while the linearised constraints cannot be met
            rho_new = min(rho * o["increase_parm"], o["rho_max"])
            qp2 = problems.handler_qp(nlp, delta=delta, rho=rho_new, name="%s_sqp_it%d_rho" % (name, it))
            ans2 = backend.solve(qp2, dict(first=False, A=False, H=False, bounds=False, delta=False, penalty=True, g=False))
            trace.append(dict(it=it, x=nlp["x"].tolist(), lam=lam_c.tolist(), delta=delta, rho=rho_new,
                              flags=dict(first=False, A=False, H=False, bounds=False, delta=False, penalty=True, g=False),
                              nWSR=ans2["nWSR"], status=ans2["status"], x_qp=np.asarray(ans2["x"]).tolist(), y_qp=np.asarray(ans2["y"]).tolist()))
            if ans2["status"] == 20 and float(np.abs(np.asarray(ans2["x"])[n:]).sum()) < infea_model - 1e-12:
                rho, ans, xq, yq = rho_new, ans2, np.asarray(ans2["x"]), np.asarray(ans2["y"])
                p = xq[:n]
            else:                                               # no progress: keep rho, hand the old value back
                flags["penalty"] = True
        qp_obj = float(ans["obj"])
        trial = nlp_fn(nlp["x"] + p, lam_c)
        infea_t = cal_infea(trial["c"], trial["c_l"], trial["c_u"])
        actual = (nlp["f"] + rho * infea) - (trial["f"] + rho * infea_t)
        pred = rho * infea - qp_obj
        it += 1
        if actual >= o["eta_s"] * pred and actual >= -o["tol"]:
            nV = n + 2 * m
            lam_c, lam_b = yq[nV:nV + m].copy(), yq[:n].copy()
            nlp = nlp_fn(nlp["x"] + p, lam_c)                   # accepted: new gradient, Jacobian, Hessian of the Lagrangian
            infea = infea_t
            flags.update(A=True, H=True, bounds=True, g=True)
            if kkt_residual(nlp, lam_c, lam_b) < o["opt_tol"] and np.abs(p).max() < 1e-6:
                break
        if actual < o["eta_c"] * pred:
            delta *= o["gamma_c"]; flags["delta"] = True
        elif actual > o["eta_e"] * pred and o["tol"] > abs(delta - np.abs(p).max()):
            delta = min(o["gamma_e"] * delta, o["delta_max"]); flags["delta"] = True
        if delta < o["delta_min"]:
            break
        if not any(flags[k] for k in ("A", "H", "bounds", "delta", "penalty", "g")):
            break                                               # "QP is not changed" (Algorithm.cpp:651-668)
    return nlp["x"], nlp["f"], it, trace


class OracleBackend:
    """the CPU oracle behind the restated optimizeQP dispatch (oracle.OracleInterface)"""

    def __init__(self, O, from_y0=False):
        self.O, self.oi, self.from_y0 = O, None, from_y0

    def solve(self, qp, flags):
        if self.oi is None:
            self.oi = self.O.OracleInterface(qp.nV, qp.nC, qp_maxiter=1000, from_y0=self.from_y0)
            self.oi.set_A_csc(qp.A_jc, qp.A_ir, qp.A_val); self.oi.set_H_csc(qp.H_jc, qp.H_ir, qp.H_val)
        else:
            if flags["A"]:
                self.oi.set_A_csc(qp.A_jc, qp.A_ir, qp.A_val)
            if flags["H"]:
                self.oi.set_H_csc(qp.H_jc, qp.H_ir, qp.H_val)
        n = self.oi.optimize_qp(qp.g, qp.lb, qp.ub, qp.lbA, qp.ubA)
        q = self.oi.qp
        return dict(x=q.x, y=q.y, obj=q.objective, nWSR=n, status=q.exitflag(), mode=self.oi.modes[-1],
                    ws_b=q.ws_bounds, ws_c=q.ws_constraints)
