"""Quick GPU-vs-oracle comparison (development aid; the real checks live in tests/)."""
import glob, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as O
from restartsqp_amd import capi
from restartsqp_amd.qpdump import read_qore_dump, dense_to_csc, QPData

def rand_qp(rng, nV, nC, dens=0.5):
    M = rng.normal(size=(nV, nV)); H = M @ M.T / nV + np.eye(nV)
    A = rng.normal(size=(nC, nV)) * (rng.random((nC, nV)) < dens)
    g = rng.normal(size=nV) * 3; xh = rng.normal(size=nV)
    lb = xh - np.abs(rng.normal(size=nV)); ub = xh + np.abs(rng.normal(size=nV))
    lbA = A @ xh - np.abs(rng.normal(size=nC)); ubA = A @ xh + np.abs(rng.normal(size=nC))
    Hc = dense_to_csc(H); Ac = dense_to_csc(A)
    return QPData(nV, nC, *Hc, *Ac, g, lb, ub, lbA, ubA)

def oracle_solve(q, nWSR=1000):
    qp = O.OracleQP(q.nV, q.nC); qp.set_A_csc(q.A_jc, q.A_ir, q.A_val); qp.set_H_csc(q.H_jc, q.H_ir, q.H_val)
    rc, n = qp.init(q.g, q.lb, q.ub, q.lbA, q.ubA, nWSR)
    return qp, rc, n

def main():
    print("devices:", capi.device_count(), capi.lib().rsqp_version().decode())
    rng = np.random.default_rng(5)
    probs = [rand_qp(rng, int(rng.integers(2, 40)), int(rng.integers(0, 50))) for _ in range(64)]
    probs += [read_qore_dump(p) for p in sorted(glob.glob(os.path.join(ROOT, "tests/golden/qore_dumps/*.log")))]
    b = capi.Batch(probs)
    t = time.time(); b.solve(capi.MODE_COLD, 1000); t = time.time() - t
    res = b.results(); ok, kkt = b.test_optimality()
    print("batch of %d solved in %.3f ms host, %.3f ms device" % (len(probs), t * 1e3, b.last_solve_ms()))
    bad = 0
    for k, (q, r) in enumerate(zip(probs, res)):
        qp, rc, n = oracle_solve(q)
        dx = np.abs(qp.x - r["x"]).max(); dy = np.abs(qp.y - r["y"]).max()
        same_ws = np.array_equal(qp.ws_bounds, r["ws_b"]) and np.array_equal(qp.ws_constraints, r["ws_c"])
        scale = max(1.0, np.abs(qp.x).max()); yscale = max(1.0, np.abs(qp.y).max())
        good = same_ws and n == r["nWSR"] and r["status"] == qp.exitflag() and dx <= 1e-8 * scale and dy <= 1e-8 * yscale
        if not good:
            bad += 1
        if not good or k >= 64:
            print("%3d %-26s nV=%3d nC=%3d nWSR gpu/orc=%3d/%3d status=%d/%d ws_equal=%s dx=%.2e dy=%.2e kkt=%.2e ok=%d" % (
                k, q.name, q.nV, q.nC, r["nWSR"], n, r["status"], qp.exitflag(), same_ws, dx, dy, kkt[k], ok[k]))
    print("mismatches: %d of %d" % (bad, len(probs)))
    # single-solver path + certificate + products
    q = probs[3]
    s = capi.Solver(q.nV, q.nC)
    s.set_A_csc(q.A_jc, q.A_ir, q.A_val); s.set_H_csc(q.H_jc, q.H_ir, q.H_val)
    for w, v in ((capi.VEC_G, q.g), (capi.VEC_LB, q.lb), (capi.VEC_UB, q.ub), (capi.VEC_LBA, q.lbA), (capi.VEC_UBA, q.ubA)):
        s.set_vector(w, v)
    n = s.optimize_qp()
    okk, st, Wc, Wb = s.test_optimality()
    qp, rc, n2 = oracle_solve(q)
    print("single: nWSR", n, n2, "status", s.status, "kkt", st.KKT_error, okk, "dx", np.abs(s.x - qp.x).max())
    v = rng.normal(size=q.nV)
    print("A*v err", np.abs(s.A_times(v) - q.dense_A() @ v).max(), "H*v err", np.abs(s.H_times(v) - q.dense_H() @ v).max())
    return bad

if __name__ == "__main__":
    sys.exit(1 if main() else 0)
