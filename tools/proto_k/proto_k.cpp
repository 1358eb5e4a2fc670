// proto_k.cpp -- CPU prototype of the "explicit KKT inverse" formulation (EngineK of qp_small_k.h): the same
// homotopy, ratio tests, tie breaks and drift correction as oracle/qp_oracle.c, but ONE symmetric matrix
// M = K^-1,  K = [H_FR,FR  A_AC,FR' ; A_AC,FR  0]  over the index set S = free variables + active constraints,
// kept current by bordering (an index enters S) and a Schur-complement step (an index leaves S). The step direction
// is one product with M. Cases the formulation does not carry (a removal that would leave Z'HZ not positive
// definite -> flipping bounds, pivots of rounding size, ambiguous independence tests, LPs) make it BAIL: the
// caller re-solves the problem with the null-space engine. Tuning / validation aid for the HIP kernel: compared
// with the oracle by tools/proto_k/check.py. Not product code, not the oracle.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {
const double EPS = 2.221e-16, INFTY = 1e20, EPS_DEN = 1e3 * EPS, BOUND_RELAX = 1e4;
enum { RET_OK = 0, RET_MAX_NWSR = 1, RET_INFEASIBLE = 2, RET_UNBOUNDED = 3, RET_SETUP_FAILED = 4, RET_BAIL = 9 };

struct K {
    int nV, nC, ld;
    std::vector<double> A, H;           // dense: A[i + v*nC], H[u + v*nV]
    std::vector<double> M;              // nS x nS, ld
    std::vector<int> Sb, Sc, posV, posC, skind, sid;   // S entry p: skind 0 var / 1 constraint, sid
    int nS = 0, nFR = 0, nAC = 0;
    std::vector<double> x, y, g, lb, ub, lbA, ubA, Ax, gN, lbN, ubN, lbAN, ubAN, dx, dy, dAx;
    double hscale = 0.0;
    int bail_reason = 0, solved = 0, infeasible = 0;
    double clampinf(double v) { return v > INFTY ? INFTY : (v < -INFTY ? -INFTY : v); }
    static double delta_of(double t, double c) { return (std::fabs(t) >= INFTY && std::fabs(c) >= INFTY) ? 0.0 : t - c; }

    void A_times(const double *v, double *out) { for (int i = 0; i < nC; i++) { double s = 0; for (int c = 0; c < nV; c++) s += A[i + (size_t)c * nC] * v[c]; out[i] = s; } }
    void AT_times(const double *yc, double *out) { for (int c = 0; c < nV; c++) { double s = 0; for (int i = 0; i < nC; i++) s += A[i + (size_t)c * nC] * yc[i]; out[c] = s; } }
    void H_times(const double *v, double *out) { for (int c = 0; c < nV; c++) { double s = 0; for (int u = 0; u < nV; u++) s += H[u + (size_t)c * nV] * v[u]; out[c] = s; } }

    // an index enters S: kvec over the current S, diagonal kappa. false = pivot not acceptable
    // want_pos: +1 the Schur complement must be clearly positive (a variable is freed), -1 clearly negative (a constraint is added)
    bool append(const std::vector<double> &kvec, double kappa, int kind, int id, int want, double scale) {
        std::vector<double> u(nS);
        for (int p = 0; p < nS; p++) { double s = 0; for (int q = 0; q < nS; q++) s += M[p + (size_t)q * ld] * kvec[q]; u[p] = s; }
        double ku = 0; for (int p = 0; p < nS; p++) ku += kvec[p] * u[p];
        const double sigma = kappa - ku;
        if (want > 0 ? !(sigma > 1e-8 * scale) : !(-sigma > 1e-10 * scale)) return false;
        for (int q = 0; q < nS; q++) for (int p = 0; p < nS; p++) M[p + (size_t)q * ld] += u[p] * u[q] / sigma;
        for (int p = 0; p < nS; p++) { M[p + (size_t)nS * ld] = -u[p] / sigma; M[nS + (size_t)p * ld] = -u[p] / sigma; }
        M[nS + (size_t)nS * ld] = 1.0 / sigma;
        skind.push_back(kind); sid.push_back(id);
        if (kind == 0) posV[id] = nS; else posC[id] = nS;
        nS++;
        return true;
    }
    // index p leaves S: M -= m_p m_p' / M_pp, the last entry moves into p
    void erase(int p) {
        const double mu = M[p + (size_t)p * ld];
        std::vector<double> m(nS);
        for (int q = 0; q < nS; q++) m[q] = M[q + (size_t)p * ld];
        for (int q = 0; q < nS; q++) for (int r = 0; r < nS; r++) M[r + (size_t)q * ld] -= m[r] * m[q] / mu;
        const int l = nS - 1;
        if (skind[p] == 0) posV[sid[p]] = -1; else posC[sid[p]] = -1;
        if (p != l) {
            for (int q = 0; q < nS; q++) M[p + (size_t)q * ld] = M[l + (size_t)q * ld];
            for (int q = 0; q < nS; q++) M[q + (size_t)p * ld] = M[q + (size_t)l * ld];
            M[p + (size_t)p * ld] = M[l + (size_t)l * ld];
            skind[p] = skind[l]; sid[p] = sid[l];
            if (skind[p] == 0) posV[sid[p]] = p; else posC[sid[p]] = p;
        }
        skind.pop_back(); sid.pop_back();
        nS--;
    }
    bool free_variable(int v) {          // bound of v leaves the working set
        std::vector<double> k(nS);
        for (int p = 0; p < nS; p++) k[p] = skind[p] == 0 ? H[sid[p] + (size_t)v * nV] : A[sid[p] + (size_t)v * nC];
        if (!append(k, H[v + (size_t)v * nV], 0, v, +1, hscale)) return false;
        Sb[v] = 0; nFR++;
        return true;
    }
    bool add_constraint(int i, int side) {
        std::vector<double> k(nS);
        double na2 = 0;
        for (int p = 0; p < nS; p++) { k[p] = skind[p] == 0 ? A[i + (size_t)sid[p] * nC] : 0.0; na2 += k[p] * k[p]; }
        if (!append(k, 0.0, 1, i, -1, na2 / hscale)) return false;
        Sc[i] = side; nAC++;
        return true;
    }
    // is the removal of index p (a constraint leaves, or a variable gets fixed) well defined?  M_pp must be clearly
    // negative for a constraint (its released direction has positive curvature), clearly positive for a variable
    bool pivot_ok(int p) {
        const double mu = M[p + (size_t)p * ld];
        if (skind[p] == 1) {
            double d2 = 0; for (int q = 0; q < nS; q++) if (skind[q] == 0) d2 += M[q + (size_t)p * ld] * M[q + (size_t)p * ld];
            return -mu > 1e-8 * hscale * d2 && d2 > 0;
        }
        return mu > 1e-10 / hscale;
    }
    void remove_constraint(int i) { erase(posC[i]); Sc[i] = 0; nAC--; }
    void fix_variable(int v, int side) { erase(posV[v]); Sb[v] = side; nFR--; }

    int setup_cold() {
        for (int v = 0; v < nV; v++) {
            int s = -1;
            if (lbN[v] <= -INFTY) s = ubN[v] < INFTY ? 1 : 0;
            Sb[v] = s; x[v] = 0; posV[v] = -1;
        }
        for (int i = 0; i < nC; i++) { Sc[i] = 0; posC[i] = -1; }
        for (int i = 0; i < nV + nC; i++) y[i] = 0;
        nS = nFR = nAC = 0; skind.clear(); sid.clear();
        for (int v = 0; v < nV; v++) if (Sb[v] == 0) { Sb[v] = -1; if (!free_variable(v)) return RET_BAIL; }
        A_times(x.data(), Ax.data());
        std::vector<double> t1(nV), t2(nV);
        AT_times(y.data() + nV, t1.data()); H_times(x.data(), t2.data());
        for (int v = 0; v < nV; v++) {
            g[v] = t1[v] + y[v] - t2[v];
            lb[v] = Sb[v] == -1 ? x[v] : std::fmin(lbN[v], x[v] - BOUND_RELAX);
            ub[v] = Sb[v] == 1 ? x[v] : std::fmax(ubN[v], x[v] + BOUND_RELAX);
        }
        for (int i = 0; i < nC; i++) { lbA[i] = std::fmin(lbAN[i], Ax[i] - BOUND_RELAX); ubA[i] = std::fmax(ubAN[i], Ax[i] + BOUND_RELAX); }
        return RET_OK;
    }

    void step_direction() {
        std::vector<double> pA(nC), pH(nV), r(nS), s(nS), dg(nV), Hdx(nV), ATdy(nV);
        for (int v = 0; v < nV; v++) {
            dx[v] = Sb[v] == -1 ? delta_of(lbN[v], lb[v]) : (Sb[v] == 1 ? delta_of(ubN[v], ub[v]) : 0.0);
            dg[v] = gN[v] - g[v];
        }
        for (int i = 0; i < nV + nC; i++) dy[i] = 0;
        A_times(dx.data(), pA.data()); H_times(dx.data(), pH.data());
        for (int p = 0; p < nS; p++) {
            if (skind[p] == 0) r[p] = -(dg[sid[p]] + pH[sid[p]]);
            else { const int i = sid[p]; r[p] = (Sc[i] == -1 ? delta_of(lbAN[i], lbA[i]) : delta_of(ubAN[i], ubA[i])) - pA[i]; }
        }
        for (int p = 0; p < nS; p++) { double t = 0; for (int q = 0; q < nS; q++) t += M[p + (size_t)q * ld] * r[q]; s[p] = t; }
        for (int p = 0; p < nS; p++) { if (skind[p] == 0) dx[sid[p]] = s[p]; else dy[nV + sid[p]] = -s[p]; }
        H_times(dx.data(), Hdx.data()); AT_times(dy.data() + nV, ATdy.data()); A_times(dx.data(), dAx.data());
        for (int v = 0; v < nV; v++) dy[v] = Sb[v] != 0 ? Hdx[v] + dg[v] - ATdy[v] : 0.0;
    }

    struct Blk { double tau; int kind, idx, side; };
    static void ratio(double num, double den, double &tau, int &hit) {
        hit = 0;
        if (den >= EPS_DEN) { double t = (num > 0.0 ? num : 0.0) / den; if (t < tau) { tau = t; hit = 1; } }
    }
    Blk ratio_tests() {
        Blk b = {1.0, 0, -1, 0}; int hit;
        for (int i = 0; i < nC; i++) { if (Sc[i] == 0) continue; double yi = y[nV + i], d = dy[nV + i];
            if (Sc[i] == -1) ratio(yi, -d, b.tau, hit); else ratio(-yi, d, b.tau, hit); if (hit) { b.kind = 1; b.idx = i; b.side = 0; } }
        for (int v = 0; v < nV; v++) { if (Sb[v] == 0) continue; double yi = y[v], d = dy[v];
            if (Sb[v] == -1) ratio(yi, -d, b.tau, hit); else ratio(-yi, d, b.tau, hit); if (hit) { b.kind = 2; b.idx = v; b.side = 0; } }
        for (int i = 0; i < nC; i++) { if (Sc[i] != 0 || lbAN[i] <= -INFTY) continue;
            ratio(Ax[i] - lbA[i], delta_of(lbAN[i], lbA[i]) - dAx[i], b.tau, hit); if (hit) { b.kind = 3; b.idx = i; b.side = -1; } }
        for (int i = 0; i < nC; i++) { if (Sc[i] != 0 || ubAN[i] >= INFTY) continue;
            ratio(ubA[i] - Ax[i], dAx[i] - delta_of(ubAN[i], ubA[i]), b.tau, hit); if (hit) { b.kind = 3; b.idx = i; b.side = 1; } }
        for (int v = 0; v < nV; v++) { if (Sb[v] != 0 || lbN[v] <= -INFTY) continue;
            ratio(x[v] - lb[v], delta_of(lbN[v], lb[v]) - dx[v], b.tau, hit); if (hit) { b.kind = 4; b.idx = v; b.side = -1; } }
        for (int v = 0; v < nV; v++) { if (Sb[v] != 0 || ubN[v] >= INFTY) continue;
            ratio(ub[v] - x[v], dx[v] - delta_of(ubN[v], ub[v]), b.tau, hit); if (hit) { b.kind = 4; b.idx = v; b.side = 1; } }
        return b;
    }

    // independence of the incoming row a from the working set, from the residual r = a_FR - A_AC,FR' xi_C of its
    // representation by the active rows (xi_C = D'a = the constraint part of M [a; 0]): K M = I gives r = H P a, and
    // Z'r = Z'a, so |Z'a| <= |r| <= cond(Z'HZ) |Z'a| -- a first-order quantity (the oracle tests |Z'a| > 1e-9 |a|), where
    // a'Pa is of second order and drowns in rounding below ~1e-7. 1 independent, 0 dependent, -1 cannot tell (bail)
    int li_test(double rn2, double na2) {
        if (nFR - nAC <= 0) return 0;
        if (!(na2 > 0.0)) return 0;
        const double rel = std::sqrt(rn2 / na2);
        if (rel > 1e-7) return 1;
        if (rel < 1e-9) return 0;
        return -1;
    }

    int change_active_set(const Blk &b) {
        if (b.kind == 1) { if (!pivot_ok(posC[b.idx])) { bail_reason = 1; return RET_BAIL; } remove_constraint(b.idx); y[nV + b.idx] = 0; return RET_OK; }
        if (b.kind == 2) { y[b.idx] = 0; if (!free_variable(b.idx)) { bail_reason = 2; return RET_BAIL; } return RET_OK; }
        // incoming row
        std::vector<double> k(nS), u(nS), afull(nV, 0.0);
        double na2 = 0, aPa = 0;
        if (b.kind == 3) { for (int v = 0; v < nV; v++) afull[v] = A[b.idx + (size_t)v * nC]; for (int p = 0; p < nS; p++) { k[p] = skind[p] == 0 ? afull[sid[p]] : 0.0; na2 += k[p] * k[p]; } }
        else { afull[b.idx] = 1.0; k[posV[b.idx]] = 1.0; na2 = 1.0; }
        for (int p = 0; p < nS; p++) { double t = 0; for (int q = 0; q < nS; q++) t += M[p + (size_t)q * ld] * k[q]; u[p] = t; }
        (void)aPa;
        std::vector<double> xiC(nC, 0.0), xiB(nV, 0.0), t(nV);
        for (int p = 0; p < nS; p++) if (skind[p] == 1) xiC[sid[p]] = u[p];
        AT_times(xiC.data(), t.data());
        double rn2 = 0;
        for (int v = 0; v < nV; v++) if (Sb[v] == 0) { const double r = afull[v] - t[v]; rn2 += r * r; }
        const int li = li_test(rn2, na2);
        if (li < 0) { bail_reason = 3; return RET_BAIL; }
        double ynew = 0.0;
        if (li == 0) {
            // exchange: c = sum xiC_j a_j (active constraints) + sum xiB_v e_v (fixed variables)
            for (int v = 0; v < nV; v++) xiB[v] = Sb[v] != 0 ? afull[v] - t[v] : 0.0;
            const double sgn = b.side == 1 ? -1.0 : 1.0;
            double tmin = INFTY; int pk = 0, pi = -1;
            for (int i = 0; i < nC; i++) { if (Sc[i] == 0) continue; double xi = sgn * xiC[i], yi = y[nV + i];
                double num = Sc[i] == -1 ? yi : -yi, den = Sc[i] == -1 ? xi : -xi;
                if (den > EPS_DEN) { double tt = (num > 0.0 ? num : 0.0) / den; if (tt < tmin) { tmin = tt; pk = 1; pi = i; } } }
            for (int v = 0; v < nV; v++) { if (Sb[v] == 0) continue; double xi = sgn * xiB[v], yi = y[v];
                double num = Sb[v] == -1 ? yi : -yi, den = Sb[v] == -1 ? xi : -xi;
                if (den > EPS_DEN) { double tt = (num > 0.0 ? num : 0.0) / den; if (tt < tmin) { tmin = tt; pk = 2; pi = v; } } }
            if (pk == 0) return RET_INFEASIBLE;
            for (int i = 0; i < nC; i++) if (Sc[i] != 0) y[nV + i] -= tmin * sgn * xiC[i];
            for (int v = 0; v < nV; v++) if (Sb[v] != 0) y[v] -= tmin * sgn * xiB[v];
            ynew = sgn * tmin;
            if (pk == 1) { if (!pivot_ok(posC[pi])) { bail_reason = 4; return RET_BAIL; } remove_constraint(pi); y[nV + pi] = 0; }
            else { y[pi] = 0; if (!free_variable(pi)) { bail_reason = 5; return RET_BAIL; } }
        }
        if (b.kind == 3) { if (!add_constraint(b.idx, b.side)) { bail_reason = 6; return RET_BAIL; } y[nV + b.idx] = ynew; }
        else { if (!pivot_ok(posV[b.idx])) { bail_reason = 7; return RET_BAIL; } fix_variable(b.idx, b.side); y[b.idx] = ynew; }
        return RET_OK;
    }

    void drift_correction() {
        std::vector<double> t1(nV), t2(nV);
        for (int v = 0; v < nV; v++) if (Sb[v] != 0) x[v] = Sb[v] == -1 ? lb[v] : ub[v];
        A_times(x.data(), Ax.data());
        for (int i = 0; i < nC; i++) { if (Sc[i] == -1) lbA[i] = Ax[i]; else if (Sc[i] == 1) ubA[i] = Ax[i]; }
        AT_times(y.data() + nV, t1.data()); H_times(x.data(), t2.data());
        for (int v = 0; v < nV; v++) g[v] = t1[v] + y[v] - t2[v];
    }

    int homotopy(int maxit, int &nWSR) {
        int iter = 0, rc = RET_OK;
        for (;;) {
            step_direction();
            Blk b = ratio_tests();
            const double tau = b.tau;
            for (int v = 0; v < nV; v++) { x[v] += tau * dx[v]; g[v] += tau * (gN[v] - g[v]); const double dl = delta_of(lbN[v], lb[v]), du = delta_of(ubN[v], ub[v]); lb[v] += tau * dl; ub[v] += tau * du; }
            for (int i = 0; i < nV + nC; i++) y[i] += tau * dy[i];
            for (int i = 0; i < nC; i++) { const double dl = delta_of(lbAN[i], lbA[i]), du = delta_of(ubAN[i], ubA[i]); lbA[i] += tau * dl; ubA[i] += tau * du; }
            if (b.kind == 0) {
                for (int v = 0; v < nV; v++) { g[v] = gN[v]; lb[v] = lbN[v]; ub[v] = ubN[v]; if (Sb[v] != 0) x[v] = Sb[v] == -1 ? lb[v] : ub[v]; }
                for (int i = 0; i < nC; i++) { lbA[i] = lbAN[i]; ubA[i] = ubAN[i]; }
                A_times(x.data(), Ax.data());
                solved = 1; break;
            }
            if (iter >= maxit) { A_times(x.data(), Ax.data()); rc = RET_MAX_NWSR; break; }
            A_times(x.data(), Ax.data());
            if (b.kind == 3) { if (b.side == -1) lbA[b.idx] = Ax[b.idx]; else ubA[b.idx] = Ax[b.idx]; }
            else if (b.kind == 4) { if (b.side == -1) lb[b.idx] = x[b.idx]; else ub[b.idx] = x[b.idx]; }
            rc = change_active_set(b);
            if (rc == RET_INFEASIBLE) { infeasible = 1; break; }
            if (rc != RET_OK) break;
            iter++;
            drift_correction();
        }
        nWSR = iter;
        return rc;
    }
};
}  // namespace

// cold solve of a dense-data QP. Returns RET_*; RET_BAIL (9) = hand the problem to the null-space engine
extern "C" int protok_solve(int nV, int nC, const double *A, const double *H, const double *g, const double *lb, const double *ub,
                            const double *lbA, const double *ubA, int maxit, double *x, double *y, int *Sb, int *Sc, int *nWSR,
                            int *info) {
    K e;
    e.nV = nV; e.nC = nC; e.ld = nV + (nV < nC ? nV : nC) + 1;
    e.A.assign(A, A + (size_t)nC * nV); e.H.assign(H, H + (size_t)nV * nV);
    e.M.assign((size_t)e.ld * e.ld, 0.0);
    e.Sb.assign(nV, 0); e.Sc.assign(nC, 0); e.posV.assign(nV, -1); e.posC.assign(nC, -1);
    for (auto *v : {&e.x, &e.g, &e.lb, &e.ub, &e.gN, &e.lbN, &e.ubN, &e.dx}) v->assign(nV, 0.0);
    for (auto *v : {&e.lbA, &e.ubA, &e.Ax, &e.lbAN, &e.ubAN, &e.dAx}) v->assign(nC, 0.0);
    e.y.assign(nV + nC, 0.0); e.dy.assign(nV + nC, 0.0);
    for (int v = 0; v < nV; v++) { e.gN[v] = g[v]; e.lbN[v] = e.clampinf(lb[v]); e.ubN[v] = e.clampinf(ub[v]); e.hscale = std::fmax(e.hscale, std::fabs(H[v + (size_t)v * nV])); }
    for (int i = 0; i < nC; i++) { e.lbAN[i] = e.clampinf(lbA[i]); e.ubAN[i] = e.clampinf(ubA[i]); }
    info[0] = info[1] = 0;
    if (!(e.hscale > 0.0)) { info[0] = 10; return RET_BAIL; }
    for (int v = 0; v < nV; v++) if (e.lbN[v] > e.ubN[v] + EPS) { info[1] = 1; *nWSR = 0; return RET_INFEASIBLE; }
    for (int i = 0; i < nC; i++) if (e.lbAN[i] > e.ubAN[i] + EPS) { info[1] = 1; *nWSR = 0; return RET_INFEASIBLE; }
    int rc = e.setup_cold();
    if (rc == RET_OK) rc = e.homotopy(maxit, *nWSR);
    info[0] = e.bail_reason; info[1] = e.infeasible; info[2] = e.solved;
    std::memcpy(x, e.x.data(), 8 * nV); std::memcpy(y, e.y.data(), 8 * (nV + nC));
    std::memcpy(Sb, e.Sb.data(), 4 * nV); if (nC) std::memcpy(Sc, e.Sc.data(), 4 * nC);
    return rc;
}
