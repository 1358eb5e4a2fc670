// proto_g.cpp -- CPU prototype of the TABLEAU formulation (EngineG of qp_small_g.h, round 4): the homotopy, ratio tests, tie
// breaks and drift correction of oracle/qp_oracle.c, with the whole linear algebra of the active-set method in ONE symmetric
// (nV + nC) x (nV + nC) matrix in FIXED slots (slot of variable v = v, of constraint i = nV + i),
//      G = - SWEEP_S(K),    K = [H A'; A 0],    S = free variables + active constraints:
//      G_SS = K_SS^-1 (the "M" of qp_small_k.h),  G_SN = -M K_SN,  G_NN = -(K_NN - K_NS M K_SN)   (N = the other slots),
// i.e. the relation  [z_S; -r_N] = G [r_S; z_N]  for K z = r. Consequences:
//   * step direction of ALL quantities = ONE product: in = (-dg on free variables, bound moves on fixed ones, limit moves on active
//     constraints, 0 on inactive ones) -> out = (dx_FR, -(H dx - A'dy) on fixed variables, -dy on active constraints, -A dx on
//     inactive ones). No carried A dx_FX / H dx_FX, no second product.
//   * a working-set change = ONE principal pivot on slot q: with g = column q, pi = G_qq:  G <- G0 - (1/pi) u u', G0 = G with row and
//     column q zeroed, u = g except u_q = +1 (q enters S) / -1 (q leaves S). The column is READ from the tableau: no product u = M k.
//   * an exchange (incoming row dependent on the working set) = ONE 2 x 2 block pivot on (partner, incoming): defined whenever the
//     exchange is, also when the partner alone would leave Z'HZ singular (hs071-like QPs: zero curvature on the slacks) -- the case
//     qp_small_k.h bailed out of. The dependency coefficients (xi of the active constraints AND of the fixed variables) are the pivot
//     column itself.
//   * a removal that would leave Z'HZ not positive definite = a flip to the opposite side ("flipping bounds"): no change of G at all.
// Validated against the oracle by tools/proto_k/check.py (libprotog). Tuning / validation aid, not product code, not the oracle.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {
const double EPS = 2.221e-16, INFTY = 1e20, EPS_DEN = 1e3 * EPS, BOUND_RELAX = 1e4;
enum { RET_OK = 0, RET_MAX_NWSR = 1, RET_INFEASIBLE = 2, RET_UNBOUNDED = 3, RET_SETUP_FAILED = 4, RET_BAIL = 9 };
const int REFRESH = 8;

struct G_ {
    int nV, nC, N;
    std::vector<double> A, H;           // dense: A[i + v*nC], H[u + v*nV]
    std::vector<double> G;              // N x N, column major
    std::vector<int> Sb, Sc;
    int nFR = 0, nAC = 0, nflips = 0;
    std::vector<double> x, yB, yC, g, lb, ub, lbA, ubA, Ax, gN, lbN, ubN, lbAN, ubAN, gy;
    std::vector<double> in, out, dx, dyB, dyC, dAx, hdv;
    double hscale = 0.0;
    int bail_reason = 0, solved = 0, infeasible = 0, unbounded = 0, since_refresh = REFRESH;
    int quick_li = 1;                   // 0: always the residual test of qp_small_k.h (A'xi product)
    long long stat_slow_li = 0, stat_exch = 0, stat_flip = 0;
    double clampinf(double v) { return v > INFTY ? INFTY : (v < -INFTY ? -INFTY : v); }
    double &g_(int i, int j) { return G[i + (size_t)j * N]; }

    void A_times(const double *v, double *o) { for (int i = 0; i < nC; i++) { double s = 0; for (int c = 0; c < nV; c++) s += A[i + (size_t)c * nC] * v[c]; o[i] = s; } }
    void AT_times(const double *yc, double *o) { for (int c = 0; c < nV; c++) { double s = 0; for (int i = 0; i < nC; i++) s += A[i + (size_t)c * nC] * yc[i]; o[c] = s; } }
    void H_times(const double *v, double *o) { for (int c = 0; c < nV; c++) { double s = 0; for (int u = 0; u < nV; u++) s += H[u + (size_t)c * nV] * v[u]; o[c] = s; } }
    bool inS(int q) const { return q < nV ? Sb[q] == 0 : Sc[q - nV] != 0; }

    // principal pivot on slot q (enters S if it is outside, leaves if inside)
    void pivot(int q) {
        const bool leaving = inS(q);
        std::vector<double> u(N);
        for (int i = 0; i < N; i++) u[i] = g_(i, q);
        const double pi = u[q];
        u[q] = leaving ? -1.0 : 1.0;
        const double c = -1.0 / pi;
        for (int j = 0; j < N; j++) for (int i = 0; i < N; i++) {
            const double base = (i == q || j == q) ? 0.0 : g_(i, j);
            g_(i, j) = base + (c * u[i]) * u[j];
        }
    }
    // 2 x 2 block pivot on (p, q); false = the block is singular to rounding
    bool pivot2(int p, int q) {
        const double sp = inS(p) ? -1.0 : 1.0, sq = inS(q) ? -1.0 : 1.0;
        std::vector<double> up(N), uq(N);
        for (int i = 0; i < N; i++) { up[i] = g_(i, p); uq[i] = g_(i, q); }
        const double pp = up[p], qq = uq[q], pq = up[q];
        const double det = pp * qq - pq * pq;
        if (!(det < 0.0) || !(-det > 1e-10 * std::fmax(std::fabs(pp * qq), pq * pq))) return false;
        const double w11 = qq / det, w12 = -pq / det, w22 = pp / det;
        up[p] = sp; up[q] = 0.0; uq[p] = 0.0; uq[q] = sq;
        for (int j = 0; j < N; j++) {
            const double cp = w11 * up[j] + w12 * uq[j], cq = w12 * up[j] + w22 * uq[j];
            for (int i = 0; i < N; i++) {
                const double base = (i == p || i == q || j == p || j == q) ? 0.0 : g_(i, j);
                g_(i, j) = base - (up[i] * cp + uq[i] * cq);
            }
        }
        return true;
    }

    int setup_cold() {
        for (int j = 0; j < N; j++) for (int i = 0; i < N; i++) {
            double k;
            if (i < nV && j < nV) k = H[i + (size_t)j * nV];
            else if (i >= nV && j < nV) k = A[(i - nV) + (size_t)j * nC];
            else if (i < nV && j >= nV) k = A[(j - nV) + (size_t)i * nC];
            else k = 0.0;
            g_(i, j) = -k;
        }
        nFR = nAC = 0;
        for (int v = 0; v < nV; v++) { Sb[v] = lbN[v] > -INFTY ? -1 : (ubN[v] < INFTY ? 1 : -2); x[v] = 0.0; yB[v] = 0.0; }
        for (int i = 0; i < nC; i++) { Sc[i] = 0; yC[i] = 0.0; Ax[i] = 0.0; }
        // variables without a finite bound are free from the start: one pivot each (Z'HZ over them has to be positive definite)
        for (int v = 0; v < nV; v++) if (Sb[v] == -2) {
            Sb[v] = -1;     // (outside S for the pivot)
            if (!(-g_(v, v) > 1e-8 * hscale)) { bail_reason = 11; return RET_BAIL; }
            pivot(v); Sb[v] = 0; nFR++;
        }
        for (int v = 0; v < nV; v++) {
            const int s = Sb[v];
            lb[v] = s == -1 ? 0.0 : std::fmin(lbN[v], -BOUND_RELAX);
            ub[v] = s == 1 ? 0.0 : std::fmax(ubN[v], BOUND_RELAX);
            g[v] = 0.0; gy[v] = 0.0;
        }
        for (int i = 0; i < nC; i++) { lbA[i] = std::fmin(lbAN[i], -BOUND_RELAX); ubA[i] = std::fmax(ubAN[i], BOUND_RELAX); }
        return RET_OK;
    }

    void refresh_exact() {
        std::vector<double> t1(nV), t2(nV);
        A_times(x.data(), Ax.data());
        AT_times(yC.data(), t1.data()); H_times(x.data(), t2.data());
        for (int v = 0; v < nV; v++) gy[v] = t1[v] - t2[v];
        since_refresh = 0;
    }
    void step_direction() {
        if (since_refresh >= REFRESH) refresh_exact();
        for (int v = 0; v < nV; v++) {
            g[v] = gy[v] + yB[v];                                  // drift correction: gradient of the current QP from stationarity
            const int s = Sb[v];
            in[v] = s == 0 ? -(gN[v] - g[v]) : (s == -1 ? lbN[v] - lb[v] : ubN[v] - ub[v]);
        }
        for (int i = 0; i < nC; i++) {
            const int s = Sc[i];
            if (s == -1) lbA[i] = Ax[i]; else if (s == 1) ubA[i] = Ax[i];
            in[nV + i] = s == 0 ? 0.0 : (s == -1 ? lbAN[i] - lbA[i] : ubAN[i] - ubA[i]);
        }
        for (int i = 0; i < N; i++) { double s = 0; for (int j = 0; j < N; j++) s += g_(i, j) * in[j]; out[i] = s; }
        for (int v = 0; v < nV; v++) {
            const double dg = gN[v] - g[v];
            if (Sb[v] == 0) { dx[v] = out[v]; dyB[v] = 0.0; hdv[v] = -dg; }
            else { dx[v] = in[v]; dyB[v] = dg - out[v]; hdv[v] = -out[v]; }
        }
        for (int i = 0; i < nC; i++) {
            if (Sc[i] != 0) { dyC[i] = -out[nV + i]; dAx[i] = in[nV + i]; }
            else { dyC[i] = 0.0; dAx[i] = -out[nV + i]; }
        }
    }

    struct Blk { double tau; int kind, idx, side; };
    static void ratio(double num, double den, double &tau, int &hit) {
        hit = 0;
        if (den >= EPS_DEN) { double t = (num > 0.0 ? num : 0.0) / den; if (t < tau) { tau = t; hit = 1; } }
    }
    Blk ratio_tests() {
        Blk b = {1.0, 0, -1, 0}; int hit;
        for (int i = 0; i < nC; i++) { if (Sc[i] == 0) continue; double yi = yC[i], d = dyC[i];
            if (Sc[i] == -1) ratio(yi, -d, b.tau, hit); else ratio(-yi, d, b.tau, hit); if (hit) { b.kind = 1; b.idx = i; b.side = 0; } }
        for (int v = 0; v < nV; v++) { if (Sb[v] == 0) continue; double yi = yB[v], d = dyB[v];
            if (Sb[v] == -1) ratio(yi, -d, b.tau, hit); else ratio(-yi, d, b.tau, hit); if (hit) { b.kind = 2; b.idx = v; b.side = 0; } }
        for (int i = 0; i < nC; i++) { if (Sc[i] != 0 || lbAN[i] <= -INFTY) continue;
            ratio(Ax[i] - lbA[i], (lbAN[i] - lbA[i]) - dAx[i], b.tau, hit); if (hit) { b.kind = 3; b.idx = i; b.side = -1; } }
        for (int i = 0; i < nC; i++) { if (Sc[i] != 0 || ubAN[i] >= INFTY) continue;
            ratio(ubA[i] - Ax[i], dAx[i] - (ubAN[i] - ubA[i]), b.tau, hit); if (hit) { b.kind = 3; b.idx = i; b.side = 1; } }
        for (int v = 0; v < nV; v++) { if (Sb[v] != 0 || lbN[v] <= -INFTY) continue;
            ratio(x[v] - lb[v], (lbN[v] - lb[v]) - dx[v], b.tau, hit); if (hit) { b.kind = 4; b.idx = v; b.side = -1; } }
        for (int v = 0; v < nV; v++) { if (Sb[v] != 0 || ubN[v] >= INFTY) continue;
            ratio(ub[v] - x[v], dx[v] - (ubN[v] - ub[v]), b.tau, hit); if (hit) { b.kind = 4; b.idx = v; b.side = 1; } }
        return b;
    }

    // 1 independent, 0 dependent, -1 cannot tell. q = slot of the incoming row, col = its tableau column
    int li_test(int q, const std::vector<double> &col, const std::vector<double> &afull) {
        if (nFR - nAC <= 0) return 0;
        double na2 = 0, pn2 = 0;
        for (int v = 0; v < nV; v++) if (Sb[v] == 0) { na2 += afull[v] * afull[v]; pn2 += col[v] * col[v]; }
        if (!(na2 > 0.0)) return 0;
        if (quick_li) {
            // |P a| (the free-variable part of the pivot column) is of first order in |Z'a|: |Z'a| / lmax <= |P a| <= |Z'a| / lmin
            const double rel = hscale * std::sqrt(pn2 / na2);
            if (rel > 1e-6) return 1;
            if (rel < 1e-12) return 0;
            stat_slow_li++;
        }
        // the residual of the row's representation by the active rows (qp_small_k.h): r = a_FR - A_AC,FR' xi_C
        const double sg = q < nV ? 1.0 : -1.0;        // xi = +column for a variable that leaves S, -column for a constraint that enters
        std::vector<double> xiC(nC, 0.0), t(nV);
        for (int i = 0; i < nC; i++) if (Sc[i] != 0) xiC[i] = sg * col[nV + i];
        AT_times(xiC.data(), t.data());
        double rn2 = 0;
        for (int v = 0; v < nV; v++) if (Sb[v] == 0) { const double r = afull[v] - t[v]; rn2 += r * r; }
        const double rel = std::sqrt(rn2 / na2);
        if (rel > 1e-7) return 1;
        if (rel < 1e-9) return 0;
        return -1;
    }

    int change_active_set(const Blk &b) {
        if (b.kind == 1) {            // constraint i leaves
            const int i = b.idx, q = nV + i, old = Sc[i];
            const double mu = g_(q, q);
            double d2 = 0; for (int v = 0; v < nV; v++) if (Sb[v] == 0) d2 += g_(v, q) * g_(v, q);
            yC[i] = 0.0;
            if (d2 > 0.0 && -mu > 1e-8 * hscale * d2) { pivot(q); Sc[i] = 0; nAC--; return RET_OK; }
            if (d2 > 0.0 && !(-mu < 1e-11 * hscale * d2)) { bail_reason = 1; return RET_BAIL; }
            // flip: the released direction has no curvature -- the constraint goes to its opposite side, G is unchanged
            if ((old == -1 && ubAN[i] >= INFTY) || (old == 1 && lbAN[i] <= -INFTY)) return RET_UNBOUNDED;
            Sc[i] = -old;
            if (old == -1) ubA[i] = Ax[i]; else lbA[i] = Ax[i];
            nflips++; stat_flip++; since_refresh = REFRESH;
            return RET_OK;
        }
        if (b.kind == 2) {            // bound of v leaves: v enters S
            const int v = b.idx, old = Sb[v];
            const double sigma = -g_(v, v);
            yB[v] = 0.0;
            if (sigma > 1e-8 * hscale) { pivot(v); Sb[v] = 0; nFR++; return RET_OK; }
            if (!(sigma < 1e-11 * hscale)) { bail_reason = 2; return RET_BAIL; }
            if ((old == -1 && ubN[v] >= INFTY) || (old == 1 && lbN[v] <= -INFTY)) return RET_UNBOUNDED;
            Sb[v] = -old;
            if (old == -1) ub[v] = x[v]; else lb[v] = x[v];
            nflips++; stat_flip++; since_refresh = REFRESH;
            return RET_OK;
        }
        // incoming row: constraint i is added (slot enters S) or variable v gets fixed (slot leaves S)
        const int q = b.kind == 3 ? nV + b.idx : b.idx;
        std::vector<double> col(N), afull(nV, 0.0);
        for (int i = 0; i < N; i++) col[i] = g_(i, q);
        if (b.kind == 3) for (int v = 0; v < nV; v++) afull[v] = A[b.idx + (size_t)v * nC];
        else afull[b.idx] = 1.0;
        const int li = li_test(q, col, afull);
        if (li < 0) { bail_reason = 3; return RET_BAIL; }
        double ynew = 0.0;
        if (li == 0) {
            stat_exch++;
            const double sg = q < nV ? 1.0 : -1.0, sgn = b.side == 1 ? -1.0 : 1.0;
            double tmin = INFTY; int pk = 0, pi = -1;
            for (int i = 0; i < nC; i++) { if (Sc[i] == 0) continue; double xi = sgn * sg * col[nV + i], yi = yC[i];
                double num = Sc[i] == -1 ? yi : -yi, den = Sc[i] == -1 ? xi : -xi;
                if (den > EPS_DEN) { double tt = (num > 0.0 ? num : 0.0) / den; if (tt < tmin) { tmin = tt; pk = 1; pi = i; } } }
            for (int v = 0; v < nV; v++) { if (Sb[v] == 0) continue; double xi = sgn * sg * col[v], yi = yB[v];
                double num = Sb[v] == -1 ? yi : -yi, den = Sb[v] == -1 ? xi : -xi;
                if (den > EPS_DEN) { double tt = (num > 0.0 ? num : 0.0) / den; if (tt < tmin) { tmin = tt; pk = 2; pi = v; } } }
            if (pk == 0) { if (getenv("PROTOG_INFEAS_DIRECT")) return RET_INFEASIBLE; bail_reason = 8; return RET_BAIL; }
            for (int i = 0; i < nC; i++) if (Sc[i] != 0) yC[i] -= tmin * sgn * sg * col[nV + i];
            for (int v = 0; v < nV; v++) if (Sb[v] != 0) yB[v] -= tmin * sgn * sg * col[v];
            ynew = sgn * tmin;
            const int p = pk == 1 ? nV + pi : pi;
            if (!pivot2(p, q)) { bail_reason = 5; return RET_BAIL; }
            if (pk == 1) { Sc[pi] = 0; yC[pi] = 0.0; nAC--; } else { Sb[pi] = 0; yB[pi] = 0.0; nFR++; }
            since_refresh = REFRESH;
        } else {
            const double pi_ = col[q];
            if (b.kind == 3) {
                double na2 = 0; for (int v = 0; v < nV; v++) if (Sb[v] == 0) na2 += afull[v] * afull[v];
                if (!(pi_ > 1e-10 * na2 / hscale)) { bail_reason = 6; return RET_BAIL; }
            } else if (!(pi_ > 1e-10 / hscale)) { bail_reason = 7; return RET_BAIL; }
            pivot(q);
        }
        if (b.kind == 3) { Sc[b.idx] = b.side; yC[b.idx] = ynew; nAC++; }
        else { Sb[b.idx] = b.side; yB[b.idx] = ynew; nFR--; }
        return RET_OK;
    }

    int homotopy(int maxit, int &nWSR) {
        int iter = 0, rc = RET_OK;
        since_refresh = REFRESH;
        for (;;) {
            step_direction();
            Blk b = ratio_tests();
            const double tau = b.tau;
            const bool done = b.kind == 0;
            for (int v = 0; v < nV; v++) {
                const int s = Sb[v];
                if (done) { g[v] = gN[v]; lb[v] = lbN[v]; ub[v] = ubN[v]; x[v] = s == -1 ? lbN[v] : (s == 1 ? ubN[v] : x[v] + tau * dx[v]); }
                else {
                    const double xn = x[v] + tau * dx[v];
                    x[v] = xn; g[v] += tau * (gN[v] - g[v]); gy[v] -= tau * hdv[v];
                    const double l = lb[v] + tau * (lbN[v] - lb[v]), u = ub[v] + tau * (ubN[v] - ub[v]);
                    const bool cap = iter >= maxit;
                    lb[v] = (!cap && b.kind == 4 && b.side == -1 && v == b.idx) ? xn : l;
                    ub[v] = (!cap && b.kind == 4 && b.side == 1 && v == b.idx) ? xn : u;
                }
                yB[v] += tau * dyB[v];
            }
            for (int i = 0; i < nC; i++) {
                yC[i] += tau * dyC[i];
                if (done) { lbA[i] = lbAN[i]; ubA[i] = ubAN[i]; }
                else {
                    const double an = Ax[i] + tau * dAx[i];
                    Ax[i] = an;
                    const double l = lbA[i] + tau * (lbAN[i] - lbA[i]), u = ubA[i] + tau * (ubAN[i] - ubA[i]);
                    const bool cap = iter >= maxit;
                    lbA[i] = (!cap && b.kind == 3 && b.side == -1 && i == b.idx) ? an : l;
                    ubA[i] = (!cap && b.kind == 3 && b.side == 1 && i == b.idx) ? an : u;
                }
            }
            if (done || iter >= maxit) {
                A_times(x.data(), Ax.data());
                if (done) {
                    // ONE step of iterative refinement on the final KKT system, residuals from the DATA: the tableau is kept current by
                    // updates only, no entry is ever re-derived from H and A, so its rounding accumulates over the changes (measured on 2000
                    // random QPs: worst |dy| 1.4e-9 after 63 changes where the explicit-KKT-inverse form, whose borderings read fresh
                    // columns of K, leaves 1e-12). in = (-(gN + H x - A'y_C) on free variables, 0, limit - A x on active constraints, 0):
                    // out = G in corrects x_FR and y_AC; then the multipliers of the fixed variables from stationarity, exactly.
                    std::vector<double> t1(nV), t2(nV);
                    if (!getenv("PROTOG_NO_POLISH")) {
                        AT_times(yC.data(), t1.data()); H_times(x.data(), t2.data());
                        for (int v = 0; v < nV; v++) in[v] = Sb[v] == 0 ? -(gN[v] + t2[v] - t1[v]) : 0.0;
                        for (int i = 0; i < nC; i++) in[nV + i] = Sc[i] == 0 ? 0.0 : ((Sc[i] == -1 ? lbAN[i] : ubAN[i]) - Ax[i]);
                        for (int i = 0; i < N; i++) { double s_ = 0; for (int j = 0; j < N; j++) s_ += g_(i, j) * in[j]; out[i] = s_; }
                        for (int v = 0; v < nV; v++) if (Sb[v] == 0) x[v] += out[v];
                        for (int i = 0; i < nC; i++) if (Sc[i] != 0) yC[i] -= out[nV + i];
                        AT_times(yC.data(), t1.data()); H_times(x.data(), t2.data());
                        for (int v = 0; v < nV; v++) yB[v] = Sb[v] != 0 ? gN[v] + t2[v] - t1[v] : 0.0;
                        A_times(x.data(), Ax.data());
                    }
                    solved = 1;
                } else rc = RET_MAX_NWSR;
                break;
            }
            rc = change_active_set(b);
            if (rc == RET_INFEASIBLE) { infeasible = 1; break; }
            if (rc == RET_UNBOUNDED) { unbounded = 1; break; }
            if (rc != RET_OK) break;
            iter++;
            since_refresh++;
            for (int v = 0; v < nV; v++) if (Sb[v] != 0) x[v] = Sb[v] == -1 ? lb[v] : ub[v];     // x exactly on its active bounds
        }
        nWSR = iter;
        return rc;
    }
};
}  // namespace

extern "C" int protok_solve(int nV, int nC, const double *A, const double *H, const double *g, const double *lb, const double *ub,
                            const double *lbA, const double *ubA, int maxit, double *x, double *y, int *Sb, int *Sc, int *nWSR,
                            int *info) {
    G_ e;
    e.nV = nV; e.nC = nC; e.N = nV + nC;
    e.quick_li = getenv("PROTOG_SLOW_LI") ? 0 : 1;
    e.A.assign(A, A + (size_t)nC * nV); e.H.assign(H, H + (size_t)nV * nV);
    e.G.assign((size_t)e.N * e.N, 0.0);
    e.Sb.assign(nV, 0); e.Sc.assign(nC, 0);
    for (auto *v : {&e.x, &e.g, &e.lb, &e.ub, &e.gN, &e.lbN, &e.ubN, &e.dx, &e.yB, &e.gy, &e.dyB, &e.hdv}) v->assign(nV, 0.0);
    for (auto *v : {&e.lbA, &e.ubA, &e.Ax, &e.lbAN, &e.ubAN, &e.dAx, &e.yC, &e.dyC}) v->assign(nC, 0.0);
    e.in.assign(e.N, 0.0); e.out.assign(e.N, 0.0);
    for (int v = 0; v < nV; v++) { e.gN[v] = g[v]; e.lbN[v] = e.clampinf(lb[v]); e.ubN[v] = e.clampinf(ub[v]); e.hscale = std::fmax(e.hscale, std::fabs(H[v + (size_t)v * nV])); }
    for (int i = 0; i < nC; i++) { e.lbAN[i] = e.clampinf(lbA[i]); e.ubAN[i] = e.clampinf(ubA[i]); }
    info[0] = info[1] = info[2] = info[3] = 0;
    if (!(e.hscale > 0.0)) { info[0] = 10; return RET_BAIL; }
    for (int v = 0; v < nV; v++) for (int u = 0; u < v; u++) if (H[u + (size_t)v * nV] != H[v + (size_t)u * nV]) { info[0] = 10; return RET_BAIL; }
    for (int v = 0; v < nV; v++) if (e.lbN[v] > e.ubN[v] + EPS) { info[1] = 1; *nWSR = 0; return RET_INFEASIBLE; }
    for (int i = 0; i < nC; i++) if (e.lbAN[i] > e.ubAN[i] + EPS) { info[1] = 1; *nWSR = 0; return RET_INFEASIBLE; }
    int rc = e.setup_cold();
    if (rc == RET_OK) rc = e.homotopy(maxit, *nWSR);
    info[0] = e.bail_reason; info[1] = e.infeasible; info[2] = e.solved; info[3] = (int)e.stat_slow_li;
    std::memcpy(x, e.x.data(), 8 * nV);
    std::memcpy(y, e.yB.data(), 8 * nV); if (nC) std::memcpy(y + nV, e.yC.data(), 8 * nC);
    std::memcpy(Sb, e.Sb.data(), 4 * nV); if (nC) std::memcpy(Sc, e.Sc.data(), 4 * nC);
    return rc;
}
