// host_replay.cpp -- drives HipQPInterface with the call sequence QPhandler produces for the
// first two SQP iterations of hs071 (reference src/Algorithm.cpp:645-697 -> src/QPhandler.cpp):
// set_A, set_H, per-element bounds, per-element g, optimizeQP, test_optimality; then a
// trust-region update (update_delta) and a hot start. Prints one line per solve; the GPU test
// tests/test_gpu_host_adapter.py compares the lines with the CPU reference restatement kept under tests.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

#include "HipQPInterface.hpp"

using namespace rsqp;

namespace {

const double INF_REF = 1.0e18;  // include/sqphot/Utils.hpp:35

// hs071 (test/CUTE_examples/hs071.nl) in closed form: what SQPTNLP hands to Algorithm
struct Hs071 {
    double x[4], c[2], grad[4];
    std::shared_ptr<SpTripletMat> J, H;
    static constexpr double x_l[4] = {1, 1, 1, 1}, x_u[4] = {5, 5, 5, 5};
    static constexpr double c_l[2] = {25, 40};
    double c_u[2] = {std::numeric_limits<double>::infinity(), 40};
    explicit Hs071(const double *xk, const double *lam = nullptr) {
        for (int i = 0; i < 4; i++) x[i] = xk[i];
        const double x1 = x[0], x2 = x[1], x3 = x[2], x4 = x[3];
        c[0] = x1 * x2 * x3 * x4; c[1] = x1 * x1 + x2 * x2 + x3 * x3 + x4 * x4;
        grad[0] = x4 * (2 * x1 + x2 + x3); grad[1] = x1 * x4; grad[2] = x1 * x4 + 1.0; grad[3] = x1 * (x1 + x2 + x3);
        J = std::make_shared<SpTripletMat>();
        J->RowNum = 2; J->ColNum = 4;
        J->RowIndex = {1, 1, 1, 1, 2, 2, 2, 2}; J->ColIndex = {1, 2, 3, 4, 1, 2, 3, 4};
        J->MatVal = {x2 * x3 * x4, x1 * x3 * x4, x1 * x2 * x4, x1 * x2 * x3, 2 * x1, 2 * x2, 2 * x3, 2 * x4};
        H = std::make_shared<SpTripletMat>();   // Hessian of f (zero multipliers), lower triangle row by row
        H->RowNum = H->ColNum = 4; H->isSymmetric = true;
        H->RowIndex = {1, 2, 2, 3, 3, 3, 4, 4, 4, 4}; H->ColIndex = {1, 1, 2, 1, 2, 3, 1, 2, 3, 4};
        H->MatVal = {2 * x4, x4, 0, x4, 0, 0, 2 * x1 + x2 + x3, x1, x1, 0};
        if (lam) {   // Hessian of the Lagrangian f - lam'c (SQPTNLP::Eval_Hessian negates lambda, src/SQPTNLP.cpp:124-126)
            const double l1 = lam[0], l2 = lam[1];
            const double hc1[10] = {0, x3 * x4, 0, x2 * x4, x1 * x4, 0, x2 * x3, x1 * x3, x1 * x2, 0};
            const bool diag[10] = {true, false, true, false, false, true, false, false, false, true};
            for (int e = 0; e < 10; e++) H->MatVal[e] -= l1 * hc1[e] + (diag[e] ? 2.0 * l2 : 0.0);
        }
    }
};
constexpr double Hs071::x_l[4], Hs071::x_u[4], Hs071::c_l[2];

// the caller side of the boundary: QPhandler's formulas (reference src/QPhandler.cpp), restated so that
// the adapter sees the call storm Algorithm produces (per-element virtual setters)
struct Handler {
    int n, m;
    int irow[2], jcol[2], size[2];
    double value[2];
    IdentityInfo I;
    std::shared_ptr<HipQPInterface> solver;
    Handler(NLPInfo info, QPType t, std::shared_ptr<const Options> opt, Ipopt::SmartPtr<Ipopt::Journalist> jnlst)
        : n(info.nVar), m(info.nCon) {
        irow[0] = irow[1] = 1; jcol[0] = n + 1; jcol[1] = n + m + 1; size[0] = size[1] = m; value[0] = 1.0; value[1] = -1.0;
        I = IdentityInfo{2, irow, jcol, size, value};                                  // QPhandler.cpp:41-51
        solver = std::make_shared<HipQPInterface>(info, t, opt, jnlst);                // :58-76
    }
    void set_bounds(double delta, const double *x_l, const double *x_u, const double *x_k, const double *c_l,
                    const double *c_u, const double *c_k) {                             // :167-201
        for (int i = 0; i < m; i++) { solver->set_lbA(i, c_l[i] - c_k[i]); solver->set_ubA(i, c_u[i] - c_k[i]); }
        for (int i = 0; i < n; i++) {
            solver->set_lb(i, std::fmax(x_l[i] - x_k[i], -delta));
            solver->set_ub(i, std::fmin(x_u[i] - x_k[i], delta));
        }
        for (int i = 0; i < 2 * m; i++) solver->set_ub(n + i, INF_REF);
    }
    // refresh_ubA: NOT the reference (its qpOASES branch leaves ubA stale, :358-360, which turns its own run infeasible after the
    // first accepted step when a constraint is an equality); the whole-trajectory replay needs the value and passes c_u
    void update_bounds(double delta, const double *x_l, const double *x_u, const double *x_k, const double *c_l,
                       const double *c_k, const double *refresh_ubA_c_u = nullptr) {    // :342-368 (ubA is not refreshed)
        for (int i = 0; i < m; i++) {
            solver->set_lbA(i, c_l[i] - c_k[i]);
            if (refresh_ubA_c_u) solver->set_ubA(i, refresh_ubA_c_u[i] - c_k[i]);
        }
        for (int i = 0; i < n; i++) {
            solver->set_lb(i, std::fmax(x_l[i] - x_k[i], -delta));
            solver->set_ub(i, std::fmin(x_u[i] - x_k[i], delta));
        }
    }
    void set_g(const double *grad, double rho) { for (int i = 0; i < n + 2 * m; i++) solver->set_g(i, i < n ? grad[i] : rho); }  // :272-297
    void set_g(double rho) { for (int i = n; i < n + 2 * m; i++) solver->set_g(i, rho); }                                       // :657-660 (LP)
    void update_penalty(double rho) { for (int i = n; i < n + 2 * m; i++) solver->set_g(i, rho); }                               // :430-441
    void update_grad(const double *grad) { for (int i = 0; i < n; i++) solver->set_g(i, grad[i]); }                              // :450-463
    void set_A(std::shared_ptr<const SpTripletMat> J) { solver->set_A(J, I); }
    void set_H(std::shared_ptr<const SpTripletMat> H) { solver->set_H(H); }
    void solveQP(std::shared_ptr<Stats> stats) {                                        // :470-499
        solver->optimizeQP(stats);
        std::vector<ActiveType> Wc(m), Wb(n + 2 * m);
        if (!solver->test_optimality(Wc.data(), Wb.data())) throw QP_NOT_OPTIMAL("KKT certificate failed");
    }
    void solveLP(std::shared_ptr<Stats> stats) { solver->optimizeLP(stats); }           // QPhandler.hpp:66-68
    double get_infea_measure_model() {                                                  // :592-594
        const double *x = solver->get_optimal_solution();
        double s = 0.0;
        for (int i = n; i < n + 2 * m; i++) s += std::fabs(x[i]);
        return s;
    }
    void report(const char *tag, std::shared_ptr<Stats> stats) {
        const double *x = solver->get_optimal_solution();
        std::printf("%s status %d qp_iter %d obj %.15g infea_model %.15g x", tag, solver->get_status(), stats->qp_iter,
                    solver->get_obj_value(), get_infea_measure_model());
        for (int i = 0; i < n + 2 * m; i++) std::printf(" %.15g", x[i]);
        std::printf("\n");
    }
};

// Algorithm::update_penalty_parameter (reference src/Algorithm.cpp:886-1028: setupLP + solveLP on the LP
// handler, then update_penalty + solveQP on the QP handler, repeatedly) followed by
// Algorithm::second_order_correction (:1144-1211: update_grad(H p + g), update_bounds at x_trial, solveQP,
// and the restore calls). Two handler objects -- LP and QP (Algorithm.cpp:561-562) -- live side by side.
int penalty_soc_trace() {
    NLPInfo info{2, 4, 8, 10};
    auto options = std::make_shared<Options>();
    auto stats = std::make_shared<Stats>();
    // hs071 with c2 relaxed to c2 >= 40 (the reference's update_bounds never refreshes ubA, QPhandler.cpp:358-360,
    // so an equality cannot be replayed through an SOC step), at a point that violates both constraints;
    // delta is small enough that the linearised constraints cannot be met: the slacks stay positive
    const double x0[4] = {1, 2, 2, 1};
    Hs071 nlp(x0);
    nlp.c_u[1] = std::numeric_limits<double>::infinity();
    const double delta = 0.25;
    double rho = 1.0;
    Handler myQP(info, QP, options, nullptr), myLP(info, LP, options, nullptr);
    myQP.set_A(nlp.J); myQP.set_H(nlp.H);
    myQP.set_bounds(delta, nlp.x_l, nlp.x_u, nlp.x, nlp.c_l, nlp.c_u, nlp.c);
    myQP.set_g(nlp.grad, rho);
    myQP.solveQP(stats);
    myQP.report("qp rho=1", stats);
    // setupLP (:700-704) + solveLP
    myLP.set_bounds(delta, nlp.x_l, nlp.x_u, nlp.x, nlp.c_l, nlp.c_u, nlp.c);
    myLP.set_g(rho);
    myLP.set_A(nlp.J);
    myLP.solveLP(stats);
    myLP.report("lp rho=1", stats);
    // the loop of :933-960 for two trial values
    for (int k = 0; k < 2; k++) {
        rho *= 10.0;
        myQP.update_penalty(rho);
        myQP.solveQP(stats);
        myQP.report(k == 0 ? "qp rho=10" : "qp rho=100", stats);
    }
    // second-order correction: p = x_qp[0..4), g := H p + grad, bounds at x_trial = x + p
    double p[4], Hp[4] = {0, 0, 0, 0}, xt[4];
    for (int i = 0; i < 4; i++) p[i] = myQP.solver->get_optimal_solution()[i];
    for (size_t e = 0; e < nlp.H->MatVal.size(); e++) {
        const int r = nlp.H->RowIndex[e] - 1, c = nlp.H->ColIndex[e] - 1;
        Hp[r] += nlp.H->MatVal[e] * p[c];
        if (r != c) Hp[c] += nlp.H->MatVal[e] * p[r];
    }
    for (int i = 0; i < 4; i++) { Hp[i] += nlp.grad[i]; xt[i] = nlp.x[i] + p[i]; }
    Hs071 trial(xt);
    myQP.update_grad(Hp);
    myQP.update_bounds(delta, nlp.x_l, nlp.x_u, trial.x, nlp.c_l, trial.c);
    myQP.solveQP(stats);
    myQP.report("qp soc", stats);
    myQP.update_grad(nlp.grad);                                                // step rejected: restore (:1204-1208)
    myQP.update_bounds(delta, nlp.x_l, nlp.x_u, nlp.x, nlp.c_l, nlp.c);
    myQP.solveQP(stats);
    myQP.report("qp restored", stats);
    // the next penalty update reuses the LP object: hot start with a new gradient (rho) and a re-set Jacobian
    myLP.set_bounds(delta, nlp.x_l, nlp.x_u, nlp.x, nlp.c_l, nlp.c_u, nlp.c);
    myLP.set_g(rho);
    myLP.set_A(nlp.J);
    myLP.solveLP(stats);
    myLP.report("lp rho=100", stats);
    return 0;
}

// "wall-clock per SQP iteration (hs071)" as the reference clocks it (src/Algorithm.cpp:57,138-139: one clock() pair per pass of
// the while loop): the QP side of EVERY iteration of a whole hs071 run -- Algorithm::setupQP (:645-697: iteration 0 set_A,
// set_H, set_bounds, set_g; later update_A / update_H / update_bounds / update_penalty / update_grad through the per-element
// setters), QPhandler::solveQP (optimizeQP + mandatory certificate, src/QPhandler.cpp:470-499) and the getters the loop
// reads (:609-622). The iterates (delta, rho, x_k, lambda_k per iteration) come from the committed trajectory
// (tests/golden/sqp_traces.json, written to a text file by the caller); NLP evaluation is closed form and not timed apart.
int trajectory_bench(const char *path, int reps) {
    std::vector<double> tr;
    {
        FILE *f = std::fopen(path, "r");
        if (!f) { std::printf("cannot open %s\n", path); return 2; }
        double v;
        while (std::fscanf(f, "%lf", &v) == 1) tr.push_back(v);
        std::fclose(f);
    }
    const int nit = (int)tr.size() / 8;
    if (nit < 1 || reps < 1) { std::printf("empty trajectory\n"); return 2; }
    NLPInfo info{2, 4, 8, 10};
    auto options = std::make_shared<Options>();
    double us_all = 0.0, us_first = 0.0, xlast[8] = {0}, ylast[10] = {0};
    int qp_iter_last = 0;
    for (int r = 0; r < reps; r++) {
        auto stats = std::make_shared<Stats>();
        Handler myQP(info, QP, options, nullptr);          // (Algorithm::allocate_memory: outside the reference's clock as well)
        double rho_prev = 0.0;
        for (int k = 0; k < nit; k++) {
            const double *t = &tr[8 * k];
            const double delta = t[0], rho = t[1];
            const auto t0 = std::chrono::steady_clock::now();
            Hs071 nlp(t + 2, t + 6);
            if (k == 0) {
                // (RSQP_TRAJ_PHASES=1: where the first iteration's time goes -- structure analysis + upload of A, of H, vectors, solve)
                static const bool phases = std::getenv("RSQP_TRAJ_PHASES") != nullptr;
                auto now = [] { return std::chrono::steady_clock::now(); };
                auto us_since = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::micro>(now() - a).count(); };
                auto p0 = now();
                myQP.set_A(nlp.J);
                const double tA = us_since(p0); p0 = now();
                myQP.set_H(nlp.H);
                const double tH = us_since(p0); p0 = now();
                myQP.set_bounds(delta, nlp.x_l, nlp.x_u, nlp.x, nlp.c_l, nlp.c_u, nlp.c);
                myQP.set_g(nlp.grad, rho);
                const double tV = us_since(p0);
                if (phases && r == reps - 1) std::printf("first_iteration_phases set_A %.2f set_H %.2f vectors %.2f us\n", tA, tH, tV);
            } else {
                myQP.set_A(nlp.J); myQP.set_H(nlp.H);      // update_A / update_H (QPhandler.cpp:508-531): value refresh
                myQP.update_bounds(delta, nlp.x_l, nlp.x_u, nlp.x, nlp.c_l, nlp.c, nlp.c_u);
                if (rho != rho_prev) myQP.update_penalty(rho);
                myQP.update_grad(nlp.grad);
            }
            rho_prev = rho;
            myQP.solveQP(stats);
            const double *x = myQP.solver->get_optimal_solution();
            const double *yb = myQP.solver->get_multipliers_bounds(), *yc = myQP.solver->get_multipliers_constr();
            for (int i = 0; i < 8; i++) { xlast[i] = x[i]; ylast[i] = yb[i]; }
            ylast[8] = yc[0]; ylast[9] = yc[1];
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            us_all += us;
            if (k == 0) us_first += us;
        }
        qp_iter_last = stats->qp_iter;
    }
    std::printf("trajectory sqp_iterations %d reps %d us_per_sqp_iteration %.3f us_first_iteration %.3f us_later_iterations %.3f qp_iter %d\n",
                nit, reps, us_all / (reps * (double)nit), us_first / reps, nit > 1 ? (us_all - us_first) / (reps * (double)(nit - 1)) : 0.0,
                qp_iter_last);
    std::printf("trajectory_last_x");
    for (int i = 0; i < 8; i++) std::printf(" %.15g", xlast[i]);
    std::printf("\ntrajectory_last_y");
    for (int i = 0; i < 10; i++) std::printf(" %.15g", ylast[i]);
    std::printf("\n");
    return 0;
}

// plain-QP ctor with data (reference src/qpOASESInterface.cpp:54-94, as test/QPsolvers_testers.cpp:220 uses
// it on the dumps of test/unsolved_QP_data), the data getters QPhandler::get_active_set reads
// (src/QPhandler.cpp:596-650) and WriteQPDataToFile in both layouts
int dump_replay(const char *path, const char *outname) {
    int nV, nC, annz, hnnz;
    if (rsqp_read_qore_dump_sizes(path, &nV, &nC, &annz, &hnnz) != RSQP_OK) { std::printf("cannot read %s\n", path); return 4; }
    auto lb = std::make_shared<Vector>(nV), ub = std::make_shared<Vector>(nV), g = std::make_shared<Vector>(nV);
    auto lbA = std::make_shared<Vector>(nC), ubA = std::make_shared<Vector>(nC);
    std::vector<int> Ajc(nV + 1), Air(annz), Hjc(nV + 1), Hir(hnnz);
    std::vector<double> Aval(annz), Hval(hnnz);
    if (rsqp_read_qore_dump(path, lb->values(), ub->values(), lbA->values(), ubA->values(), g->values(), Ajc.data(), Air.data(),
                            Aval.data(), Hjc.data(), Hir.data(), Hval.data()) != RSQP_OK) { std::printf("bad dump %s\n", path); return 4; }
    auto A = std::make_shared<SpHbMat>(nC, nV, Ajc.data(), Air.data(), Aval.data());
    auto H = std::make_shared<SpHbMat>(nV, nV, Hjc.data(), Hir.data(), Hval.data());
    HipQPInterface qp(H, A, g, lb, ub, lbA, ubA, std::make_shared<Options>());
    auto stats = std::make_shared<Stats>();
    bool solved = true;
    try { qp.optimizeQP(stats); } catch (const QP_NOT_OPTIMAL &) { solved = false; }
    std::printf("dump status %d qp_iter %d solved %d obj %.15g\n", qp.get_status(), stats->qp_iter, solved ? 1 : 0, qp.get_obj_value());
    // QPhandler::get_active_set, qpOASES branch (incl. its getUbA-twice quirk, QPhandler.cpp:641-642)
    const double sqrt_m_eps = 1.4901161193847656e-08;
    auto x = std::make_shared<Vector>(nV), Ax = std::make_shared<Vector>(nC);
    x->copy_vector(qp.get_optimal_solution());
    qp.getA()->times(x, Ax);
    auto l = qp.getLb(), u = qp.getUb();
    std::printf("A_b");
    for (int i = 0; i < nV; i++) {
        const bool lo = std::fabs(x->values(i) - l->values(i)) < sqrt_m_eps, hi = std::fabs(u->values(i) - x->values(i)) < sqrt_m_eps;
        std::printf(" %d", lo ? (hi ? ACTIVE_BOTH_SIDE : ACTIVE_BELOW) : (hi ? ACTIVE_ABOVE : INACTIVE));
    }
    auto la = qp.getUbA(), ua = qp.getUbA();
    std::printf("\nA_c");
    for (int i = 0; i < nC; i++) {
        const bool lo = std::fabs(Ax->values(i) - la->values(i)) < sqrt_m_eps, hi = std::fabs(ua->values(i) - Ax->values(i)) < sqrt_m_eps;
        std::printf(" %d", lo ? (hi ? ACTIVE_BOTH_SIDE : ACTIVE_BELOW) : (hi ? ACTIVE_ABOVE : INACTIVE));
    }
    std::printf("\nAx");
    for (int i = 0; i < nC; i++) std::printf(" %.15g", Ax->values(i));
    std::printf("\ngetG %d getH_nnz %d getA_nnz %d\n", qp.getG()->Dim(), qp.getH()->EntryNum(), qp.getA()->EntryNum());
    qp.WriteQPDataToFile(Ipopt::J_LAST_LEVEL, Ipopt::J_USER1, outname);   // QPhandler::WriteQPData (:569-571)
    qp.WriteQPDataToFileQORE(outname);
    return 0;
}

}  // namespace

int main(int argc, char **argv) {
    try {
        if (argc > 1 && std::strcmp(argv[1], "--penalty") == 0) return penalty_soc_trace();
        if (argc > 3 && std::strcmp(argv[1], "--dump") == 0) return dump_replay(argv[2], argv[3]);
        if (argc > 3 && std::strcmp(argv[1], "--trajectory") == 0) return trajectory_bench(argv[2], std::atoi(argv[3]));
    } catch (const std::exception &e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 2;
    }
    const int bench_iters = (argc > 2 && std::strcmp(argv[1], "--bench") == 0) ? std::atoi(argv[2]) : 0;
    const double INF = 1.0e18;  // include/sqphot/Utils.hpp:35
    NLPInfo info{2, 4, 8, 10};
    auto options = std::make_shared<Options>();
    auto stats = std::make_shared<Stats>();
    try {
        HipQPInterface qp(info, QP, options);
        // hs071 at x0 = (1,5,5,1): J (1-based COO), Hessian of f (lower triangle)
        auto J = std::make_shared<SpTripletMat>();
        J->RowNum = 2; J->ColNum = 4;
        J->RowIndex = {1, 1, 1, 1, 2, 2, 2, 2}; J->ColIndex = {1, 2, 3, 4, 1, 2, 3, 4};
        J->MatVal = {25, 5, 5, 25, 2, 10, 10, 2};
        auto H = std::make_shared<SpTripletMat>();
        H->RowNum = H->ColNum = 4; H->isSymmetric = true;
        H->RowIndex = {1, 2, 2, 3, 3, 3, 4, 4, 4, 4}; H->ColIndex = {1, 1, 2, 1, 2, 3, 1, 2, 3, 4};
        H->MatVal = {2, 1, 0, 1, 0, 0, 12, 1, 1, 0};
        int irow[2] = {1, 1}, jcol[2] = {5, 7}, size[2] = {2, 2};
        double value[2] = {1.0, -1.0};
        IdentityInfo I{2, irow, jcol, size, value};  // QPhandler.cpp:41-51
        qp.set_A(J, I);
        qp.set_H(H);
        const double x_l[4] = {1, 1, 1, 1}, x_u[4] = {5, 5, 5, 5}, x_k[4] = {1, 5, 5, 1};
        const double c_l[2] = {25, 40}, c_u[2] = {std::numeric_limits<double>::infinity(), 40}, c_k[2] = {25, 52};
        const double grad[4] = {12, 1, 2, 11};
        double delta = 1.0, rho = 1.0;
        for (int i = 0; i < 2; i++) { qp.set_lbA(i, c_l[i] - c_k[i]); qp.set_ubA(i, c_u[i] - c_k[i]); }
        for (int i = 0; i < 4; i++) {
            qp.set_lb(i, std::fmax(x_l[i] - x_k[i], -delta));
            qp.set_ub(i, std::fmin(x_u[i] - x_k[i], delta));
        }
        for (int i = 0; i < 4; i++) qp.set_ub(4 + i, INF);
        for (int i = 0; i < 8; i++) qp.set_g(i, i < 4 ? grad[i] : rho);
        for (int solve = 0; solve < 2; solve++) {
            if (solve == 1) {  // QPhandler::update_delta (QPhandler.cpp:533-567)
                delta = 0.5;
                for (int i = 0; i < 4; i++) {
                    qp.set_lb(i, std::fmax(x_l[i] - x_k[i], -delta));
                    qp.set_ub(i, std::fmin(x_u[i] - x_k[i], delta));
                }
            }
            qp.optimizeQP(stats);
            ActiveType Wc[2], Wb[8];
            bool ok = qp.test_optimality(Wc, Wb);
            const double *x = qp.get_optimal_solution();
            std::printf("solve %d status %d qp_iter %d kkt_ok %d kkt %.3e obj %.15g x", solve, qp.get_status(),
                        stats->qp_iter, ok ? 1 : 0, qp.get_optimality_status().KKT_error, qp.get_obj_value());
            for (int i = 0; i < 8; i++) std::printf(" %.15g", x[i]);
            std::printf(" Wc %d %d\n", (int)Wc[0], (int)Wc[1]);
        }
        if (bench_iters > 0) {
            // "wall-clock per SQP iteration" at the boundary: QPhandler::update_delta (8 scalar
            // setters) + solveQP (optimizeQP + mandatory KKT certificate), alternating radii
            for (int pass = 0; pass < 2; pass++) {
                const bool with_cert = pass == 0;
                auto t0 = std::chrono::steady_clock::now();
                for (int it = 0; it < bench_iters; it++) {
                    delta = (it & 1) ? 0.5 : 1.0;
                    for (int i = 0; i < 4; i++) {
                        qp.set_lb(i, std::fmax(x_l[i] - x_k[i], -delta));
                        qp.set_ub(i, std::fmin(x_u[i] - x_k[i], delta));
                    }
                    qp.optimizeQP(stats);
                    if (with_cert) {
                        ActiveType Wc[2], Wb[8];
                        if (!qp.test_optimality(Wc, Wb)) { std::printf("bench: certificate failed\n"); return 3; }
                    }
                }
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                std::printf("bench %s us_per_sqp_iteration %.2f iters %d\n", with_cert ? "solveQP(optimizeQP+certificate)" : "optimizeQP_only",
                            us / bench_iters, bench_iters);
            }
        }
    } catch (const std::exception &e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 2;
    }
    return 0;
}
