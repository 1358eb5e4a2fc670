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

#include "HipQPInterface.hpp"

using namespace rsqp;

int main(int argc, char **argv) {
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
