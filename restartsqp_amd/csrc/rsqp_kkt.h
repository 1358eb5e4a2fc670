// rsqp_kkt.h -- the per-entry rules of the reference's working-set mapping and KKT certificate
// (src/qpOASESInterface.cpp:835-895 get_working_set, :498-684 test_optimality), shared by the certificate kernels of
// sparse.hip and the fused certificate at the end of the hs071-scale solve kernel (qp_tiny.hip).
#pragma once
#include <hip/hip_runtime.h>

#define RSQP_K_ABOVE 1
#define RSQP_K_BELOW (-1)
#define RSQP_K_BOTH (-99)
#define RSQP_K_INACTIVE 0
#define RSQP_K_INVALID 12345
#define RSQP_K_INFTY 1.0e20

__device__ inline int map_bound(int ws, double x, double lb, double ub) {
    // src/qpOASESInterface.cpp:846-868
    if (ws == 1) return fabs(x - lb) < 1.0e-8 ? RSQP_K_BOTH : RSQP_K_ABOVE;
    if (ws == -1) return fabs(x - ub) < 1.0e-8 ? RSQP_K_BOTH : RSQP_K_BELOW;
    return ws == 0 ? RSQP_K_INACTIVE : RSQP_K_INVALID;
}
__device__ inline int map_constr(int ws, double Ax, double lbA, double ubA) {
    // src/qpOASESInterface.cpp:871-892 -- fabs() wraps the comparison there, so the
    // test is the SIGNED one
    if (ws == 1) return (Ax - lbA < 1.0e-8) ? RSQP_K_BOTH : RSQP_K_ABOVE;
    if (ws == -1) return (Ax - ubA < 1.0e-8) ? RSQP_K_BOTH : RSQP_K_BELOW;
    return ws == 0 ? RSQP_K_INACTIVE : RSQP_K_INVALID;
}
__device__ inline void kkt_terms(int W, double yv, double val, double lo, double hi, double &dual,
                                 double &compl_, int &bad) {
    // dual feasibility :533-578 and complementarity :611-658
    switch (W) {
    case RSQP_K_INACTIVE: dual += fabs(yv); compl_ += fabs(yv); break;
    case RSQP_K_BELOW: dual += -fmin(0.0, yv); compl_ += fabs(yv * (val - lo)); break;
    case RSQP_K_ABOVE: dual += fmax(0.0, yv); compl_ += fabs(yv * (hi - val)); break;
    case RSQP_K_BOTH: break;
    default: bad = 1;
    }
}
