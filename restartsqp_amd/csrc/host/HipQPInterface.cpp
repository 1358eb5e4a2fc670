// HipQPInterface.cpp -- see HipQPInterface.hpp. No arithmetic happens here: every method is
// one call into librsqp_hip.so; error codes become the reference's exceptions.
#include "HipQPInterface.hpp"

namespace rsqp {

void HipQPInterface::check(int rc, const char *what) const {
    if (rc >= 0) return;
    std::string msg = std::string(what) + ": " + rsqp_last_error();
    if (rc == RSQP_ERR_WORKING_SET) throw INVALID_WORKING_SET(msg);
    throw QP_INTERNAL_ERROR(msg);
}

HipQPInterface::HipQPInterface(NLPInfo nlp_info, QPType, std::shared_ptr<const Options> options,
                               Ipopt::SmartPtr<Ipopt::Journalist> jnlst, int device)
    : HipQPInterface(nlp_info.nVar + 2 * nlp_info.nCon, nlp_info.nCon, options, device) {
    jnlst_ = jnlst;
}

// src/qpOASESInterface.cpp:54-94: sizes from A, data handed over once; the reference wraps the CSC arrays of
// H and A in qpOASES matrix objects (:77-90), here they go to the device
HipQPInterface::HipQPInterface(std::shared_ptr<SpHbMat> H, std::shared_ptr<SpHbMat> A, std::shared_ptr<Vector> g,
                               std::shared_ptr<Vector> lb, std::shared_ptr<Vector> ub, std::shared_ptr<Vector> lbA,
                               std::shared_ptr<Vector> ubA, std::shared_ptr<Options> options, int device)
    : HipQPInterface(A->ColNum(), A->RowNum(), options, device) {
    if (H->isCompressedRow() || A->isCompressedRow())
        throw QP_INTERNAL_ERROR("HipQPInterface: H and A must be compressed-column (as for qpOASES)");
    if (H->RowNum() != nVar_QP_ || H->ColNum() != nVar_QP_ || g->Dim() != nVar_QP_ || lb->Dim() != nVar_QP_ ||
        ub->Dim() != nVar_QP_ || lbA->Dim() != nConstr_QP_ || ubA->Dim() != nConstr_QP_)
        throw QP_INTERNAL_ERROR("HipQPInterface: inconsistent dimensions");
    set_A_csc(A->ColIndex(), A->RowIndex(), A->MatVal());
    set_H_csc(H->ColIndex(), H->RowIndex(), H->MatVal());
    set_g(g); set_lb(lb); set_ub(ub); set_lbA(lbA); set_ubA(ubA);
}

HipQPInterface::HipQPInterface(int nVar_QP, int nConstr_QP, std::shared_ptr<const Options> options, int device)
    : nVar_QP_(nVar_QP), nConstr_QP_(nConstr_QP), options_(options), x_qp_(nVar_QP), y_qp_(nVar_QP + nConstr_QP) {
    check(rsqp_create(nVar_QP_, nConstr_QP_, device, &solver_), "rsqp_create");
    if (options_) check(rsqp_set_options(solver_, options_->qp_maxiter, options_->lp_maxiter), "rsqp_set_options");
}

HipQPInterface::~HipQPInterface() { rsqp_destroy(solver_); }

void HipQPInterface::set_A_csc(const int *jc, const int *ir, const double *val) {
    check(rsqp_set_A_csc(solver_, jc, ir, val), "rsqp_set_A_csc");
}
void HipQPInterface::set_H_csc(const int *jc, const int *ir, const double *val) {
    check(rsqp_set_H_csc(solver_, jc, ir, val), "rsqp_set_H_csc");
}

const std::shared_ptr<Vector> &HipQPInterface::refresh(int which, std::shared_ptr<Vector> &v) const {
    if (!v) v = std::make_shared<Vector>(which <= RSQP_VEC_UB ? nVar_QP_ : nConstr_QP_);
    check(rsqp_get_vector(solver_, which, v->values()), "rsqp_get_vector");
    return v;
}

// host mirror of the device CSC (values read back) with the products bound to the device copy
std::shared_ptr<const SpHbMat> HipQPInterface::mirror(bool isA) const {
    const int nnz = isA ? rsqp_get_A_nnz(solver_) : rsqp_get_H_nnz(solver_);
    auto M = std::make_shared<SpHbMat>(isA ? nConstr_QP_ : nVar_QP_, nVar_QP_, false);
    M->ColIndex_.assign(nVar_QP_ + 1, 0);
    if (nnz >= 0) {
        M->RowIndex_.resize(nnz); M->MatVal_.resize(nnz); M->order_.resize(nnz);
        check(isA ? rsqp_get_A_csc(solver_, M->ColIndex_.data(), M->RowIndex_.data(), M->MatVal_.data(), M->order_.data())
                  : rsqp_get_H_csc(solver_, M->ColIndex_.data(), M->RowIndex_.data(), M->MatVal_.data(), M->order_.data()),
              "rsqp_get_*_csc");
    }
    rsqp_solver *h = solver_;
    if (isA) {
        M->times_ = [h](const double *p, double *r) { if (rsqp_A_times(h, p, r) < 0) throw QP_INTERNAL_ERROR(rsqp_last_error()); };
        M->ttimes_ = [h](const double *p, double *r) { if (rsqp_A_transposed_times(h, p, r) < 0) throw QP_INTERNAL_ERROR(rsqp_last_error()); };
    } else {   // symmetric: both products are the same
        M->times_ = M->ttimes_ = [h](const double *p, double *r) { if (rsqp_H_times(h, p, r) < 0) throw QP_INTERNAL_ERROR(rsqp_last_error()); };
    }
    return M;
}
std::shared_ptr<const SpHbMat> HipQPInterface::getA() const { return mirror(true); }
std::shared_ptr<const SpHbMat> HipQPInterface::getH() const { return mirror(false); }

void HipQPInterface::WriteQPDataToFile(Ipopt::EJournalLevel, Ipopt::EJournalCategory, const std::string filename) {
    check(rsqp_write_qp_data(solver_, ("qpOASES" + filename).c_str(), RSQP_DUMP_QPOASES), "rsqp_write_qp_data");
}
void HipQPInterface::WriteQPDataToFileQORE(const std::string filename) {
    check(rsqp_write_qp_data(solver_, ("QORE_" + filename).c_str(), RSQP_DUMP_QORE), "rsqp_write_qp_data");
}

void HipQPInterface::fetch_solution() {
    check(rsqp_get_primal(solver_, x_qp_.values()), "rsqp_get_primal");
    check(rsqp_get_dual(solver_, y_qp_.values()), "rsqp_get_dual");
}

// src/qpOASESInterface.cpp:137-224 -- dispatch, handle_error and stats accounting are inside
// rsqp_optimize_qp; the throw of handle_error (:754-756) happens here
void HipQPInterface::optimizeQP(std::shared_ptr<Stats> stats) {
    int nWSR = 0;
    check(rsqp_optimize_qp(solver_, &nWSR), "rsqp_optimize_qp");
    if (stats != nullptr) stats->qp_iter_addValue(nWSR);
    fetch_solution();
    if (!rsqp_is_solved(solver_)) throw QP_NOT_OPTIMAL("QP solver did not reach optimality");
}

// src/qpOASESInterface.cpp:227-284
void HipQPInterface::optimizeLP(std::shared_ptr<Stats> stats) {
    int nWSR = 0;
    check(rsqp_optimize_lp(solver_, &nWSR), "rsqp_optimize_lp");
    if (stats != nullptr) stats->qp_iter_addValue(nWSR);
    fetch_solution();
    if (!rsqp_is_solved(solver_)) throw LP_NOT_OPTIMAL("LP solver did not reach optimality");
}

double HipQPInterface::get_obj_value() { return rsqp_get_objective(solver_); }
Exitflag HipQPInterface::get_status() { return rsqp_get_status(solver_); }

void HipQPInterface::get_working_set(ActiveType *W_constr, ActiveType *W_bounds) {
    static_assert(sizeof(ActiveType) == sizeof(int), "ActiveType must be int sized");
    check(rsqp_get_working_set(solver_, reinterpret_cast<int *>(W_constr), reinterpret_cast<int *>(W_bounds)),
          "rsqp_get_working_set");
}

bool HipQPInterface::test_optimality(ActiveType *W_c, ActiveType *W_b) {
    rsqp_optimality_status st;
    int rc = rsqp_test_optimality(solver_, reinterpret_cast<int *>(W_c), reinterpret_cast<int *>(W_b), &st);
    check(rc, "rsqp_test_optimality");
    qpOptimalStatus_.primal_violation = st.primal_violation;
    qpOptimalStatus_.dual_violation = st.dual_violation;
    qpOptimalStatus_.compl_violation = st.compl_violation;
    qpOptimalStatus_.stationarity_violation = st.stationarity_violation;
    qpOptimalStatus_.KKT_error = st.KKT_error;
    return rc == 1;
}

void HipQPInterface::set_lb(int l, double v) { check(rsqp_set_entry(solver_, RSQP_VEC_LB, l, v), "set_lb"); }
void HipQPInterface::set_ub(int l, double v) { check(rsqp_set_entry(solver_, RSQP_VEC_UB, l, v), "set_ub"); }
void HipQPInterface::set_lbA(int l, double v) { check(rsqp_set_entry(solver_, RSQP_VEC_LBA, l, v), "set_lbA"); }
void HipQPInterface::set_ubA(int l, double v) { check(rsqp_set_entry(solver_, RSQP_VEC_UBA, l, v), "set_ubA"); }
void HipQPInterface::set_g(int l, double v) { check(rsqp_set_entry(solver_, RSQP_VEC_G, l, v), "set_g"); }
void HipQPInterface::set_lb(std::shared_ptr<const Vector> r) { check(rsqp_set_vector(solver_, RSQP_VEC_LB, r->values()), "set_lb"); }
void HipQPInterface::set_ub(std::shared_ptr<const Vector> r) { check(rsqp_set_vector(solver_, RSQP_VEC_UB, r->values()), "set_ub"); }
void HipQPInterface::set_lbA(std::shared_ptr<const Vector> r) { check(rsqp_set_vector(solver_, RSQP_VEC_LBA, r->values()), "set_lbA"); }
void HipQPInterface::set_ubA(std::shared_ptr<const Vector> r) { check(rsqp_set_vector(solver_, RSQP_VEC_UBA, r->values()), "set_ubA"); }
void HipQPInterface::set_g(std::shared_ptr<const Vector> r) { check(rsqp_set_vector(solver_, RSQP_VEC_G, r->values()), "set_g"); }

// src/qpOASESInterface.cpp:400-442: structure on the first call, values afterwards
void HipQPInterface::set_H(std::shared_ptr<const SpTripletMat> rhs) {
    check(rsqp_set_H_triplet(solver_, rhs->EntryNum(), rhs->RowIndex.data(), rhs->ColIndex.data(), rhs->MatVal.data(),
                             rhs->isSymmetric ? 1 : 0), "rsqp_set_H_triplet");
}
void HipQPInterface::set_A(std::shared_ptr<const SpTripletMat> rhs, IdentityInfo I) {
    check(rsqp_set_A_triplet(solver_, rhs->EntryNum(), rhs->RowIndex.data(), rhs->ColIndex.data(), rhs->MatVal.data(),
                             I.length, I.irow, I.jcol, I.size, I.value), "rsqp_set_A_triplet");
}
void HipQPInterface::reset_constraints() { check(rsqp_reset_constraints(solver_), "rsqp_reset_constraints"); }

}  // namespace rsqp
