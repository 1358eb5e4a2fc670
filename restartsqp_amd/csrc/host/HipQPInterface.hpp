// HipQPInterface.hpp -- C++ host adapter: the QPSolverInterface subclass a RestartSQP
// maintainer compiles into the reference. Mirrors qpOASESInterface
// (reference include/sqphot/qpOASESInterface.hpp:37-262) method for method and forwards to the
// C ABI of include/rsqp_hip.h. Inside RestartSQP the types below are the reference's own
// (sqphot/Vector.hpp, SpTripletMat.hpp, Types.hpp, Ipopt's DECLARE_STD_EXCEPTION); this header
// carries minimal equivalents so that the adapter builds and is tested stand-alone.
#pragma once
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/rsqp_hip.h"

namespace rsqp {

// ---- the reference's plain types that cross the boundary (include/sqphot/Types.hpp) ----
enum ActiveType { ACTIVE_ABOVE = 1, ACTIVE_BELOW = -1, ACTIVE_BOTH_SIDE = -99, INACTIVE = 0 };
enum QPType { LP = 1, QP = 2 };
typedef int Exitflag;  // numeric values of Types.hpp:51-73 (QP_OPTIMAL = 20, ...)
struct IdentityInfo { int length; int *irow; int *jcol; int *size; double *value; };
struct NLPInfo { int nCon, nVar, nnz_jac_g, nnz_h_lag; };
struct OptimalityStatus {
    double primal_violation = 0, dual_violation = 0, compl_violation = 0, stationarity_violation = 0, KKT_error = 0;
};
struct Options { int qp_maxiter = 1000, lp_maxiter = 100, qpPrintLevel = 0; };
struct Stats { int qp_iter = 0; void qp_iter_addValue(int n) { qp_iter += n; } };

// exceptions of include/sqphot/QPsolverInterface.hpp:26-32
struct QP_NOT_OPTIMAL : std::runtime_error { using std::runtime_error::runtime_error; };
struct LP_NOT_OPTIMAL : std::runtime_error { using std::runtime_error::runtime_error; };
struct QP_INTERNAL_ERROR : std::runtime_error { using std::runtime_error::runtime_error; };
struct INVALID_WORKING_SET : std::runtime_error { using std::runtime_error::runtime_error; };

// dense vector (include/sqphot/Vector.hpp): contiguous double[]
class Vector {
public:
    explicit Vector(int n) : v_(n, 0.0) {}
    int Dim() const { return (int)v_.size(); }
    double *values() { return v_.data(); }
    const double *values() const { return v_.data(); }
    double values(int i) const { return v_[i]; }
    void setValueAt(int i, double x) { v_[i] = x; }
    void copy_vector(const double *p) { v_.assign(p, p + v_.size()); }
private:
    std::vector<double> v_;
};

// 1-based COO (include/sqphot/SpTripletMat.hpp); symmetric matrices store one triangle
struct SpTripletMat {
    int RowNum = 0, ColNum = 0;
    bool isSymmetric = false;
    std::vector<int> RowIndex, ColIndex;
    std::vector<double> MatVal;
    int EntryNum() const { return (int)MatVal.size(); }
};

// ---- the plug-in interface (include/sqphot/QPsolverInterface.hpp:43-194) ----
class QPSolverInterface {
public:
    virtual ~QPSolverInterface() = default;
    virtual void optimizeQP(std::shared_ptr<Stats> stats) = 0;
    virtual void optimizeLP(std::shared_ptr<Stats> stats) = 0;
    virtual double *get_optimal_solution() = 0;
    virtual double get_obj_value() = 0;
    virtual double *get_multipliers_bounds() = 0;
    virtual double *get_multipliers_constr() = 0;
    virtual void get_working_set(ActiveType *W_constr, ActiveType *W_bounds) = 0;
    virtual Exitflag get_status() = 0;
    virtual bool test_optimality(ActiveType *W_c = nullptr, ActiveType *W_b = nullptr) = 0;
    virtual OptimalityStatus get_optimality_status() = 0;
    virtual void set_lb(int location, double value) = 0;
    virtual void set_ub(int location, double value) = 0;
    virtual void set_lbA(int location, double value) = 0;
    virtual void set_ubA(int location, double value) = 0;
    virtual void set_g(int location, double value) = 0;
    virtual void set_lb(std::shared_ptr<const Vector> rhs) = 0;
    virtual void set_ub(std::shared_ptr<const Vector> rhs) = 0;
    virtual void set_lbA(std::shared_ptr<const Vector> rhs) = 0;
    virtual void set_ubA(std::shared_ptr<const Vector> rhs) = 0;
    virtual void set_g(std::shared_ptr<const Vector> rhs) = 0;
    virtual void set_H(std::shared_ptr<const SpTripletMat> rhs) = 0;
    virtual void set_A(std::shared_ptr<const SpTripletMat> rhs, IdentityInfo I_info) = 0;
    virtual void reset_constraints() = 0;
};

class HipQPInterface : public QPSolverInterface {
public:
    // qpOASESInterface(NLPInfo, QPType, Options) -- src/qpOASESInterface.cpp:35-50
    HipQPInterface(NLPInfo nlp_info, QPType qptype, std::shared_ptr<const Options> options, int device = -1);
    // plain-QP form (src/qpOASESInterface.cpp:54-94): sizes only; matrices via set_*_csc
    HipQPInterface(int nVar_QP, int nConstr_QP, std::shared_ptr<const Options> options, int device = -1);
    ~HipQPInterface() override;
    HipQPInterface(const HipQPInterface &) = delete;
    HipQPInterface &operator=(const HipQPInterface &) = delete;

    void set_A_csc(const int *jc, const int *ir, const double *val);
    void set_H_csc(const int *jc, const int *ir, const double *val);

    void optimizeQP(std::shared_ptr<Stats> stats) override;
    void optimizeLP(std::shared_ptr<Stats> stats) override;
    double *get_optimal_solution() override { return x_qp_.values(); }
    double get_obj_value() override;
    double *get_multipliers_bounds() override { return y_qp_.values(); }
    double *get_multipliers_constr() override { return y_qp_.values() + nVar_QP_; }
    void get_working_set(ActiveType *W_constr, ActiveType *W_bounds) override;
    Exitflag get_status() override;
    bool test_optimality(ActiveType *W_c = nullptr, ActiveType *W_b = nullptr) override;
    OptimalityStatus get_optimality_status() override { return qpOptimalStatus_; }
    void set_lb(int location, double value) override;
    void set_ub(int location, double value) override;
    void set_lbA(int location, double value) override;
    void set_ubA(int location, double value) override;
    void set_g(int location, double value) override;
    void set_lb(std::shared_ptr<const Vector> rhs) override;
    void set_ub(std::shared_ptr<const Vector> rhs) override;
    void set_lbA(std::shared_ptr<const Vector> rhs) override;
    void set_ubA(std::shared_ptr<const Vector> rhs) override;
    void set_g(std::shared_ptr<const Vector> rhs) override;
    void set_H(std::shared_ptr<const SpTripletMat> rhs) override;
    void set_A(std::shared_ptr<const SpTripletMat> rhs, IdentityInfo I_info) override;
    void reset_constraints() override;

    int nVar_QP() const { return nVar_QP_; }
    int nConstr_QP() const { return nConstr_QP_; }

private:
    void check(int rc, const char *what) const;
    void fetch_solution();
    int nVar_QP_, nConstr_QP_;
    std::shared_ptr<const Options> options_;
    rsqp_solver *solver_ = nullptr;
    Vector x_qp_, y_qp_;
    OptimalityStatus qpOptimalStatus_;
};

}  // namespace rsqp
