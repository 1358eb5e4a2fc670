// HipQPInterface.hpp -- C++ host adapter: the QPSolverInterface subclass a RestartSQP
// maintainer compiles into the reference. Mirrors qpOASESInterface
// (reference include/sqphot/qpOASESInterface.hpp:37-262) method for method and forwards to the
// C ABI of include/rsqp_hip.h. Inside RestartSQP the types below are the reference's own
// (sqphot/Vector.hpp, SpTripletMat.hpp, Types.hpp, Ipopt's DECLARE_STD_EXCEPTION); this header
// carries minimal equivalents so that the adapter builds and is tested stand-alone.
#pragma once
#include <cstdarg>
#include <cstdio>
#include <functional>
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

// Ipopt's journal types as far as WriteQPDataToFile and the NLP ctor name them (IpJournalist.hpp).
// Inside RestartSQP these are Ipopt's own; stand-alone a journalist is just a sink for Printf.
namespace Ipopt {
enum EJournalLevel { J_NONE = 0, J_WARNING = 3, J_LAST_LEVEL = 13 };
enum EJournalCategory { J_DBG = 0, J_MAIN = 4, J_USER1 = 14 };
class Journalist {
public:
    explicit Journalist(FILE *sink = nullptr) : sink_(sink) {}
    void Printf(EJournalLevel, EJournalCategory, const char *fmt, ...) const {
        if (!sink_) return;
        va_list ap;
        va_start(ap, fmt);
        std::vfprintf(sink_, fmt, ap);
        va_end(ap);
    }
private:
    FILE *sink_;
};
template <class T> using SmartPtr = std::shared_ptr<T>;
}  // namespace Ipopt

enum Solver { QPOASES_SOLVER = 0, QORE_SOLVER = 1 };   // include/sqphot/Types.hpp:91-97 (the two with a dump layout)

// Harwell-Boeing matrix as the boundary sees it (include/sqphot/SpHbMat.hpp): compressed column
// (or row) pointers + indices + values + the triplet->position permutation `order`. What getA() /
// getH() hand to QPhandler::get_active_set (src/QPhandler.cpp:603-611), which calls times() on it:
// the products are forwarded to the device copy (rsqp_A_times ...), the arrays are host mirrors.
class SpHbMat {
public:
    typedef std::function<void(const double *, double *)> Product;
    SpHbMat(int RowNum, int ColNum, bool isCompressedRow)
        : RowNum_(RowNum), ColNum_(ColNum), isCompressedRow_(isCompressedRow) {}
    // plain-QP form: CSC arrays given by the caller (the reference wraps them without copying,
    // src/qpOASESInterface.cpp:77-90; here they are copied once to the device)
    SpHbMat(int RowNum, int ColNum, const int *jc, const int *ir, const double *val)
        : RowNum_(RowNum), ColNum_(ColNum), isCompressedRow_(false), ColIndex_(jc, jc + ColNum + 1),
          RowIndex_(ir, ir + jc[ColNum]), MatVal_(val, val + jc[ColNum]) {
        order_.resize(MatVal_.size());
        for (size_t i = 0; i < order_.size(); i++) order_[i] = (int)i;
    }
    int RowNum() const { return RowNum_; }
    int ColNum() const { return ColNum_; }
    int EntryNum() const { return (int)MatVal_.size(); }
    bool isCompressedRow() const { return isCompressedRow_; }
    const int *RowIndex() const { return RowIndex_.data(); }
    const int *ColIndex() const { return ColIndex_.data(); }
    const double *MatVal() const { return MatVal_.data(); }
    const int *order() const { return order_.data(); }
    int RowIndex(int i) const { return RowIndex_[i]; }
    int ColIndex(int i) const { return ColIndex_[i]; }
    double MatVal(int i) const { return MatVal_[i]; }
    // SpHbMat::times / transposed_times (src/SpHbMat.cpp:659-737)
    void times(std::shared_ptr<const Vector> p, std::shared_ptr<Vector> result) const {
        if (!times_) throw std::logic_error("SpHbMat::times: matrix is not bound to a device copy");
        times_(p->values(), result->values());
    }
    void transposed_times(std::shared_ptr<const Vector> p, std::shared_ptr<Vector> result) const {
        if (!ttimes_) throw std::logic_error("SpHbMat::transposed_times: matrix is not bound to a device copy");
        ttimes_(p->values(), result->values());
    }
    // SpHbMat::write_to_file, file form (src/SpHbMat.cpp:554-578)
    void write_to_file(const char *, Ipopt::SmartPtr<Ipopt::Journalist> jnlst, Ipopt::EJournalLevel level,
                       Ipopt::EJournalCategory category, Solver solver) const {
        const int nptr = (isCompressedRow_ ? RowNum_ : ColNum_) + 1, nnz = EntryNum();
        const std::vector<int> &ptr = isCompressedRow_ ? RowIndex_ : ColIndex_, &idx = isCompressedRow_ ? ColIndex_ : RowIndex_;
        if (solver == QORE_SOLVER) {   // pointers first
            for (int i = 0; i < nptr; i++) jnlst->Printf(level, category, "%d\n", ptr[i]);
            for (int i = 0; i < nnz; i++) jnlst->Printf(level, category, "%d\n", idx[i]);
        } else {                       // indices first
            for (int i = 0; i < nnz; i++) jnlst->Printf(level, category, "%d\n", idx[i]);
            for (int i = 0; i < nptr; i++) jnlst->Printf(level, category, "%d\n", ptr[i]);
        }
        for (int i = 0; i < nnz; i++) jnlst->Printf(level, category, "%23.16e\n", MatVal_[i]);
    }
private:
    friend class HipQPInterface;
    int RowNum_, ColNum_;
    bool isCompressedRow_;
    std::vector<int> ColIndex_, RowIndex_;   // compressed column: ColIndex_ = pointers (ColNum+1), RowIndex_ = row of each entry
    std::vector<double> MatVal_;
    std::vector<int> order_;
    Product times_, ttimes_;
};

// ---- the plug-in interface (include/sqphot/QPsolverInterface.hpp:43-194) ----
class QPSolverInterface {
public:
    virtual ~QPSolverInterface() = default;
    virtual const std::shared_ptr<Vector> &getLb() const = 0;
    virtual const std::shared_ptr<Vector> &getUb() const = 0;
    virtual const std::shared_ptr<Vector> &getLbA() const = 0;
    virtual const std::shared_ptr<Vector> &getUbA() const = 0;
    virtual const std::shared_ptr<Vector> &getG() const = 0;
    virtual std::shared_ptr<const SpHbMat> getH() const = 0;
    virtual std::shared_ptr<const SpHbMat> getA() const = 0;
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
    virtual void WriteQPDataToFile(Ipopt::EJournalLevel level, Ipopt::EJournalCategory category,
                                   const std::string filename) = 0;
};

class HipQPInterface : public QPSolverInterface {
public:
    // qpOASESInterface(NLPInfo, QPType, Options, Journalist) -- include/sqphot/qpOASESInterface.hpp:39-41,
    // src/qpOASESInterface.cpp:35-50
    HipQPInterface(NLPInfo nlp_info, QPType qptype, std::shared_ptr<const Options> options,
                   Ipopt::SmartPtr<Ipopt::Journalist> jnlst = nullptr, int device = -1);
    // plain-QP ctor with data (include/sqphot/qpOASESInterface.hpp:44-51, src/qpOASESInterface.cpp:54-94;
    // used by test/QPsolvers_testers.cpp:220): H and A in compressed-column form
    HipQPInterface(std::shared_ptr<SpHbMat> H, std::shared_ptr<SpHbMat> A, std::shared_ptr<Vector> g,
                   std::shared_ptr<Vector> lb, std::shared_ptr<Vector> ub, std::shared_ptr<Vector> lbA,
                   std::shared_ptr<Vector> ubA, std::shared_ptr<Options> options = nullptr, int device = -1);
    // sizes only; matrices via set_*_csc (not in the reference: used by the replay harness)
    HipQPInterface(int nVar_QP, int nConstr_QP, std::shared_ptr<const Options> options, int device = -1);
    ~HipQPInterface() override;
    HipQPInterface(const HipQPInterface &) = delete;
    HipQPInterface &operator=(const HipQPInterface &) = delete;

    void set_A_csc(const int *jc, const int *ir, const double *val);
    void set_H_csc(const int *jc, const int *ir, const double *val);

    // data getters (QPsolverInterface.hpp:47-59): adapter-owned host copies, refreshed on every call
    const std::shared_ptr<Vector> &getLb() const override { return refresh(RSQP_VEC_LB, lb_); }
    const std::shared_ptr<Vector> &getUb() const override { return refresh(RSQP_VEC_UB, ub_); }
    const std::shared_ptr<Vector> &getLbA() const override { return refresh(RSQP_VEC_LBA, lbA_); }
    const std::shared_ptr<Vector> &getUbA() const override { return refresh(RSQP_VEC_UBA, ubA_); }
    const std::shared_ptr<Vector> &getG() const override { return refresh(RSQP_VEC_G, g_); }
    std::shared_ptr<const SpHbMat> getH() const override;
    std::shared_ptr<const SpHbMat> getA() const override;

    void optimizeQP(std::shared_ptr<Stats> stats = nullptr) override;
    void optimizeLP(std::shared_ptr<Stats> stats = nullptr) override;
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
    // src/qpOASESInterface.cpp:791-814: file "qpOASES" + filename, lb lbA ub ubA g A H
    void WriteQPDataToFile(Ipopt::EJournalLevel level, Ipopt::EJournalCategory category,
                           const std::string filename) override;
    // the QORE layout of the same data (src/QOREInterface.cpp:582-598): file "QORE_" + filename
    void WriteQPDataToFileQORE(const std::string filename);

    int nVar_QP() const { return nVar_QP_; }
    int nConstr_QP() const { return nConstr_QP_; }

private:
    void check(int rc, const char *what) const;
    void fetch_solution();
    const std::shared_ptr<Vector> &refresh(int which, std::shared_ptr<Vector> &v) const;
    std::shared_ptr<const SpHbMat> mirror(bool isA) const;
    int nVar_QP_, nConstr_QP_;
    std::shared_ptr<const Options> options_;
    Ipopt::SmartPtr<Ipopt::Journalist> jnlst_;
    rsqp_solver *solver_ = nullptr;
    mutable std::shared_ptr<Vector> lb_, ub_, lbA_, ubA_, g_;
    Vector x_qp_, y_qp_;
    OptimalityStatus qpOptimalStatus_;
};

}  // namespace rsqp
