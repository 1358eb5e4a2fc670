"""Enums / structs of the reference that cross the QP boundary
(``include/sqphot/Types.hpp:36-128``, ``include/sqphot/Utils.hpp:35-37``)."""
from dataclasses import dataclass, field
from typing import List

INF = 1.0e18          # Utils.hpp:35
m_eps = 1.0e-16       # Utils.hpp:36
sqrt_m_eps = 1.0e-8   # Utils.hpp:37

# ActiveType (Types.hpp:84-89)
ACTIVE_ABOVE, ACTIVE_BELOW, ACTIVE_BOTH_SIDE, INACTIVE = 1, -1, -99, 0
# Exitflag (Types.hpp:51-73), QP part
QP_OPTIMAL = 20
QPERROR_INTERNAL_ERROR, QPERROR_INFEASIBLE, QPERROR_UNBOUNDED, QPERROR_EXCEED_MAX_ITER = 21, 22, 23, 24
QPERROR_NOTINITIALISED, QPERROR_PREPARINGAUXILIARYQP, QPERROR_AUXILIARYQPSOLVED = 25, 26, 27
QPERROR_PERFORMINGHOMOTOPY, QPERROR_HOMOTOPYQPSOLVED, QPERROR_UNKNOWN = 28, 29, 30
# QPType (Types.hpp:45-48)
LP, QP = 1, 2


@dataclass
class IdentityInfo:      # Types.hpp:36-42 (1-based positions)
    irow: List[int] = field(default_factory=list)
    jcol: List[int] = field(default_factory=list)
    size: List[int] = field(default_factory=list)
    value: List[float] = field(default_factory=list)

    @property
    def length(self):
        return len(self.irow)

    def blocks(self):
        return list(zip(self.irow, self.jcol, self.size, self.value))


@dataclass
class NLPInfo:           # Types.hpp:100-105
    nCon: int
    nVar: int
    nnz_jac_g: int = 0
    nnz_h_lag: int = 0


@dataclass
class OptimalityStatus:  # Types.hpp:107-119
    primal_feasibility: bool = False
    primal_violation: float = 0.0
    dual_feasibility: bool = False
    dual_violation: float = 0.0
    complementarity: bool = False
    compl_violation: float = 0.0
    stationarity: bool = False
    stationarity_violation: float = 0.0
    first_order_opt: bool = False
    KKT_error: float = 0.0
    Second_order_opt: bool = False


@dataclass
class Options:           # the fields the adapter reads (src/Options.cpp:45,54,23)
    qp_maxiter: int = 1000
    lp_maxiter: int = 100
    qpPrintLevel: int = 0


@dataclass
class Stats:             # include/sqphot/Stats.hpp: only qp_iter crosses the boundary
    qp_iter: int = 0

    def qp_iter_addValue(self, n):
        self.qp_iter += int(n)


@dataclass
class SpTripletMat:
    """Host triplet matrix as handed to set_A / set_H: 1-based COO
    (``include/sqphot/SpTripletMat.hpp``); symmetric matrices store one triangle."""
    RowNum: int
    ColNum: int
    RowIndex: list
    ColIndex: list
    MatVal: list
    isSymmetric: bool = False

    @property
    def EntryNum(self):
        return len(self.MatVal)


class QP_NOT_OPTIMAL(Exception):        # QPsolverInterface.hpp:26
    pass


class LP_NOT_OPTIMAL(Exception):        # :28
    pass


class QP_INTERNAL_ERROR(Exception):     # :30
    pass


class INVALID_WORKING_SET(Exception):   # :32
    pass
