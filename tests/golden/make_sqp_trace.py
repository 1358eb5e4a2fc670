"""Regenerates tests/golden/sqp_traces.json: whole SQP trajectories of hs071, hs035 and hs065 (analytic evaluators
restartsqp_amd.problems.hs071_nlp / hs035_nlp / hs065_nlp, the minimal driver tests/sqp_driver.py) with the ORACLE behind the restated
optimizeQP dispatch as the QP solver. Per QP solve: the iterate (x_k, lambda_k), delta, rho, the dirty flags handed to the
boundary, and the oracle's answer (dispatch mode, nWSR, status, x, y, working sets). Inputs for the GPU replay test and
for bench.py's trajectory batch -- not outputs of the reference (qpOASES is not available: "parity unpinned")."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O  # noqa: E402
import sqp_driver as D  # noqa: E402
from restartsqp_amd import problems  # noqa: E402


class Recording(D.OracleBackend):
    def solve(self, qp, flags):
        a = super().solve(qp, flags)
        self.log.append(dict(mode=a["mode"], ws_b=a["ws_b"].tolist(), ws_c=a["ws_c"].tolist(), obj=a["obj"]))
        return a


def main():
    out = {}
    for name, fn in (("hs071", problems.hs071_nlp), ("hs035", problems.hs035_nlp), ("hs065", problems.hs065_nlp)):
        be = Recording(O)
        be.log = []
        x, f, it, trace = D.run_sqp(fn, be, name)
        for t, extra in zip(trace, be.log):
            t.update(extra)
        out[name] = dict(x_star=x.tolist(), f_star=f, iterations=it, qps=trace)
        print(name, "f* =", f, "iterations", it, "QPs", len(trace))
    with open(os.path.join(ROOT, "tests/golden/sqp_traces.json"), "w") as fh:
        json.dump(out, fh, indent=0)


if __name__ == "__main__":
    main()
