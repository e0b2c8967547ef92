"""Writes a file in the format of oracle/ref_harness/ref_dump.cpp's output -- "k" (result pose of the k-iteration run), "s"
(one per ceres::Solve), "i" (one per minimiser iteration) and "x" (state after the iteration) lines -- from the C ORACLE's own
run of a golden case.  NOT a pin: it exists so that the reader and the comparator of tests/test_golden_ref.py are exercised (and
shown to have teeth) in an image where the reference itself cannot be built; the real files come from oracle/ref_harness/run.sh."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def write(case_npz, out_path, terminating_iteration_listed=True):
    """terminating_iteration_listed: whether the iteration that ends a Solve through the parameter / function tolerance appears in
    Solver::Summary::iterations (the restatement cannot tell from here: the comparator accepts both)."""
    from oracle import binding as ob
    from test_golden import load_case
    z, cfg, _ = load_case(case_npz)
    n_outer = int(z["n_outer"])
    with open(out_path, "w") as f:
        f.write("# %s: written by tests/golden_ref_tools/oracle_as_ref.py from the C oracle (format exercise, not the reference)\n"
                % os.path.basename(case_npz)[:-4])
        for k in range(1, n_outer + 1):
            O = ob.Oracle(ob.make_config(**dict(cfg, max_iterations=k)))
            for kind in range(4):
                O.set_source(kind, z[f"src{kind}"]); O.set_target(kind, z[f"tgt{kind}"])
            rc, T, st = O.scan_match(z["T_pred"])
            assert rc == 0
            tr = O.get_trace()
            for j in sorted({r["solve"] for r in tr}):
                rows = [r for r in tr if r["solve"] == j]
                if not terminating_iteration_listed and rows[-1]["exit_kind"] in (1, 2):
                    rows = rows[:-1]
                ok = sum(r["step_ok"] for r in rows)
                f.write("s %d %d initial_cost %.17g final_cost %.17g successful %d unsuccessful %d termination 0\n"
                        % (k, j, rows[0]["cost"], rows[-1]["cost"], ok, len(rows) - ok))
                for r in rows:
                    f.write("i %d %d %d cost %.17g change %.17g step_ok %d radius %.17g step_norm %.17g rel %.17g gmax %.17g valid %d\n"
                            % (k, j, r["iteration"], r["cost"], r["cost_change"], r["step_ok"], r["radius"], r["step_norm"],
                               r["relative_decrease"], r["gradient_max_norm"], 0 if r["exit_kind"] == 3 else 1))
                for r in rows:
                    f.write("x %d %d %d %s rel %.17g gmax %.17g\n" % (k, j, r["iteration"], " ".join("%.17g" % v for v in r["x"]),
                                                                      r["relative_decrease"], r["gradient_max_norm"]))
            f.write("k %d %s\n" % (k, " ".join("%.17g" % v for v in np.asarray(T).reshape(-1))))
    return n_outer


if __name__ == "__main__":
    write(sys.argv[1], sys.argv[2])
