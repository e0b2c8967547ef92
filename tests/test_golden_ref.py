"""The pin, when it exists.  tests/golden_ref/case_*.ref.txt are the result poses of the REFERENCE's own
LocalRegistration::scanMatching (built against real Eigen / Ceres 2.0 / Open3D 0.12 by oracle/ref_harness, run with
max_iterations = 1..n_outer on the inputs of the committed golden cases).  They cannot be produced in the round's image
-- none of those packages exists here -- so this module skips until someone with the toolchain drops the files in; then
the C oracle and the golden vectors are held against the real thing: pose after every outer GNC iteration within the
north-star tolerance (|dt| < 1e-6 m, |dR| < 1e-6 rad).

CPU (always): the exporter writes the harness's inputs, the harness sources reference the reference, and the CMake
recipe refuses to configure without the real packages (no stand-ins)."""
import glob
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from conftest import pose_delta

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = sorted(glob.glob(os.path.join(ROOT, "tests", "golden_ref", "case_*.ref.txt")))


def _read_ref(path):
    out = {}
    for line in open(path):
        if line.startswith("k "):
            v = line.split()
            out[int(v[1])] = np.array(v[2:18], float).reshape(4, 4)
    return out


def _read_solves(path):
    """Every ceres::Solve call as ref_dump's interposer wrote it: {(run k, call j): {"initial_cost", "final_cost", "successful",
    "unsuccessful", "iters": {iteration: {cost, change, step_ok, radius, step_norm, rel, gmax}}, "states": {iteration: x[6]}}}
    ("i" lines: the Solver::Summary's IterationSummary list; "x" lines: the state after the iteration, from the callback)."""
    out = {}
    for line in open(path):
        v = line.split()
        if line.startswith("s "):
            out[(int(v[1]), int(v[2]))] = {"iters": {}, "states": {}, "initial_cost": float(v[4]), "final_cost": float(v[6]),
                                           "successful": int(v[8]), "unsuccessful": int(v[10])}
        elif line.startswith("i "):
            f = dict(zip(v[4::2], v[5::2]))
            out[(int(v[1]), int(v[2]))]["iters"][int(v[3])] = {
                "cost": float(f["cost"]), "change": float(f["change"]), "step_ok": int(f["step_ok"]), "radius": float(f["radius"]),
                "step_norm": float(f["step_norm"]), "rel": float(f.get("rel", "nan")), "gmax": float(f.get("gmax", "nan"))}
        elif line.startswith("x "):
            n = v.index("rel") if "rel" in v else len(v)
            out[(int(v[1]), int(v[2]))]["states"][int(v[3])] = np.array(v[4:n], float)
    return out


def _compare_solves(solves, run, trace, label):
    """The Solves of run `run` (max_iterations = run) of the reference against a per-iteration trace of the SAME frame stepped
    through a restatement (oracle.binding.Oracle.get_trace): where parity is actually won or lost -- the
    accept / reject bookkeeping, the radius schedule and the state after EVERY minimiser iteration.
    Two things about Ceres' bookkeeping are recalled, not re-read (SURVEY 8(c) provenance caveat), and are accepted either way:
    whether iteration 0 counts as a successful step in the Summary, and whether the iteration that ends a Solve through the
    parameter / function tolerance is listed in Summary::iterations."""
    calls = sorted(j for (k, j) in solves if k == run)
    assert calls, (label, "no Solve recorded for run", run)
    for j in calls:
        S = solves[(run, j)]
        rows = {r["iteration"]: r for r in trace if r["solve"] == j}
        assert rows, (label, "the restatement ran fewer Solves", j)
        last = max(rows)
        n_ref = max(S["iters"]) + 1
        assert n_ref in ((last + 1), last + (0 if rows[last]["exit_kind"] in (1, 2) else 1)), (label, "iterations", j, n_ref, last + 1)
        for i, it in sorted(S["iters"].items()):
            r = rows[i]
            assert abs(it["cost"] - r["cost"]) <= 1e-9 * abs(it["cost"]) + 1e-300, (label, "cost", j, i, it["cost"], r["cost"])
            assert abs(it["radius"] - r["radius"]) <= 1e-12 * it["radius"], (label, "radius", j, i, it["radius"], r["radius"])
            if i >= 1:
                assert it["step_ok"] == r["step_ok"], (label, "accept / reject", j, i)
                assert abs(it["change"] - r["cost_change"]) <= 1e-7 * abs(it["cost"]) + 1e-9 * abs(it["change"]), (label, "cost change", j, i)
            if i in S["states"]:
                assert np.max(np.abs(S["states"][i] - r["x"])) < 1e-9, (label, "state", j, i, S["states"][i], r["x"])
        assert abs(S["final_cost"] - rows[min(last, n_ref - 1)]["cost"]) <= 1e-9 * abs(S["final_cost"]), (label, "final cost", j)


def _check_case_against(path, runner, label):
    """runner(cfg_with_max_iterations, z) -> (T 4x4, per-iteration trace or None): every run k = 1..n_outer of the reference
    against the runner's: result pose within the north-star tolerance, the Solves state by state when a trace is there."""
    from test_golden import load_case
    z, cfg, _ = load_case(os.path.join(ROOT, "tests", "golden", os.path.basename(path)[:-8] + ".npz"))
    ref = _read_ref(path)
    solves = _read_solves(path)
    assert ref, (label, "no result poses in", path)
    for k, T_ref in sorted(ref.items()):
        T, trace = runner(dict(cfg, max_iterations=k), z)
        dt, dr = pose_delta(T, T_ref)
        assert dt < 1e-6 and dr < 1e-6, (label, k, dt, dr)
        if trace is not None and any(kk == k for (kk, _) in solves):
            _compare_solves(solves, k, trace, "%s run %d" % (label, k))
    return z, cfg, ref


def _oracle_runner(cfg, z):
    from oracle import binding as ob
    O = ob.Oracle(ob.make_config(**cfg))
    for kind in range(4):
        O.set_source(kind, z[f"src{kind}"]); O.set_target(kind, z[f"tgt{kind}"])
    rc, T, st = O.scan_match(z["T_pred"])
    assert rc == 0
    return T, O.get_trace()


NO_REF = pytest.mark.skipif(not REF, reason="tests/golden_ref/ is empty: the reference cannot be built in this image "
                                            "(oracle/ref_harness: Eigen3, Ceres 2.0, Open3D 0.12, yaml-cpp, ROS; one command "
                                            "with docker and a network: oracle/ref_harness/run.sh)")


@NO_REF
@pytest.mark.parametrize("path", REF or ["-"], ids=[os.path.basename(p)[5:-8] for p in REF] or ["none"])
def test_oracle_and_golden_match_the_reference(path):
    from tloam_amd import synth
    z, cfg, ref = _check_case_against(path, _oracle_runner, "oracle")
    for k, T_ref in sorted(ref.items()):   # the committed golden vector: the pose after outer iteration k
        dt, dr = pose_delta(synth.se3_exp_np(z[f"it{k - 1}_x"]), T_ref)
        assert dt < 1e-6 and dr < 1e-6, ("golden", k, dt, dr)


@NO_REF
@pytest.mark.gpu
@pytest.mark.parametrize("path", REF or ["-"], ids=[os.path.basename(p)[5:-8] for p in REF] or ["none"])
def test_hip_path_matches_the_reference(path, hip_module):
    """The PRODUCT against the reference itself: tloam_scan_match through the C ABI on the inputs the reference was run on, with
    max_iterations = 1..n_outer -- the pose after every outer GNC iteration within the north-star tolerance -- and the
    minimiser's counters of the full run against the reference's Solver::Summary of every Solve."""
    reg = hip_module
    last = {}

    def hip_runner(cfg, z):
        H = reg.HipRegistration(reg.default_config(**cfg))
        for kind in range(4):
            H.set_source(kind, z[f"src{kind}"]); H.set_target(kind, z[f"tgt{kind}"])
        rc, T, st = H.scan_match(z["T_pred"])
        H.close()
        assert rc == 0
        last.update(st=st, k=cfg["max_iterations"])
        return T, None

    z, cfg, ref = _check_case_against(path, hip_runner, "hip")
    solves = _read_solves(path)
    n_outer = max(ref)
    calls = sorted(j for (k, j) in solves if k == n_outer)
    if calls and last.get("k") == n_outer:
        # (the device keeps no per-iteration trace; its counters over the whole frame against the reference's Summaries)
        it_ref = sum(max(solves[(n_outer, j)]["iters"]) for j in calls)
        ok_ref = sum(sum(v["step_ok"] for i, v in solves[(n_outer, j)]["iters"].items() if i >= 1) for j in calls)
        assert last["st"]["accepted_steps"] == ok_ref, ("accepted steps", last["st"]["accepted_steps"], ok_ref)
        assert last["st"]["gn_iterations"] in (it_ref, it_ref + len(calls)), ("iterations", last["st"]["gn_iterations"], it_ref)


def test_ref_file_reader_and_comparator_have_teeth(tmp_path):
    """No reference here, so the reader and the comparator are exercised on a file of the SAME FORMAT written from the C oracle's
    own run (tests/golden_ref_tools/oracle_as_ref.py -- a format exercise, not a pin): it must pass against the oracle, in both
    conventions for the terminating iteration, and a file with one rejected step turned into an accepted one, a radius off by
    a factor of two or a state off by 1e-6 must FAIL."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden_ref_tools"))
    import oracle_as_ref
    case = os.path.join(ROOT, "tests", "golden", "case_street_seed0.npz")
    for listed in (True, False):
        path = str(tmp_path / ("case_street_seed0.ref.txt" if listed else "case_street_seed0.ref.txt"))
        n_outer = oracle_as_ref.write(case, path, terminating_iteration_listed=listed)
        solves = _read_solves(path)
        assert len(_read_ref(path)) == n_outer and any(S["states"] for S in solves.values())
        _check_case_against(path, _oracle_runner, "self")
    text = open(path).read().split("\n")
    rejected = next(i for i, l in enumerate(text) if l.startswith("i ") and " step_ok 0 " in l)

    def broken(edit):
        lines = list(text)
        edit(lines)
        open(path, "w").write("\n".join(lines))
        with pytest.raises(AssertionError):
            _check_case_against(path, _oracle_runner, "broken")

    broken(lambda L: L.__setitem__(rejected, L[rejected].replace(" step_ok 0 ", " step_ok 1 ")))

    def halve_radius(L):
        v = L[rejected].split(); i = v.index("radius"); v[i + 1] = repr(float(v[i + 1]) * 0.5); L[rejected] = " ".join(v)
    broken(halve_radius)

    def shift_state(L):
        i = next(i for i, l in enumerate(L) if l.startswith("x ") and l.split()[3] == "1")
        v = L[i].split(); v[4] = repr(float(v[4]) + 1e-6); L[i] = " ".join(v)
    broken(shift_state)


def test_exporter_writes_the_harness_inputs(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden_ref_tools"))
    import export_ref_inputs as ex
    names = ex.main(str(tmp_path))
    assert len(names) >= 7 and "case_small_rotation_perturbed" not in names   # the Random() case cannot be pinned
    assert {"case_kitti_caps", "case_kitti_caps_no_sphere"} <= set(names)     # the size the metric is quoted on IS exportable
    big = open(tmp_path / "case_kitti_caps.bin", "rb").read()
    assert np.frombuffer(big[4 + 256: 4 + 264], np.int64)[0] > 2500            # more planar source points than the cap
    raw = open(tmp_path / (names[0] + ".bin"), "rb").read()
    z = np.load(os.path.join(ROOT, "tests", "golden", names[0] + ".npz"), allow_pickle=False)
    assert np.frombuffer(raw[:4], np.int32)[0] == int(z["n_outer"])
    assert np.array_equal(np.frombuffer(raw[4 + 128: 4 + 256], np.float64).reshape(4, 4), z["T_pred"])
    n0 = np.frombuffer(raw[4 + 256: 4 + 264], np.int64)[0]
    assert n0 == len(z["src0"])


def test_harness_compiles_the_reference_itself_and_needs_the_real_packages(tmp_path):
    src = open(os.path.join(ROOT, "oracle", "ref_harness", "ref_dump.cpp")).read()
    cm = open(os.path.join(ROOT, "oracle", "ref_harness", "CMakeLists.txt")).read()
    assert '#include "tloam/models/registration/registration.hpp"' in src and "tloam::LocalRegistration reg(" in src
    assert "RTLD_NEXT" in src and "summary->iterations" in src     # the Solve summaries come from an interposer, not from an edited reference
    assert "IterationCallback" in src and "update_state_every_iteration" in src   # ... and so does the state after every iteration
    dock = open(os.path.join(ROOT, "oracle", "ref_harness", "Dockerfile")).read()
    for need in ("ros:melodic", "ceres-solver", "2.0.0", "BUILD_SHARED_LIBS=ON", "v0.12.0", "zhoupengwei/tloam", "ref_dump"):
        assert need in dock, need
    assert os.access(os.path.join(ROOT, "oracle", "ref_harness", "run.sh"), os.X_OK)
    assert "${TLOAM_REFERENCE_DIR}/src/models/registration/registration.cpp" in cm
    for pkg in ("Eigen3", "Ceres", "Open3D", "yaml-cpp"):
        assert f"find_package({pkg}" in cm
    cmake = shutil.which("cmake")
    if not cmake or not os.path.isdir("/root/reference"):
        pytest.skip("cmake / the reference checkout are not on this machine")
    r = subprocess.run([cmake, "-S", os.path.join(ROOT, "oracle", "ref_harness"), "-B", str(tmp_path / "b"),
                        "-DTLOAM_REFERENCE_DIR=/root/reference"], capture_output=True, text=True)
    if r.returncode == 0:
        pytest.skip("this machine HAS the reference's toolchain: build oracle/_ref and generate tests/golden_ref/")
    assert any(p in (r.stderr + r.stdout) for p in ("Eigen3", "Ceres", "Open3D", "yaml-cpp", "roscpp")), r.stderr[-400:]
