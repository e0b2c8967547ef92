#!/usr/bin/env python3
"""TEST INFRASTRUCTURE, NOT PRODUCT -- how much of the parity claim hangs on each recalled upstream behaviour.

SURVEY.md 8(c): what Ceres 2.0 / Open3D 0.12 / Eigen do inside the calls of registration.cpp is recalled, not re-read
(none of them is on this machine).  oracle_np.ASSUMED_UPSTREAM names every such recollection and puts it behind a switch;
this script flips ONE switch at a time to its most plausible alternative and runs the numpy restatement over

  golden    the eleven committed cases of tests/golden/case_*.npz
  kitti200  the first 200 frames of the synthetic KITTI-density sequence (bench.kitti_frame, BASELINE.json configs[1])

against the same inputs with every switch at its default, and reports per (switch, alternative) the largest pose difference
(|dt| in m, |dR| in rad), how many units moved by more than 1e-9 / 1e-6, and how many changed a counter (outer iterations,
minimiser evaluations / iterations / accepted steps, correspondence counts, early convergence).

A flip that moves no pose by 1e-6 is IRRELEVANT to the north-star claim (|dt| < 1e-6 m, |dR| < 1e-6 rad): whichever way the real
library behaves, the result is the same to the tolerance.  The others are what the pin has to settle; oracle/ref_harness/
ref_dump.cpp prints the quantity that decides each of them.

    python oracle/assumption_sensitivity.py [--frames 200] [--procs 8] [--out tests/golden/assumption_sensitivity.json]
    python oracle/assumption_sensitivity.py --markdown      # the table of DESIGN.md section 3 from the committed json
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import oracle_np as onp  # noqa: E402

COUNTERS = ("outer_iterations", "gn_evaluations", "gn_iterations", "accepted_steps", "n_corr", "converged_early")
OUT_DEFAULT = os.path.join(ROOT, "tests", "golden", "assumption_sensitivity.json")


def flips():
    return [(name, alt) for name, spec in onp.ASSUMED_UPSTREAM.items() for alt in spec[1]]


def pose_delta(A, B):
    D = np.linalg.inv(A) @ B
    R = D[:3, :3]
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) * 0.5
    return float(np.linalg.norm(D[:3, 3])), float(np.arctan2(np.linalg.norm(w), (np.trace(R) - 1.0) * 0.5))


def load_unit(unit):
    """unit = ("golden", case name) | ("kitti200", frame number) -> (clouds src[4], tgt[4], T_pred, cfg overrides, omega)"""
    from tloam_amd import synth
    if unit[0] == "golden":
        z = dict(np.load(os.path.join(ROOT, "tests", "golden", f"case_{unit[1]}.npz"), allow_pickle=False))
        if "scene_json" in z:
            sc = synth.make_scene(**json.loads(str(z["scene_json"])))
            src = [sc.source.cloud(k) for k in range(4)]
            tgt = [sc.target.cloud(k) for k in range(4)]
        else:
            src = [np.asarray(z[f"src{k}"], np.float64) for k in range(4)]
            tgt = [np.asarray(z[f"tgt{k}"], np.float64) for k in range(4)]
        omega = z["omega"] if z["omega"].size else None
        return src, tgt, z["T_pred"], json.loads(str(z["cfg_json"])), omega, z["T_result"]
    import bench
    sc = bench.kitti_frame(synth, 0, int(unit[1]))
    return [sc.source.cloud(k) for k in range(4)], [sc.target.cloud(k) for k in range(4)], sc.T_pred, {}, None, None


def run(src, tgt, T_pred, cfg, omega, assume):
    N = onp.NpRegistration(cfg, assume=assume)
    for k in range(4):
        N.set_source(k, src[k])
        N.set_target(k, tgt[k])
    T, st = N.scan_match(T_pred, omega=omega)
    return T, {k: (list(st[k]) if isinstance(st[k], (list, tuple)) else int(st[k])) for k in COUNTERS}


def unit_task(args):
    unit, only = args
    src, tgt, T_pred, cfg, omega, T_golden = load_unit(unit)
    T0, c0 = run(src, tgt, T_pred, cfg, omega, None)
    out = {"unit": list(unit), "default_vs_golden": None if T_golden is None else pose_delta(T0, T_golden), "flips": {}}
    for name, alt in flips():
        if only and name not in only:
            continue
        try:
            T1, c1 = run(src, tgt, T_pred, cfg, omega, {name: alt})
            dt, dr = pose_delta(T0, T1)
            out["flips"][f"{name}={alt}"] = dict(dt=dt, dr=dr, counters_changed=[k for k in COUNTERS if c0[k] != c1[k]])
        except Exception as e:   # an alternative that breaks the solve outright is itself a finding
            out["flips"][f"{name}={alt}"] = dict(dt=float("inf"), dr=float("inf"), counters_changed=["error: " + repr(e)])
    return out


def summarise(results, n_frames):
    table = {}
    for name, alt in flips():
        key = f"{name}={alt}"
        row = {"switch": name, "default": onp.ASSUMED_UPSTREAM[name][0], "alternative": alt,
               "upstream": onp.ASSUMED_UPSTREAM[name][2]}
        for group in ("golden", "kitti200"):
            rs = [r for r in results if r["unit"][0] == group and key in r["flips"]]
            if not rs:
                continue
            dts = [r["flips"][key]["dt"] for r in rs]
            drs = [r["flips"][key]["dr"] for r in rs]
            w = int(np.argmax(np.maximum(dts, drs)))
            row[group] = dict(units=len(rs), max_dt=max(dts), max_dr=max(drs), worst_unit=str(rs[w]["unit"][1]),
                              moved_1e9=sum(1 for a, b in zip(dts, drs) if max(a, b) >= 1e-9),
                              moved_1e6=sum(1 for a, b in zip(dts, drs) if max(a, b) >= 1e-6),
                              counters_changed=sum(1 for r in rs if r["flips"][key]["counters_changed"]))
        worst = max([max(row[g]["max_dt"], row[g]["max_dr"]) for g in ("golden", "kitti200") if g in row] or [0.0])
        row["relevant_to_1e-6_claim"] = bool(worst >= 1e-6)
        table[key] = row
    dvg = [r["default_vs_golden"] for r in results if r["default_vs_golden"] is not None]
    return {"tolerance": 1e-6, "kitti_frames": n_frames, "default_vs_committed_golden_max": [max(d[0] for d in dvg), max(d[1] for d in dvg)] if dvg else None,
            "rows": table}


def markdown(js):
    lines = ["| switch (default -> alternative) | upstream | golden: max dt / dR, moved >= 1e-6, counters changed | kitti200: max dt / dR, moved >= 1e-6, counters changed | matters at 1e-6? |",
             "|---|---|---|---|---|"]
    for key, r in js["rows"].items():
        def cell(g):
            if g not in r:
                return "-"
            c = r[g]
            return f"{c['max_dt']:.1e} m / {c['max_dr']:.1e} rad, {c['moved_1e6']}/{c['units']}, {c['counters_changed']}/{c['units']}"
        lines.append(f"| `{r['switch']}` ({r['default']} -> {r['alternative']}) | {r['upstream']} | {cell('golden')} | {cell('kitti200')} | "
                     f"{'**yes**' if r['relevant_to_1e-6_claim'] else 'no'} |")
    return "\n".join(lines)


def relevant_flips(js=None):
    """the (switch, alternative) pairs whose flip moved a pose by 1e-6 or more in the committed table"""
    js = js or json.load(open(OUT_DEFAULT))
    return [(r["switch"], r["alternative"]) for r in js["rows"].values() if r["relevant_to_1e-6_claim"]]


def poses_after_each_outer_iteration(unit, assume):
    """{k: 4x4 pose after k outer iterations} of one run (the GNC schedule of iteration k does not depend on max_iterations:
    the pose after k iterations is the result of the k-iteration run, which is what oracle/ref_harness/ref_dump.cpp records)"""
    src, tgt, T_pred, cfg, omega, _ = load_unit(unit)
    N = onp.NpRegistration(cfg, assume=assume)
    for k in range(4):
        N.set_source(k, src[k])
        N.set_target(k, tgt[k])
    trace = []
    N.scan_match(T_pred, omega=omega, trace=trace)
    return {t["iter"] + 1: onp.se3_exp(t["x"]) for t in trace}


def diagnose(ref_poses, unit, flips_to_try=None, tol=1e-6):
    """Which recalled behaviours are CONSISTENT with result poses of the real reference (tests/golden_ref/case_*.ref.txt "k"
    lines: {k: 4x4}) on golden case `unit`?  Returns {"default" | "switch=alt": (consistent?, worst dt, worst dR)}.  With the
    pin in hand this names the recollection to correct: the default must be consistent, every relevant alternative on a case
    that is sensitive to it must not be."""
    cands = [("default", None)] + [(f"{n}={a}", {n: a}) for n, a in (flips_to_try or relevant_flips())]
    out = {}
    for label, assume in cands:
        mine = poses_after_each_outer_iteration(unit, assume)
        wt = wr = 0.0
        for k, T_ref in ref_poses.items():
            T = mine.get(k, mine[max(mine)])      # (a run that converged early keeps its last pose)
            dt, dr = pose_delta(T, np.asarray(T_ref, float))
            wt, wr = max(wt, dt), max(wr, dr)
        out[label] = (bool(wt < tol and wr < tol), wt, wr)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--procs", type=int, default=max(1, (os.cpu_count() or 2)))
    ap.add_argument("--out", default=OUT_DEFAULT)
    ap.add_argument("--only", default="", help="comma-separated switch names")
    ap.add_argument("--small", action="store_true", help="golden cases that store their clouds only (what the CPU test re-runs)")
    ap.add_argument("--markdown", action="store_true")
    a = ap.parse_args()
    if a.markdown:
        print(markdown(json.load(open(a.out))))
        return
    names = sorted(os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(ROOT, "tests", "golden", "case_*.npz")))
    if a.small:
        names = [n for n in names if not n.startswith("kitti_")]
    units = [("golden", n) for n in names] + [("kitti200", f) for f in range(0 if a.small else a.frames)]
    only = set(x for x in a.only.split(",") if x)
    t0 = time.time()
    if a.procs > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(a.procs) as pool:
            results = []
            for i, r in enumerate(pool.imap_unordered(unit_task, [(u, only) for u in units], chunksize=1)):
                results.append(r)
                if (i + 1) % 10 == 0:
                    print(f"{i + 1}/{len(units)} units, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    else:
        results = [unit_task((u, only)) for u in units]
    results.sort(key=lambda r: (r["unit"][0], str(r["unit"][1]).zfill(6)))
    js = summarise(results, a.frames)
    js["per_unit"] = {f"{r['unit'][0]}:{r['unit'][1]}": {k: [v["dt"], v["dr"], v["counters_changed"]] for k, v in r["flips"].items()}
                      for r in results if r["unit"][0] == "golden"}
    with open(a.out, "w") as f:
        json.dump(js, f, indent=1)
    print(markdown(js))
    print(f"# {len(units)} units x {len(flips())} flips in {time.time() - t0:.0f} s -> {a.out}", file=sys.stderr)


if __name__ == "__main__":
    main()
