"""Writes the section of profiles/README.md for one round's files FROM the files (so the README cannot drift from what is
there): usage  profiles_readme.py <tag> <dir>  -> markdown on stdout.  Called by scripts/gpu_profile_r0N.sh."""
import csv, glob, json, os, sys
tag, d = sys.argv[1], sys.argv[2]
WHAT = {
    "bench_default.json": "the line `python bench.py` printed (no profiler attached)",
    "bench_default_under_rocprof.json": "the bench line of the profiled run (host launch rate limited by the profiler: frame times inflated)",
    "bench_default_kernel_stats.csv": "`rocprofv3 --kernel-trace --stats` of the same command, per kernel (`scripts/rocpd_stats.py`) -- ALL workloads of the bench mixed; the roofline kernel alone is in the two files below",
    "k3_prebuilt_kernel_stats.csv": "`rocprofv3 --kernel-trace --stats -- python scripts/k3_only.py 50 1`: K3 (`k3_accumulate<false, false>`) ALONE on the contract's pre-built 760k:200k:40k set (74.88 MB per sweep, Infinity-Cache resident across launches) -- `roofline.l3_resident` is recomputed from this row: 74 880 000 B / avg duration / 8 TB/s",
    "k3_cold_kernel_stats.csv": "the same with `k3_only.py 50 4`: four times the set (299.52 MB per sweep > 256 MiB Infinity Cache, every byte from HBM) -- the headline `roofline.frac`: 299 520 000 B / avg duration / 8 TB/s",
    "pmc_k3_prebuilt.json": "separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of the scale-1 run (`scripts/pmc_summary.py`, FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md); quoted by the bench line as `roofline.l3_resident.traffic`",
    "pmc_k3_prebuilt_cold.json": "the same for the cold set; quoted as `roofline.traffic`",
    "kitti_frame_timeline.txt": "`scripts/frame_timeline.py` on a kernel trace of `bench.py --workload kitti`: the launches of a KITTI-density frame in order, median duration and gap before each",
    "m1_frame_timeline.txt": "the same for the 1 M frame (`bench.py --workload m1`)",
    "solve_all_timeline.txt": "`scripts/solve_profile2.py` on a `-DTLOAM_STEP_PROFILE` build: wall-clock stamps (10 ns) inside the one-launch Solve `k_solve_all` per GN iteration -- lead block's stepper wave and one other wave",
    "kitti_sequence_4540.json": "`python bench.py --kitti-frames 4540 --no-m1 --no-cpu-baseline` (`scripts/gpu_seq4540.sh`): the KITTI-density sequence over the WHOLE published KITTI-00 trajectory of the reference (SURVEY 8(d) config 2); numbers below",
    "bench_gpus2_one_device.json": "`bench.py --gpus 2` launched as the driver launches it (`torch.distributed.run`, one rank per process) with `TLOAM_BENCH_ONE_DEVICE=1` -- every rank on the ONE GPU of the box, launcher collectives over gloo.  NOT a scaling measurement (the ranks share the GPU): the replica headline aggregating over the ranks, and the SELF-VERIFYING sharded 1 M frame (BASELINE.json configs[3]): every exchange form against the one-rank solve of the same frame (`sharded_1m_verified`, `sharded_1m_pose_delta`, ranks bit-identical, counters equal) in flat top-level scalars, run in a child process of every rank (`sharded_1m.ran_in`; more than four ranks on ONE device: in the ranks' own processes -- sixteen processes oversubscribe the device's hardware queues); RCCL refuses two ranks on one device and is reported as such (`rccl_nranks` null)",
    "bench_gpus4_one_device.json": "the same with 4 ranks",
    "bench_gpus8_one_device.json": "the same with 8 ranks",
    "pmc_sq_k_solve_all.json": "`scripts/gpu_pmc_kitti.sh`: one `rocprofv3 --pmc` pass (SQ counters, `--kernel-trace` only) over `bench.py --workload kitti`, per-launch means for the one-launch Solve: `SQ_WAVE_CYCLES` / `SQ_WAIT_ANY` / `SQ_ACTIVE_INST_ANY` / `SQ_WAIT_INST_ANY` count quad-cycles summed over the launch's 64 waves -- 70 % parked (block barrier, row polls), 27 % issuing, 2.6 % issue stalls; 4.6 k VALU instructions per wave per launch",
    "pmc_sq_k_build_sorted.json": "the same pass, the sixteen-lane correspondence search `k_build_sorted<16>` (2432 one-wave workgroups): 45 % parked on memory, 35 % issuing, 20 % issue stalls",
    "pmc_sq_k_grid_count_all.json": "the same pass, the first launch of the grid build (histogram atomics + the frame's AoS -> SoA start): 94 % parked",
    "cu_mask_ROC_GLOBAL_CU_MASK.json": "`tests/test_gpu_cu_mask.py`: the golden pairs, a KITTI-density frame, the 1 M frame and the submap on 32 of 256 CUs (`ROC_GLOBAL_CU_MASK=0xffffffff`, set before the HIP runtime starts) against the full chip: what `hipDeviceAttributeMultiprocessorCount` said, seconds per workload on either chip; results equal, no bounded wait ran out",
    "cu_mask_HSA_CU_MASK.json": "the same with `HSA_CU_MASK=0:0-31` (the ROCr spelling: the attribute does NOT see this mask, so the launch plan is the full chip's and the results are the same bits)",
    "solve_small_timeline.txt": "the same with `TLOAM_SOLVE_V1=1` (round 3's `k_solve_small`: one consumer wave for the whole grid)",
}
print("## Round %s\n" % tag[1:].lstrip("0"))
print("Written by `scripts/gpu_profile_%s.sh` (this section by `scripts/profiles_readme.py` from the files themselves).\n" % tag)
print("| file | what |\n|---|---|")
for p in sorted(glob.glob(os.path.join(d, tag + "_*"))):
    name = os.path.basename(p)
    print("| `%s` | %s |" % (name, WHAT.get(name[len(tag) + 1:], "(see the script)")))
def k3_row(path):
    try:
        for row in csv.DictReader(open(path)):
            nm = row.get("name") or row.get("Name") or ""
            if "k3_accumulate<false, false" in nm:   # (<false, false, 8>: the wide blocks of round 6)
                return row
    except OSError:
        pass
    return None
print()
for name, alg in (("k3_prebuilt_kernel_stats.csv", 74880000.0), ("k3_cold_kernel_stats.csv", 299520000.0)):
    row = k3_row(os.path.join(d, "%s_%s" % (tag, name)))
    if row:
        try:
            avg_us = float(row["AverageNs"]) / 1e3
            print("`%s_%s`: `k3_accumulate<false, false>` avg %.3f us over %s dispatches -> %.0f B / %.3f us / 8 TB/s = **%.3f**"
                  % (tag, name, avg_us, row.get("calls") or row.get("Calls") or "?", alg, avg_us, alg / (avg_us * 1e-6) / 8e12))
        except (ValueError, KeyError):
            print("`%s_%s`: %s" % (tag, name, row))
try:
    b = json.loads(open(os.path.join(d, tag + "_bench_default.json")).read().strip().splitlines()[-1])
    r = b["roofline"]
    print("\nBench line of the same run: value %.1f GN iter/s, ms_per_step %.4f; roofline.frac %.4f (avg %.3f us), l3_resident %.4f (avg %.3f us)."
          % (b["value"], b["ms_per_step"], r["frac"], r["avg_launch_us"], r["l3_resident"]["frac"], r["l3_resident"]["avg_launch_us"]))
except Exception as e:  # noqa: BLE001
    print("\n(bench line not readable: %r)" % (e,))
try:
    # the sweep INSIDE the 1 M frame and the GN iteration around it, from the kernel trace of the frame (the bench line's
    # roofline.in_frame_frac / gn_iteration_frac come from HIP event pairs / the device's clock in an unprofiled run)
    rows = []
    for ln in open(os.path.join(d, tag + "_m1_frame_timeline.txt")):
        f = ln.split()
        if len(f) > 5 and f[0].isdigit() and "dur" in f:
            rows.append((" ".join(f[1:f.index("dur")]), float(f[f.index("dur") + 1]), float(f[f.index("gap_before") + 1])))
    k3 = [r[1] for r in rows if r[0].startswith("k3_accumulate")]
    periods = [rows[i][1] + rows[i][2] + rows[i + 1][1] + rows[i + 1][2] for i in range(len(rows) - 2)
               if rows[i][0].startswith("k3_accumulate") and rows[i + 1][0].startswith("k_reduce_and_step") and rows[i + 2][0].startswith("k3_accumulate")]
    byt = b["roofline"]["in_frame"]["algorithmic_bytes_per_launch"]
    k3m, pm = sum(k3) / len(k3), sum(periods) / len(periods)
    print("\n`%s_m1_frame_timeline.txt`: %d sweeps of a frame, mean of the medians %.2f us -> %.0f B / %.2f us / 8 TB/s = **%.3f** in the frame "
          "(bench: in_frame_frac %.3f by event pairs, which add ~1 us); sweep + fold / step with its successor's gap, %d periods: %.2f us -> "
          "**%.3f** of the HBM peak per GN iteration (bench: gn_iteration_us %.2f by the device's clock, gn_iteration_frac %.3f)."
          % (tag, len(k3), k3m, byt, k3m, byt / (k3m * 1e-6) / 8e12, b["roofline"]["in_frame_frac"], len(periods), pm,
             byt / (pm * 1e-6) / 8e12, b["roofline"]["gn_iteration_us"], b["roofline"]["gn_iteration_frac"]))
    rs = b["roofline"]
    print("\nRead-stream ceiling measured by the same bench run (`tloam_time_read_stream`): %.0f GB/s from HBM, %.0f GB/s Infinity-Cache "
          "resident; the cold sweep is at %.3f of the former, the contract set at %.3f of the latter."
          % (rs["read_stream_GBps"], rs["read_stream_l3_GBps"], rs["frac_of_read_stream"], rs["config3_frac_of_read_stream_l3"]))
except Exception as e:  # noqa: BLE001
    print("\n(1 M frame timeline not readable: %r)" % (e,))
try:
    q = json.loads(open(os.path.join(d, tag + "_kitti_sequence_4540.json")).read().strip().splitlines()[-1])["kitti_sequence"]
    print("\n`%s_kitti_sequence_4540.json`: %d frames, %.4f ms/frame mean / %.4f p50 / %.4f p99, %.1f GN iter/s, %.2f GN iterations per frame; "
          "set_source + scan_match %.4f ms, set_target %.4f ms; pose error against the generator %.2f mm mean / %.2f mm max."
          % (tag, q["frames"], q["ms_per_frame"], q["ms_per_frame_p50"], q["ms_per_frame_p99"], q["gn_iters_per_sec"], q["gn_iters_per_frame"],
             q["per_call"]["set_source_plus_scan_match_ms"], q["per_call"]["set_target_ms"], q["pose_err_vs_truth_m"]["mean"] * 1e3,
             q["pose_err_vs_truth_m"]["max"] * 1e3))
except Exception:  # noqa: BLE001
    pass
