#!/bin/bash
# Round-6 profile set.  Everything the docs and the bench line quote comes out of THIS script and is copied into profiles/
# (of the repo copy on the GPU box -> gpurun_out/$TAG/profiles/ -> committed by hand), together with the section of
# profiles/README.md that describes the files (scripts/profiles_readme.py writes it from what is actually there).
#   1. single-workload kernel traces of the roofline kernel: K3 alone on the contract set (scale 1, Infinity-Cache resident) and on
#      the cold set (scale 4) -- rocprofv3 --kernel-trace --stats, one kernel name per CSV row, nothing else mixed in
#   2. PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, kernel-trace only alongside) of the same two runs
#   3. the default bench line, and the kernel trace of the same command
#   4. KITTI-density frame timeline, 1 M frame timeline, in-kernel timeline of the one-launch Solve (k_solve_all; needs the stepprof
#      variant build: python -m tloam_amd.build --variant stepprof --units tl_gn.hip -- -DTLOAM_STEP_PROFILE)
#   5. the GPU suite first (a profile of a tree whose tests fail is worth nothing)
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; P=$O/profiles; mkdir -p $O $P
(cd $R && timeout 1500 python -X faulthandler -m pytest tests -m gpu -q --timeout 300 2>&1 | grep -E "passed|failed|rror" | tail -5) > $O/pytest.txt; cat $O/pytest.txt
cd /tmp && export TMPDIR=/tmp
for S in 1 4; do
  N=$([ $S = 1 ] && echo prebuilt || echo cold)
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/k3trace_$S -o t -- python $R/scripts/k3_only.py 50 $S > $O/k3_only_$S.txt 2> $O/k3trace_$S.err
  (cd $R && python scripts/rocpd_stats.py $(find $O/k3trace_$S -name "*.db" | head -1) $P/${TAG}_k3_${N}_kernel_stats.csv > /dev/null)
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch_$S -o p -- python $R/scripts/k3_only.py 50 $S > /dev/null 2> $O/pmc_fetch_$S.err
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write_$S -o p -- python $R/scripts/k3_only.py 50 $S > /dev/null 2> $O/pmc_write_$S.err
  M=$([ $S = 1 ] && echo prebuilt || echo prebuilt_cold)
  (cd $R && python scripts/pmc_summary.py "k3_accumulate<false" $P/${TAG}_pmc_k3_${M}.json $(find $O/pmc_fetch_$S -name "*.db" | head -1) $(find $O/pmc_write_$S -name "*.db" | head -1) > /dev/null)
  cp $P/${TAG}_pmc_k3_${M}.json $R/profiles/   # (the bench line below quotes the counters collected on the kernel it times)
  rm -rf $O/k3trace_$S $O/pmc_fetch_$S $O/pmc_write_$S
done
cd $R
timeout 900 python bench.py > $P/${TAG}_bench_default.json 2> $O/bench_default.err
TLOAM_HIP_LIB=$R/tloam_amd/_variants/lib_stepprof.so timeout 100 python scripts/solve_profile2.py > $P/${TAG}_solve_all_timeline.txt 2>&1
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python $R/bench.py > $P/${TAG}_bench_default_under_rocprof.json 2> $O/trace.err
(cd $R && python scripts/rocpd_stats.py $(find $O/trace -name "*.db" | head -1) $P/${TAG}_bench_default_kernel_stats.csv > /dev/null); rm -rf $O/trace
timeout 300 rocprofv3 --kernel-trace -d $O/ktrace -o t -- python $R/bench.py --workload kitti --no-m1 --no-kitti --steps 150 --warmup 20 --no-cpu-baseline > /dev/null 2> $O/ktrace.err
(cd $R && python scripts/frame_timeline.py $(find $O/ktrace -name "*.db" | head -1) > $P/${TAG}_kitti_frame_timeline.txt 2>&1); rm -rf $O/ktrace
timeout 300 rocprofv3 --kernel-trace -d $O/mtrace -o t -- python $R/bench.py --workload m1 --steps 12 --warmup 2 --no-cpu-baseline --no-kitti --no-side > /dev/null 2> $O/mtrace.err
(cd $R && python scripts/frame_timeline.py $(find $O/mtrace -name "*.db" | head -1) > $P/${TAG}_m1_frame_timeline.txt 2>&1); rm -rf $O/mtrace
cd $R
# 6. the N > 1 line as the driver launches it, all ranks on the one GPU of this box (TLOAM_BENCH_ONE_DEVICE): the self-verifying
#    sharded frame (every form against the one-rank solve), NOT a scaling curve
for N in 2 4 8; do
  TLOAM_BENCH_ONE_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29520 + N)) bench.py --gpus $N --steps 20 --warmup 5 --m1-steps 4 > $P/${TAG}_bench_gpus${N}_one_device.json 2> $O/bench_gpus$N.err
done
# 7. the reduced chip (tests/test_gpu_cu_mask.py leaves its records in gpurun_out/)
cp $R/gpurun_out/cu_mask_*.json $P/ 2>/dev/null
for f in $P/cu_mask_*.json; do [ -f "$f" ] && mv "$f" "$P/${TAG}_$(basename $f)"; done
python scripts/profiles_readme.py $TAG $P > $P/README_${TAG}_section.md
cat $O/k3_only_1.txt $O/k3_only_4.txt; head -8 $P/${TAG}_k3_prebuilt_kernel_stats.csv | cut -c1-160; head -8 $P/${TAG}_k3_cold_kernel_stats.csv | cut -c1-160
tail -12 $P/${TAG}_kitti_frame_timeline.txt
python - <<PY
import json
d = json.loads(open("$P/${TAG}_bench_default.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value", d["value"], "ms_per_step", d["ms_per_step"], "roofline.frac (cold)", r["frac"], "l3_resident", r.get("l3_resident", {}).get("frac"),
      "m1", d["m1_frame"]["ms_per_frame"], "per_call", {k: d["config"].get(k) for k in ("set_source_ms", "set_target_ms", "scan_match_ms", "ms_per_frame_incl_pcie_upload")},
      "odometry", d.get("odometry_loop", {}).get("ms_per_frame"), "submap", d.get("adjacent_rows", {}).get("submap_update_ms"))
PY
