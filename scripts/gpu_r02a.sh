#!/bin/bash
# round 2, exploration call A: baseline tests, K3 grid/occupancy sweep, K3 timeline, step-kernel phase profile
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r02a; mkdir -p $O; cd $R
V=$R/tloam_amd/_variants
(timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3) > $O/pytest.txt; cat $O/pytest.txt
{
timeout 300 python scripts/k3_sweep.py 60 auto,512,256,384,640,768,977,1024
TLOAM_HIP_LIB=$V/lib_w2d21.so timeout 200 python scripts/k3_sweep.py 60 auto,512,768,977
TLOAM_HIP_LIB=$V/lib_w3d11.so timeout 200 python scripts/k3_sweep.py 60 auto,512,768,1024,1536
TLOAM_HIP_LIB=$V/lib_w4d11.so timeout 300 python scripts/k3_sweep.py 60 auto,512,768,1024,1536,1954,2048
TLOAM_HIP_LIB=$V/lib_w4d11nt.so timeout 200 python scripts/k3_sweep.py 60 auto,1024
} 2>&1 | grep -v "^$" | tee $O/k3_sweep.txt
TLOAM_HIP_LIB=$V/lib_k3prof.so timeout 200 python scripts/k3_profile.py 2>&1 | tee $O/k3_timeline.txt
TLOAM_HIP_LIB=$V/lib_stepprof.so timeout 200 python scripts/step_profile.py 2>&1 | tee $O/step_profile.txt
