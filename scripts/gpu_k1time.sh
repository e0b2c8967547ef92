#!/bin/bash
# K1 launch time, default lib vs variants.  usage: gpu_k1time.sh "<variant names>"
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for L in default $1; do
if [ $L = default ]; then unset TLOAM_HIP_LIB; else export TLOAM_HIP_LIB=$R/tloam_amd/_variants/lib_$L.so; fi
timeout 200 python scripts/k1_time.py 20 2>&1 | grep "^lib"
done
