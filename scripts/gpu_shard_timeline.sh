#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/tl; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
N=${1:-8}; FORM=${2:-mailbox}
timeout 300 rocprofv3 --kernel-trace -d $O/strace -o t -- python $R/scripts/shard_loopback.py $N $FORM 12 > $O/shard.txt 2> $O/strace.err
cat $O/shard.txt
(cd $R && python scripts/frame_timeline.py $(find $O/strace -name "*.db" | head -1) > $O/shard_timeline_n${N}_$FORM.txt 2>&1); rm -rf $O/strace
cat $O/shard_timeline_n${N}_$FORM.txt
