#!/bin/bash
# usage: tools/profile_sec.sh NAME  (GPU box): rocprofv3 kernel stats of the ViT-L/14 16-frame training block (4 steps)
NAME=$1
export TMPDIR=/tmp
D=gpurun_out/$NAME.prof
rm -rf $D
rocprofv3 --kernel-trace --stats -d $D -o run -- python3 bench.py --secondary-only > gpurun_out/$NAME.log 2>&1
DB=$(find $D -name 'run_results.db' | head -1)
python3 tools/prof_summary.py "$DB" 4 "rocprofv3 --kernel-trace --stats: python3 bench.py --secondary-only (ViT-L/14, 16 frames, 32 clips; 1 warm-up + 3 timed steps)" > gpurun_out/$NAME.md
rm -rf $D
