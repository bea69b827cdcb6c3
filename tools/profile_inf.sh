#!/bin/bash
# usage: tools/profile_inf.sh NAME  (GPU box): rocprofv3 kernel stats of the fp8 / bf16 multi-view inference block
NAME=$1
export TMPDIR=/tmp
D=gpurun_out/$NAME.prof
rm -rf $D
rocprofv3 --kernel-trace --stats -d $D -o run -- python3 bench.py --inference-only > gpurun_out/$NAME.log 2>&1
DB=$(find $D -name 'run_results.db' | head -1)
python3 tools/prof_summary.py "$DB" 12 "rocprofv3 --kernel-trace --stats: python3 bench.py --inference-only (6 bf16 + 6 fp8 steps of 12 views; per 'step' = 1/12 of the trace)" > gpurun_out/$NAME.md
rm -rf $D
