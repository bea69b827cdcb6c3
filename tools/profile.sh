#!/bin/bash
# usage: tools/profile.sh NAME [bench args...]   (on the GPU box, from the repo root)
# rocprofv3 --kernel-trace --stats of bench.py's primary workload -> gpurun_out/NAME.md (per-kernel table).
NAME=$1; shift
export TMPDIR=/tmp
D=gpurun_out/$NAME.prof
rm -rf $D
rocprofv3 --kernel-trace --stats -d $D -o run -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --no-inference "$@" > gpurun_out/$NAME.log 2>&1
DB=$(find $D -name 'run_results.db' | head -1)
python3 tools/prof_summary.py "$DB" 6 "rocprofv3 --kernel-trace --stats: python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --no-inference $*" main > gpurun_out/$NAME.md
grep -o '"value": [0-9.]*, "unit": "clips/s"' gpurun_out/$NAME.log | head -1
rm -rf $D
