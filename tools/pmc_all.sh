#!/bin/bash
# usage: tools/pmc_all.sh NAME   (GPU box, repo root): FETCH_SIZE and WRITE_SIZE passes of one bench step -> gpurun_out/NAME.md
NAME=$1
export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-secondary --no-inference"
for pass in "f:FETCH_SIZE" "w:WRITE_SIZE"; do
  tag=${pass%%:*}; ctr=${pass#*:}
  rm -rf gpurun_out/$NAME.$tag
  rocprofv3 --kernel-trace --pmc $ctr -d gpurun_out/$NAME.$tag -o run -- python3 bench.py $ARGS > gpurun_out/$NAME.$tag.log 2>&1 || echo "pass $tag failed"
done
python3 tools/pmc_all.py $(find gpurun_out/$NAME.f -name run_results.db | head -1) $(find gpurun_out/$NAME.w -name run_results.db | head -1) 2 > gpurun_out/$NAME.md
rm -rf gpurun_out/$NAME.f gpurun_out/$NAME.w
head -40 gpurun_out/$NAME.md
