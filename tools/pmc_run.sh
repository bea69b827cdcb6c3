#!/bin/bash
# usage: tools/pmc_run.sh NAME   (GPU box, repo root): three PMC passes of one bench step -> gpurun_out/NAME.json
NAME=$1
export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-secondary --no-inference"
for pass in "f:FETCH_SIZE" "w:WRITE_SIZE" "s:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE SQ_WAVE_CYCLES"; do
  tag=${pass%%:*}; ctr=${pass#*:}
  rm -rf gpurun_out/$NAME.$tag
  rocprofv3 --kernel-trace --pmc $ctr -d gpurun_out/$NAME.$tag -o run -- python3 bench.py $ARGS > gpurun_out/$NAME.$tag.log 2>&1 || echo "pass $tag failed"
done
python3 tools/pmc_summary.py $(find gpurun_out/$NAME.f -name run_results.db | head -1) $(find gpurun_out/$NAME.w -name run_results.db | head -1) $(find gpurun_out/$NAME.s -name run_results.db | head -1) > gpurun_out/$NAME.json
rm -rf gpurun_out/$NAME.f gpurun_out/$NAME.w gpurun_out/$NAME.s
head -c 600 gpurun_out/$NAME.json
