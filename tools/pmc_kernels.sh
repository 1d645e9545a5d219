#!/bin/bash
# Per-kernel rocprofv3 evidence for any script.  Run on the GPU box from the repo root (through gpurun):
#   bash tools/pmc_kernels.sh <tag> <script.py> [args...]   -> gpurun_out/pmck_<tag>/{stats,fetch,write,mfma}/ + pmck_<tag>.json
# One run per counter group (--kernel-trace --pmc only) and one --kernel-trace --stats run; summarised per kernel name by
# tools/pmc_kernels_summary.py.  The program after `--` is the interpreter itself (no env / bash -c hop).
tag=$1; shift
script=$GRAFT_REPO_ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmck_$tag
mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o run -- /usr/bin/python3 $script "$@" > $out/stats.log 2>&1
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $out/$name -o run -- /usr/bin/python3 $script $ARGS > $out/$name.log 2>&1; }
ARGS="$@"
run fetch FETCH_SIZE
run write WRITE_SIZE
run mfma GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU
if [ -n "$PMCK_MORE" ]; then   # optional groups: where the waves wait, LDS conflicts, L2 hit rate
  run wait GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
  run lds GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS
  run l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
fi
cd $GRAFT_REPO_ROOT && python3 tools/pmc_kernels_summary.py gpurun_out/pmck_$tag gpurun_out/pmck_$tag.json | tee gpurun_out/pmck_$tag.txt
