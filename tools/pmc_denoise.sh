#!/bin/bash
# PMC passes on one fused denoise launch (4096 latents x 100 steps).  Run on the GPU box from the repo root:
#   bash tools/pmc_denoise.sh <tag>      -> gpurun_out/pmc_<tag>/<group>/...
# One counter group per run (--kernel-trace --pmc only), summarised by tools/pmc_summary.py.
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
run() { name=$1; shift; timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $out/$name -o run -- /usr/bin/python3 $GRAFT_REPO_ROOT/tools/run_denoise_once.py ${NLAT:-5120} 100 > $out/$name.log 2>&1; }
run busy GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH
run l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
run inst SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA
cd $GRAFT_REPO_ROOT && python3 tools/pmc_summary.py gpurun_out/pmc_$tag gpurun_out/pmc_$tag.json | tee gpurun_out/pmc_$tag.txt
