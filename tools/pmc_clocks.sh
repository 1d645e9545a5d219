#!/bin/bash
# Effective shader clock and MFMA-pipe occupancy per kernel of one bench pass (one PMC run, kernel trace only):
#   bash tools/pmc_clocks.sh <tag>   -> gpurun_out/clk_<tag>.txt
tag=${1:-x}
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/clk_$tag -o run -- /usr/bin/python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --streams 1 > $R/gpurun_out/clk_$tag.log 2>&1
cd $R
python3 tools/pmc_clocks.py gpurun_out/clk_$tag | tee gpurun_out/clk_$tag.txt
