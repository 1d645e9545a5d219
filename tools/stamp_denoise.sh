#!/bin/bash
# Per-op cycle stamps of one denoise step (diagnostic build).  Run from the repo root IN THE BUILD CONTAINER:
#   bash tools/stamp_denoise.sh [tag]      -> gpurun_out/stamp_<tag>.txt
# builds graspldm_amd/csrc/build/libgldm_hip_dbg.so (-DGLDM_DEBUG_KNOBS) here, runs it on the GPU box.
set -e
tag=${1:-cur}
rm -f graspldm_amd/csrc/build/dbg/resnet1d.o
make -C graspldm_amd/csrc EXTRA="-DGLDM_DEBUG_KNOBS $EXTRA_DEFS" BUILD=$PWD/graspldm_amd/csrc/build/dbg OUT=$PWD/graspldm_amd/csrc/build/libgldm_hip_dbg.so -j4 2>&1 | grep -i "error\|warning: v" || true
gpurun --timeout 600 -- "GLDM_LIB=graspldm_amd/csrc/build/libgldm_hip_dbg.so GLDM_R1D_STAMP=1 python tools/run_denoise_once.py 4096 20 > gpurun_out/stamp_$tag.txt 2>&1; GLDM_LIB=graspldm_amd/csrc/build/libgldm_hip_dbg.so python tools/run_denoise_once.py 5120 100 >> gpurun_out/stamp_$tag.txt 2>&1" > gpurun_out/stamp_call.log 2>&1
grep -v "^\[gpurun\] sending" gpurun_out/stamp_call.log | head -3
awk '/^op/{a[$0]=1} END{}' gpurun_out/stamp_$tag.txt
tail -45 gpurun_out/stamp_$tag.txt
