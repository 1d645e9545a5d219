#!/bin/bash
# A/B of two builds of libgldm_hip.so on ONE box (box-to-box spread is ~1-2 %): the fused denoise launch, alternating.
#   gpurun -- bash tools/ab_denoise.sh graspldm_amd/csrc/build/lib_A.so graspldm_amd/libgldm_hip.so [rounds]
A=$1; B=$2; R=${3:-3}
for i in $(seq $R); do
  for L in $A $B; do
    printf "%s: " $L; GLDM_LIB=$L python tools/run_denoise_once.py 5120 100 2>/dev/null | tail -1
  done
done
