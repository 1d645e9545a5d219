#!/bin/bash
# ISA lint of the hot kernels (no GPU needed: cross-compiles the device code, ~1 min).  From the repo root:
#   bash tools/isa/lint.sh
# Prints, per kernel, prefetches the scheduler has sunk to their first use (waits on a load issued <= 16 instructions
# earlier, near MFMAs) and the 200-instruction windows with >= 20 branches (wave-uniform tests left in per-value code).
set -e
tmp=$(mktemp -d)
dis() {  # source file -> listing
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Iinclude --cuda-device-only -c $1 -o $tmp/x.co
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$tmp/x.co --output=$tmp/x.elf --unbundle
  /opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn $tmp/x.elf > $2
}
dis graspldm_amd/csrc/resnet1d.hip $tmp/r1d.s
dis graspldm_amd/csrc/voxel_conv.hip $tmp/vc.s
for k in r1d_kernelILi64ELi4 r1d_kernelILi32ELi16 pointwise_mlp_sp_kernel; do python3 tools/isa/sunk_prefetch_scan.py $tmp/r1d.s $k; python3 tools/isa/branch_density.py $tmp/r1d.s $k; done
# hand-written DPP blocks (quad_narrow.h, resnet1d.hip): no VALU write closer than 2 wait states in front of a DPP read
# of the same register -- nothing checks that inside an asm statement; fails the lint on any hit
rc=0
for k in r1d_kernelILi64ELi4 r1d_kernelILi64ELi16 r1d_kernelILi32ELi4 r1d_kernelILi32ELi16 sa_mlp3_kernel sa_mlp2_kernel pointwise_mlp_sp_kernel; do
  python3 tools/isa/dpp_hazard_scan.py $tmp/r1d.s $k || rc=1
done
python3 tools/isa/dpp_hazard_scan.py $tmp/vc.s conv3d_k3 || rc=1
for k in conv3d_k3_pl_kernelILi3ELi24ELi24ELi8ELb1 conv3d_k3_pl_kernelILi6ELi12ELi12ELi4ELb0 conv3d_k3_kernelILi3ELi6ELi4; do python3 tools/isa/sunk_prefetch_scan.py $tmp/vc.s $k; python3 tools/isa/branch_density.py $tmp/vc.s $k; done
rm -rf $tmp
exit $rc
