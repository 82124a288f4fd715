#!/bin/bash
# Emits gfx950 assembly of the SEGW=16 kernels to /tmp/isa/k16.s and splits the revolute rollout kernels out (diagnostic).
mkdir -p /tmp/isa
cd /root/repo/ppr-diffphys_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=fast -fno-hip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp -DPD_SEGW=16 -S --cuda-device-only pd_kernels.hip -o /tmp/isa/k16.s 2>/dev/null
cd /tmp/isa
awk '/^_Z13k_rollout_fwdILi16ELi1ELb1E/{f=1} f{print} /s_endpgm/{if(f){exit}}' k16.s > fwd.s
awk '/^_Z13k_rollout_bwdILi16ELi1ELb1E/{f=1} f{print} /s_endpgm/{if(f){exit}}' k16.s > bwd.s
grep -A12 "^    .name:           _Z13k_rollout_...ILi16ELi1ELb1E" k16.s | grep "name\|vgpr_count\|sgpr_count\|spill"
