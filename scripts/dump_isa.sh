#!/bin/bash
# Emits gfx950 assembly of the SEGW=$1 (default 16) kernels to /tmp/isa/k$1.s and prints register / spill metadata of
# every rollout kernel (diagnostic).
W=${1:-16}
mkdir -p /tmp/isa
cd /root/repo/ppr-diffphys_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=fast -fno-hip-fp32-correctly-rounded-divide-sqrt -fno-signed-zeros -fno-slp-vectorize -falign-loops=32 -mllvm -amdgpu-sched-strategy=iterative-ilp -DPD_SEGW=$W $EXTRA -S --cuda-device-only pd_kernels.hip -o /tmp/isa/k$W.s 2>/dev/null
python3 - $W <<'PY'
import re,sys
w=sys.argv[1]
s=open('/tmp/isa/k%s.s'%w).read()
for m in re.finditer(r'\.name:\s+(_Z\S+)\n(.*?)\.wavefront_size', s, re.S):
    name=m.group(1)
    if 'rollout' not in name: continue
    body=m.group(2)
    g=lambda k: (re.search(r'\.%s:\s+(\d+)'%k, body) or [0,'?'])[1]
    print("%-62s vgpr %s agpr %s spill %s sgpr %s sspill %s" % (name[:62], g('vgpr_count'), g('agpr_count'), g('vgpr_spill_count'), g('sgpr_count'), g('sgpr_spill_count')))
PY
