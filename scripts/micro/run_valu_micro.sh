#!/bin/bash
# The VALU issue ceiling behind bench.py's roofline.valu.peak, as a FILE (VERDICT r5 missing #5).
#   build container:  bash scripts/micro/run_valu_micro.sh build      (hipcc cross-compiles; binaries under scripts/micro/bin travel with gpurun)
#   GPU box:          bash scripts/micro/run_valu_micro.sh run > gpurun_out/r06_micro_valu.txt   -> copy to profiles/
# Built with the LIBRARY's code-generation flags (csrc/Makefile CXXFLAGS; -fno-slp-vectorize is the one that matters: plain -O3 packs two
# chains into v_pk_fma_f32 and halves the figure per FMA -- the round-1 logs' 2.4-2.6 "cycles per FMA" were that artefact).
set -e
cd "$(dirname "$0")"
FLAGS="-w -O3 -std=c++17 --offload-arch=gfx950 -fno-fast-math -ffp-contract=fast -fno-hip-fp32-correctly-rounded-divide-sqrt -fno-signed-zeros -fno-slp-vectorize -falign-loops=32 -mllvm -amdgpu-sched-strategy=iterative-ilp"
case "$1" in
build)
  mkdir -p bin
  for f in valu_chain pk_issue issue_rate issue_mix; do /opt/rocm/bin/hipcc $FLAGS $f.hip -o bin/$f; done
  # what the inner loops compiled to: the count of v_fma_f32 / v_pk_fma_f32 in each binary's device code, so the run's reader can see that
  # "cycles per FMA" is per v_fma_f32 (non-packed) in valu_chain and per named instruction in pk_issue
  for f in valu_chain pk_issue; do
    /opt/rocm/bin/hipcc $FLAGS --cuda-device-only -S $f.hip -o bin/$f.s
    echo "$f: v_fma_f32 $(grep -c 'v_fma_f32\|v_fmac_f32' bin/$f.s)  v_pk_fma_f32 $(grep -c v_pk_fma_f32 bin/$f.s)  v_pk_mul_f32 $(grep -c v_pk_mul_f32 bin/$f.s)  v_pk_add_f32 $(grep -c v_pk_add_f32 bin/$f.s)" | tee bin/$f.mix
  done
  ;;
run)
  echo "# scripts/micro/run_valu_micro.sh run -- $(date -u +%Y-%m-%dT%H:%MZ) on $(/opt/rocm/bin/rocminfo 2>/dev/null | grep -m1 'Marketing Name.*MI' | sed 's/.*: *//')"
  echo "# flags: $FLAGS"
  echo "# static instruction mix of the two binaries' device code (whole file, all template instantiations):"
  cat bin/valu_chain.mix bin/pk_issue.mix | sed 's/^/#   /'
  echo "# NOTE the guide (/opt/skills/guides/MI355X_MICROARCH.md, instruction table) lists v_fma_f32 wave64 at 2 cyc (SIMD-32) throughput and 4 for one wave alone;"
  echo "#      what follows is what THIS box measures, wall time x an ASSUMED 2.4 GHz (valu_chain) and s_memtime ticks of 100 MHz x24 (pk_issue prints both)."
  for rep in 1 2; do
    echo "== valu_chain (non-packed v_fma_f32, 1/2/4/8 independent chains per lane; 1, 2, 4 waves per SIMD), repeat $rep"
    ./bin/valu_chain
  done
  for rep in 1 2; do
    echo "== pk_issue (v_fma_f32 vs v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, 8 chains; 1, 2, 3, 4 waves per SIMD), repeat $rep"
    ./bin/pk_issue
  done
  for rep in 1 2; do
    echo "== issue_rate (round 6: the same 64-instruction loop by OPERAND KIND and encoding -- all-VGPR VOP2 / VOP3, an SGPR operand, an inline constant, a literal; 8 chains; 1 .. 4 waves per SIMD), repeat $rep"
    ./bin/issue_rate
  done
  echo "== issue_mix (round 6: what a non-vector instruction costs a wave whose stream is a chain of v_fma_f32: 64 FMAs + 32 extras per iteration; 1 and 2 waves per SIMD)"
  ./bin/issue_mix
  ;;
*) echo "usage: $0 build|run"; exit 2;;
esac
