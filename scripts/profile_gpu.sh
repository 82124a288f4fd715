#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes of the bench command.
#   scripts/profile_gpu.sh TAG [extra bench.py arguments, e.g. --robot human --bs 1024]
# Outputs under gpurun_out/prof_$TAG/ ; scripts/summarize_profile.py turns them into profiles/*.
TAG=${1:-r02}
shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 10 --warmup 2 --repeats 2 --no-cpu-baseline --no-boundary $@"
echo "$BENCH" > $OUT/command.txt
# which library these counters are of: the source half of pd_build_id() -- bench.py drops counter-derived fields when it differs from the library it runs
python3 -c "import sys; sys.path.insert(0, '$REPO/ppr-diffphys_amd'); from diffphys_amd import hip_backend; print(hip_backend.build_id().split('+')[-1])" > $OUT/source_hash.txt 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- $BENCH > $OUT/pmc_sq2.log 2>&1
find $OUT -name "*.csv" | head -50 > $OUT/files.txt
# keep the merge small: drop anything that is not a csv/log/txt
find $OUT -type f ! -name "*.csv" ! -name "*.log" ! -name "*.txt" -delete
du -sh $OUT
