#!/bin/bash
# usage: tools/pmc_run.sh <outdir-under-gpurun_out> [bench args...]
# SQ counter passes for the fused kernel (one rocprofv3 --pmc run each, wrapped in timeout).
out=$PWD/gpurun_out/$1; shift
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS" \
           "SQ_IFETCH SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU" \
           "SQC_ICACHE_MISSES SQC_ICACHE_HITS SQC_ICACHE_REQ SQ_LEVEL_WAVES SQ_CYCLES SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  timeout 180 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -- python3 $OLDPWD/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras "$@" > $out/p$i.log 2>&1
  f=$(find $out/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $OLDPWD/tools/pmc_summary.py "oct_fused_kernel" $f
done
