#!/bin/bash
# usage: gpurun_pmc.sh <tag> <bench args...>   (run on the GPU box from the repo root)
tag=$1; shift
R=$PWD
mkdir -p $R/gpurun_out/pmc_$tag
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_$tag_$n --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > /tmp/pmc_$tag_$n.log 2>&1
  python3 $R/profiles/tools/pmc_summary.py /tmp/pmc_$tag_$n q_phase > $R/gpurun_out/pmc_$tag/$n.json
done
cat $R/gpurun_out/pmc_$tag/*.json
