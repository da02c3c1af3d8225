#!/bin/bash
# usage (on the GPU box, from the repo root): bash profiles/tools/collect_pmc.sh <tag> [bench.py args...]
# Writes gpurun_out/prof_<tag>/: kernel_stats.csv (rocprofv3 --kernel-trace --stats of `python3 bench.py <args>`),
# one JSON per PMC pass (counter passes are separate runs of ONE step each: the TCC counters do not fit one pass), bench.json.
tag=$1; shift
R=$PWD
OUT=$R/gpurun_out/prof_$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_${tag}_stats
rocprofv3 --kernel-trace --stats -d /tmp/rp_${tag}_stats --output-format csv -- python3 $R/bench.py --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/bench.err
f=$(find /tmp/rp_${tag}_stats -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/kernel_stats.csv
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
           "SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_WAVE_CYCLES SQ_WAVES" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/rp_${tag}_$i
  rocprofv3 --kernel-trace --pmc $set -d /tmp/rp_${tag}_$i --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extras "$@" --steps 1 --warmup 0 > /tmp/rp_${tag}_$i.log 2>&1
  python3 $R/profiles/tools/pmc_summary.py /tmp/rp_${tag}_$i q_phase > $OUT/pmc_pass$i.json
done
tail -1 $OUT/bench.json | cut -c1-200
head -5 $OUT/kernel_stats.csv
cat $OUT/pmc_pass3.json $OUT/pmc_pass4.json | grep -E "SIZE|Scratch"
