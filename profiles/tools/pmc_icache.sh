#!/bin/bash
# usage: pmc_icache.sh <tag> [bench args] -- instruction-cache and wait-class counters of the q_phase kernels
tag=$1; shift
R=$PWD
OUT=$R/gpurun_out/pmci_$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAVES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INST_CYCLES_SALU SQ_INSTS_SENDMSG"; do
  i=$((i+1))
  rm -rf /tmp/ri_${tag}_$i
  rocprofv3 --kernel-trace --pmc $set -d /tmp/ri_${tag}_$i --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras "$@" > /tmp/ri_${tag}_$i.log 2>&1
  python3 $R/profiles/tools/pmc_summary.py /tmp/ri_${tag}_$i q_phase_kernel > $OUT/pass$i.json
done
python3 - <<PY
import json
d={}
for i in (1,2,3):
    for k,v in json.load(open("$OUT/pass%d.json"%i)).items():
        d.setdefault(k,{}).update(v)
for k,v in d.items():
    print(k[:70])
    for c in sorted(v):
        if not c.startswith("_"): print("   %-28s %.4g"%(c,v[c]))
PY
