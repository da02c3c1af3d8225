#!/bin/bash
# usage: pmc_quick.sh <tag> [bench args]
tag=$1; shift
R=$PWD
OUT=$R/gpurun_out/pmcq_$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rm -rf /tmp/rq_${tag}_$i
  rocprofv3 --kernel-trace --pmc $set -d /tmp/rq_${tag}_$i --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras "$@" > /tmp/rq_${tag}_$i.log 2>&1
  python3 $R/profiles/tools/pmc_summary.py /tmp/rq_${tag}_$i q_phase_kernel > $OUT/pass$i.json
done
python3 - <<PY
import json
a=json.load(open("$OUT/pass1.json")); b=json.load(open("$OUT/pass2.json"))
for k in a:
    d=dict(a[k]); d.update(b.get(k,{}))
    w=d.get("SQ_WAVES",1)
    print(k[:60])
    for c in ("SQ_WAVES","SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_INSTS_LDS","SQ_WAVE_CYCLES","SQ_ACTIVE_INST_VALU","SQ_WAIT_INST_ANY","SQ_WAIT_ANY","SQ_ACTIVE_INST_LDS","SQ_ACTIVE_INST_SCA","SQ_LDS_IDX_ACTIVE","SQ_LDS_BANK_CONFLICT","SQ_WAIT_INST_LDS","_VGPR_Count","_Scratch_Size"):
        if c in d: print("   %-22s %.4g   per wave %.4g"%(c,d[c],d[c]/w))
PY
