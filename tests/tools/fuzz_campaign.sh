#!/bin/bash
# usage (GPU box, repo root): bash tests/tools/fuzz_campaign.sh [N small] [N big]  -> gpurun_out/fuzz_campaign.txt
# tests/fuzz_random_models.py (HIP == oracle at tolerance 0 on random trees, every launch repeated in the big flavour) under the
# developer switches that aim at the individual kernel shapes, with the scratch + register poisoner on.
NS=${1:-200}; NB=${2:-100}
out=gpurun_out/fuzz_campaign.txt; mkdir -p gpurun_out; : > $out
echo "# library build $(cat stac_mjx_amd/csrc/libstac_hip.so.stamp | cut -c1-16); poison 0x7FC00000 before every q_phase launch" >> $out
export STAC_TEST_POISON=off STAC_FUZZ_POISON=7FC00000
run() { n=$1; flav=$2; shift 2; echo "== $flav $*" >> $out; env "$@" timeout 1500 python tests/fuzz_random_models.py $n $flav 2>&1 | grep -v amdgpu.ids | tail -3 >> $out; }
run $((3*NS)) "" X=1
for sw in STAC_HIP_SPEC=0 STAC_HIP_FLAGS=4 STAC_HIP_FLAGS=1 STAC_HIP_FLAGS=2 STAC_HIP_NOFAST=1 STAC_HIP_NOFREE0=1 STAC_HIP_NOPRUNE=1 STAC_HIP_NODIET=1 \
          STAC_HIP_WPE=3 STAC_HIP_WPE=2 "STAC_HIP_QUEUE=2 STAC_HIP_SPEC=0" "STAC_HIP_HANDOFF=2 STAC_HIP_SPEC=0" \
          "STAC_HIP_SPEC=1 STAC_HIP_SPECG=8" "STAC_HIP_SPEC=1 STAC_HIP_SPECG=16" "STAC_HIP_SPEC=1 STAC_HIP_SPECG=32"; do
  run $NS "" $sw
done
run $((3*NB)) big X=1
for sw in STAC_HIP_SPEC=0 "STAC_HIP_SPEC=1 STAC_HIP_SPECG=32" "STAC_HIP_SPEC=1 STAC_HIP_SPECG=64" STAC_HIP_WPE=4 STAC_HIP_WPE=3 "STAC_HIP_QUEUE=8 STAC_HIP_SPEC=0" \
          "STAC_HIP_HANDOFF=8 STAC_HIP_SPEC=0" STAC_HIP_FLAGS=2 STAC_HIP_NOFAST=1; do
  run $NB big $sw
done
run $((2*NS)) lean X=1
for sw in STAC_HIP_SPEC=0 STAC_HIP_NOFAST=1 STAC_HIP_NOPRUNE=1 "STAC_HIP_QUEUE=8 STAC_HIP_SPEC=0" "STAC_HIP_HANDOFF=8 STAC_HIP_SPEC=0" \
          "STAC_HIP_SPEC=1 STAC_HIP_SPECG=16" "STAC_HIP_SPEC=1 STAC_HIP_SPECG=16 STAC_HIP_SPECR=8" "STAC_HIP_SPEC=1 STAC_HIP_SPECG=32"; do
  run $NS lean $sw
done
run $NS leanwide X=1
for sw in STAC_HIP_SPEC=0 "STAC_HIP_SPEC=1 STAC_HIP_SPECG=32" STAC_HIP_NOFAST=1; do
  run $NB leanwide $sw
done
# round 6: trees of the lean kind WITH oriented bodies (the fruit fly's kind: one more product of the quaternion pass each), and the
# switch that moves ranges between the by-component and the whole-range sums
run $((2*NS)) leanbq X=1
for sw in STAC_HIP_SPEC=0 STAC_HIP_NOFAST=1 STAC_HIP_NOPRUNE=1 STAC_HIP_RSPLIT=0 STAC_HIP_RSPLIT=8 "STAC_HIP_QUEUE=8 STAC_HIP_SPEC=0" \
          "STAC_HIP_HANDOFF=8 STAC_HIP_SPEC=0" "STAC_HIP_SPEC=1 STAC_HIP_SPECG=16" "STAC_HIP_SPEC=1 STAC_HIP_SPECG=16 STAC_HIP_SPECR=8" \
          "STAC_HIP_SPEC=1 STAC_HIP_SPECG=32" STAC_HIP_NOFK3BQ=1; do
  run $NS leanbq $sw
done
run $NB leanwidebq X=1
for sw in STAC_HIP_RSPLIT=0 STAC_HIP_RSPLIT=8; do
  run $NS lean $sw
done
echo "== LM" >> $out
timeout 900 python tests/fuzz_lm_random_models.py $NB 2>&1 | grep -v amdgpu.ids | tail -2 >> $out
cat $out
