#!/bin/bash
# usage (GPU box, repo root): bash profiles/tools/soak_round.sh <tag>  -> gpurun_out/<tag>/final_soak_<tag>.txt, fuzz_<tag>.txt
# Repeatability of the default bench on the final build (six consecutive runs + one under torch.distributed.run), smoke(), and the
# poisoned fuzz campaigns (every launch preceded by a kernel that fills scratch and all vector registers with quiet NaN).
tag=$1
OUT=gpurun_out/$tag; mkdir -p $OUT
out=$OUT/final_soak_$tag.txt; : > $out
echo "# library build $(cut -c1-16 stac_mjx_amd/csrc/libstac_hip.so.stamp)" >> $out
echo "== python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 bench.py --gpus 1 --steps 20 --warmup 5 (RCCL init, barrier, max over ranks)" >> $out
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('value', round(d['value']), 'ms_per_step', round(d['ms_per_step'],3), 'collective_backend', d['config']['collective_backend'], 'per_rank', d['config']['per_rank'], 'predicted_value', round(d['config']['predicted_value']) if d['config'].get('predicted_value') else None)" >> $out
echo "== consecutive default runs: python bench.py --gpus 1 --steps 20 --warmup 5" >> $out
for i in 1 2 3 4 5 6; do python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('run $i value', round(d['value']), 'ms_per_step', round(d['ms_per_step'],3), 'clips250', round(d['config']['clips250']['frames_per_s']), 'us/iter', round(d['config']['clips250']['us_per_pg_iteration'],3), 'lm', round(d['config']['lm_solver_same_batch']['frames_per_s']), 'traffic', d['roofline']['traffic'], 'valu_busy', round(d['roofline_valu']['valu_issue_busy_frac'],3) if 'roofline_valu' in d and 'valu_issue_busy_frac' in d['roofline_valu'] else None, 'cpu', round(d['cpu_baseline']['value']))" >> $out; done
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/${tag}_bench_plain.json
echo "== __graft_entry__.smoke()" >> $out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke >> $out
export STAC_FUZZ_POISON=7fc00000
f=$OUT/fuzz_$tag.txt; : > $f
echo "# library build $(cut -c1-16 stac_mjx_amd/csrc/libstac_hip.so.stamp); STAC_FUZZ_POISON=7fc00000 (scratch + 512 VGPRs of every SIMD filled with quiet NaN before every launch)" >> $f
for spec in "700" "300 big" "500 lean" "120 leanwide" "600 leanbq" "120 leanwidebq"; do
  echo "== python tests/fuzz_random_models.py $spec" >> $f
  ( time timeout 1500 python tests/fuzz_random_models.py $spec ) 2>&1 | grep -v amdgpu.ids | tail -5 >> $f
done
for envs in "STAC_HIP_SPEC=1 STAC_HIP_SPECG=16" "STAC_HIP_SPEC=1 STAC_HIP_SPECG=32" "STAC_HIP_SPEC=1 STAC_HIP_SPECG=16 STAC_HIP_SPECR=8" "STAC_HIP_SPEC=0" "STAC_HIP_NOLEAN=1"; do
  echo "== $envs python tests/fuzz_random_models.py 150 leanbq" >> $f
  ( time env $envs timeout 900 python tests/fuzz_random_models.py 150 leanbq ) 2>&1 | grep -v amdgpu.ids | tail -5 >> $f
done
unset STAC_FUZZ_POISON
echo "== python tests/fuzz_lm_random_models.py 120" >> $f
( time timeout 900 python tests/fuzz_lm_random_models.py 120 ) 2>&1 | grep -v amdgpu.ids | tail -4 >> $f
cat $out
