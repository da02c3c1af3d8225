#!/bin/bash
# usage (on the GPU box, from the repo root): bash profiles/tools/collect_round.sh <round tag, e.g. r03>
# Everything a round's numbers come from, in one go; results under gpurun_out/<tag>/ (copy what is to be judged into profiles/<tag>/):
#   <tag>a_*      the default bench (BASELINE configs[1]): bench line, rocprofv3 kernel stats, five PMC passes
#   <tag>clips_*  the same with the reference's default chaining (250 frames per clip)
#   secondary.json   raw bench lines of the other configurations quoted in README / DESIGN
#   lat_sweep.txt    latency-mode shapes against the automatic choice
tag=$1
R=$PWD
OUT=$R/gpurun_out/$tag
mkdir -p $OUT
cat stac_mjx_amd/csrc/libstac_hip.so.stamp > $OUT/lib_digest.txt
bash profiles/tools/collect_pmc.sh ${tag}a --steps 20 --warmup 5 > $OUT/collect_a.log 2>&1
bash profiles/tools/collect_pmc.sh ${tag}clips --steps 2 --warmup 1 --frames-per-clip 250 --no-extras > $OUT/collect_clips.log 2>&1
for t in a clips; do
  mkdir -p $OUT/${tag}$t; cp gpurun_out/prof_${tag}$t/* $OUT/${tag}$t/ 2>/dev/null
done
{
  echo "["
  first=1
  run() {  # label, args...
    label=$1; shift
    line=$(python3 bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | tail -1)
    [ -z "$line" ] && line='null'
    [ $first = 1 ] || echo ","
    first=0
    echo "{\"label\": \"$label\", \"args\": \"$*\", \"line\": $line}"
  }
  run "100k frames, n_frames_per_clip=1" --frames 100000 --steps 3 --warmup 1
  run "125k frames, n_frames_per_clip=250 (config 4 share)" --frames 125000 --frames-per-clip 250 --steps 1 --warmup 1
  run "250k frames, n_frames_per_clip=250" --frames 250000 --frames-per-clip 250 --steps 1 --warmup 1
  run "375k frames, n_frames_per_clip=250 (1500 chains: the N = 3 share of configs[3])" --frames 375000 --frames-per-clip 250 --steps 1 --warmup 1
  run "500k frames, n_frames_per_clip=250 (2000 chains: the N = 2 share of configs[3])" --frames 500000 --frames-per-clip 250 --steps 1 --warmup 1
  run "750k frames, n_frames_per_clip=250 (3000 chains)" --frames 750000 --frames-per-clip 250 --steps 1 --warmup 1
  run "fly 10k frames" --model fly --steps 5 --warmup 2
  run "fly 25k frames at 250 per clip (config 5: the per-GPU share of 200k frames on 8 GPUs)" --model fly --frames 25000 --frames-per-clip 250 --steps 2 --warmup 1
  run "fly 25k frames, n_frames_per_clip=1" --model fly --frames 25000 --steps 3 --warmup 1
  run "LM 10k frames, 40 steps per solve" --solver lm --lm-maxiter 40 --steps 10 --warmup 3
  run "mouse 10k frames" --model mouse --steps 2 --warmup 1
  run "LM 10k frames" --solver lm --steps 10 --warmup 3
  run "LM 100k frames" --solver lm --frames 100000 --steps 3 --warmup 1
  run "LM 40 clips x 250" --solver lm --frames-per-clip 250 --steps 2 --warmup 1
  run "fit mode, 1000 frames as 100 clips" --mode fit --frames 1000 --frames-per-clip 10 --steps 3 --warmup 1
  run "fit mode, one chain of 1000 frames (reference sequencing, config 3)" --mode fit --frames 1000 --frames-per-clip 1000 --steps 1 --warmup 0
  run "run mode: Stac.ik_only end to end, 100k frames at 250 per clip, all outputs + result file" --mode run --frames 100000 --frames-per-clip 250 --steps 2 --warmup 1
  run "strong scaling, N = 1 point: BASELINE configs[3], 1 M frames as 4000 clips of 250 on one GPU" --scaling strong --frames 1000000 --frames-per-clip 250 --steps 1 --warmup 1
  echo "]"
} > $OUT/secondary.json
# the FK output pass on its own: rocprofv3 kernel stats of a run-mode bench (fk_kernel's line: duration against 2 744 B per pose)
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/rp_run && rocprofv3 --kernel-trace --stats -d /tmp/rp_run --output-format csv -- python3 $R/bench.py --mode run --frames 100000 --frames-per-clip 250 --steps 1 --warmup 0 > /tmp/rp_run.log 2>&1; f=$(find /tmp/rp_run -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/run_kernel_stats.csv )
# VALU issue ceiling (micro-benchmark) and the kernel coverage of the GPU suite
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 profiles/tools/valu_issue_micro.hip -o /tmp/valu_issue_micro && /tmp/valu_issue_micro > $OUT/valu_issue_micro.txt 2>&1
bash profiles/tools/suite_coverage.sh > $OUT/suite_coverage.log 2>&1
grep -v "at::native\|__amd_rocclr\|^ *[0-9]* *$" gpurun_out/gpu_suite_kernels.txt > $OUT/gpu_suite_kernels.txt
python3 profiles/tools/lat_sweep.py rodent 250 40 128 256 500 700 1000 1500 2000 3000 > $OUT/lat_sweep.txt 2>&1
python3 profiles/tools/lat_sweep.py fly 100 40 256 500 1000 2000 > $OUT/lat_sweep_fly.txt 2>&1
python3 profiles/tools/lat_sweep.py mouse 20 40 256 500 1000 > $OUT/lat_sweep_mouse.txt 2>&1
echo done
