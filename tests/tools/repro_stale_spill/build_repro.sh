#!/bin/bash
# Rebuilds the round-3 library WITH the latency-kernel shape that round 3 dropped (q_phase_kernel<8,16,2,8>: 16 solver registers
# per lane at 8 lanes per role; 150 spilled VGPRs + 235 SGPRs spilled into VGPR lanes, 320 B of scratch) from this repository's
# own history, under four sets of compiler flags.  Run from the repository root; results: build/libr3_{A,B,C,D}.so
#   A  round 3's flags (-amdgpu-sched-strategy=max-memory-clause)                      -> wrong answers, launch-to-launch differences
#   B  A + -amdgpu-spill-sgpr-to-vgpr=0                                               -> still wrong
#   C  A + -amdgpu-opt-vgpr-liverange=false   (SIOptimizeVGPRLiveRange off)           -> correct under every poison pattern
#   D  default scheduler, pass on                                                      -> correct (the failing allocation does not arise)
# then on a GPU box:  for v in A B C D; do STAC_HIP_LIB=build/libr3_$v.so python tests/tools/repro_stale_spill/repro_dropped.py 60 7FC00000; done
set -e
REV=37c81a6   # "round 3: VERDICT + ADVICE + BENCH": the tree round 3 ended on
W=build/r3src
mkdir -p $W
for f in stac_kernels.hip stac_device.hpp stac_plan.hpp stac_abi.hip stac_lm.hip; do git show $REV:stac_mjx_amd/csrc/$f > $W/$f; done
git show $REV:include/stac_hip.h > $W/include_stac_hip.h
python3 - <<'PY'
W = "build/r3src/"
s = open(W + "stac_kernels.hip").read()
a, b = s.index("        STAC_TRY_SPEC(8, 10, 2, 4)"), s.index("#undef STAC_TRY_SPEC")
s = s[:a] + "        STAC_TRY_SPEC(8, 16, 2, 8) STAC_TRY_SPEC(8, 10, 2, 8)\n" + s[b:]          # the dropped shape back in
a, b = s.index("    STAC_TRY(4, 20) STAC_TRY(4, 32)"), s.index("#undef STAC_TRY\n")
s = s[:a] + "    STAC_TRY(16, 5) STAC_TRY(16, 8) STAC_TRY(16, 16)\n" + s[b:]                    # (fewer instantiations: faster build)
open(W + "k_repro.hip", "w").write(s)
t = open(W + "stac_abi.hip").read()
t = t.replace("        if (sg == 8 && m->h.nq > 80) sg = 16;\n", "")                              # ... and reachable
t = t.replace('#include "../../include/stac_hip.h"', '#include "include_stac_hip.h"')
open(W + "abi_repro.hip", "w").write(t)
PY
cd $W
base="--offload-arch=gfx950 -O3 -fPIC -shared -std=c++17 -ffp-contract=off -fno-fast-math -fno-strict-aliasing -fno-slp-vectorize"
mc="-mllvm -amdgpu-sched-strategy=max-memory-clause"
b() { n=$1; shift; /opt/rocm/bin/hipcc $base "$@" k_repro.hip stac_lm.hip abi_repro.hip -o ../libr3_$n.so && echo built $n; }
b A $mc & b B $mc -mllvm -amdgpu-spill-sgpr-to-vgpr=0 & b C $mc -mllvm -amdgpu-opt-vgpr-liverange=false & b D & wait
