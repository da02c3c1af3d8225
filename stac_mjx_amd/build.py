"""Builds the HIP extension in-tree: ``stac_mjx_amd/csrc/libstac_hip.so`` (gfx950 only).

``hipcc`` cross-compiles without a GPU.  ``-ffp-contract=off -fno-fast-math`` is part of the
arithmetic contract of the kernels (bit-identical to the CPU oracle), not a tuning choice.
"""

from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
LIB = CSRC / "libstac_hip.so"
SOURCES = [CSRC / "stac_kernels.hip", CSRC / "stac_lm.hip", CSRC / "stac_abi.hip"]
HEADERS = [CSRC / "stac_plan.hpp", CSRC / "stac_device.hpp", CSRC.parents[1] / "include" / "stac_hip.h"]
# -amdgpu-opt-vgpr-liverange=false: SIOptimizeVGPRLiveRange is the pass behind round 3's stale-state defect (a latency-kernel
# shape whose results depended on what the previous launch left in registers / scratch): with it off that shape is correct under
# every poison pattern, with it on it is wrong on 36 of 36 models (profiles/r04/stale_spill_repro.txt, reproducer:
# tests/tools/repro_stale_spill/).  Costs nothing on the throughput bench, 2 % on 250-frame clips.
# -amdgpu-sched-strategy=max-ilp: with the split kinematics (round 5) the kernels wait on dependent instructions more than on LDS reads
# (dependent VALU instructions of a wavefront do not overlap: profiles/r04/valu_issue_micro.txt): +3 % on the 10 000-frame bench, +4.6 % on
# 250-frame clips over max-memory-clause, which rounds 3-4 used (profiles/r05/NOTES.md).
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-strict-aliasing", "-fno-slp-vectorize",
         "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-mllvm", "-amdgpu-opt-vgpr-liverange=false"]
FLAGS += os.environ.get("STAC_HIP_EXTRA_FLAGS", "").split()


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libstac_hip.so)")


STAMP = CSRC / "libstac_hip.so.stamp"


def source_digest() -> str:
    """sha256 over the sources, headers and flags: what the library was built from (modification times do not survive a copy
    of the tree to another machine)."""
    import hashlib

    h = hashlib.sha256(" ".join(FLAGS).encode())
    for p in SOURCES + HEADERS:
        h.update(p.name.encode())
        h.update(p.read_bytes())
    return h.hexdigest()


def is_stale() -> bool:
    if not LIB.exists() or not STAMP.exists():
        return True
    return STAMP.read_text().strip() != source_digest()


def build_extension(force: bool = False, verbose: bool = False) -> Path:
    if not force and not is_stale():
        return LIB
    import time

    cmd = [hipcc_path(), *FLAGS, *map(str, SOURCES), "-o", str(LIB)]
    t0 = time.perf_counter()
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    STAMP.write_text(source_digest() + "\n")
    if verbose:
        print(res.stdout + res.stderr)
        print(f"[stac build] {' '.join(cmd)}\n[stac build] compiled in {time.perf_counter() - t0:.0f} s, sources sha256 {source_digest()[:16]}")
    return LIB


if __name__ == "__main__":
    print(build_extension(force=True, verbose=True))
