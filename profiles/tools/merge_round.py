#!/usr/bin/env python3
"""Turn what profiles/tools/collect_round.sh left under gpurun_out/<tag>/ into the committed artefacts of a round:

    profiles/<tag>/<tag>a_{bench.json,kernel_stats.csv,pmc.json}      the default bench (BASELINE configs[1])
    profiles/<tag>/<tag>clips_{bench.json,kernel_stats.csv,pmc.json}  250 frames per clip
    profiles/<tag>/secondary.json, lat_sweep.txt
    profiles/valu.json, profiles/traffic.json                         per-step counters bench.py relates to its own timing,
                                                                      tied to the library build they were measured on

usage: python profiles/tools/merge_round.py <tag> [lib digest; default: gpurun_out/<tag>/lib_digest.txt]
Counter totals are divided by the number of dispatches of each kernel in the pass (one bench step per dispatch)."""
import glob
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]


def merged(dirname):
    out = {}
    for f in sorted(glob.glob(str(dirname / "pmc_pass*.json"))):
        for k, v in json.load(open(f)).items():
            n = max(int(v.get("_dispatches", 1)), 1)
            d = out.setdefault(k, {})
            for c, x in v.items():
                d[c] = x if c.startswith("_") else x / n
            d["_dispatches_per_pass"] = n
    return out


def main():
    tag = sys.argv[1]
    src, dst = ROOT / "gpurun_out" / tag, ROOT / "profiles" / tag
    dst.mkdir(parents=True, exist_ok=True)
    digest = sys.argv[2] if len(sys.argv) > 2 else (src / "lib_digest.txt").read_text().strip()
    per = {}
    for leg in ("a", "clips"):
        d = src / f"{tag}{leg}"
        shutil.copy(d / "kernel_stats.csv", dst / f"{tag}{leg}_kernel_stats.csv")
        line = [l for l in open(d / "bench.json").read().splitlines() if l.startswith("{")][-1]
        (dst / f"{tag}{leg}_bench.json").write_text(line + "\n")
        pm = merged(d)
        pm["_note"] = ("rocprofv3 --pmc passes of `python3 bench.py --no-extras` (separate runs per counter set), per bench step "
                       "(= per dispatch of each kernel); SQ_* are sums over the chip; FETCH_SIZE / WRITE_SIZE in KiB; "
                       f"library build {digest[:16]}")
        (dst / f"{tag}{leg}_pmc.json").write_text(json.dumps(pm, indent=1) + "\n")
        per[leg] = pm
    for f in ("secondary.json", "lat_sweep.txt", "lat_sweep_fly.txt", "lat_sweep_mouse.txt", "run_kernel_stats.csv", "valu_issue_micro.txt",
              "gpu_suite_kernels.txt"):
        if (src / f).exists():
            shutil.copy(src / f, dst / f)

    def tot(pm, c):
        return sum(v.get(c, 0.0) for k, v in pm.items() if isinstance(v, dict) and "q_phase_kernel" in k)

    valu = {"note": "SQ_INSTS_VALU (wave-instructions) per bench step from rocprofv3 --pmc (profiles/tools/collect_round.sh), summed over the "
                    "throughput launch and the hand-off launch; bench.py relates it to the evaluations the kernel's counters report and "
                    "uses an entry only when lib_digest is the build it runs on",
            "entries": []}
    for leg, fpc in (("a", 1), ("clips", 250)):
        valu["entries"].append({"frames": 10000, "frames_per_clip": fpc, "model": "rodent", "sq_insts_valu": tot(per[leg], "SQ_INSTS_VALU"),
                                "sq_insts_salu": tot(per[leg], "SQ_INSTS_SALU"), "sq_insts_lds": tot(per[leg], "SQ_INSTS_LDS"),
                                "sq_insts_branch": tot(per[leg], "SQ_INSTS_BRANCH"),
                                "sq_active_inst_valu": tot(per[leg], "SQ_ACTIVE_INST_VALU"), "lib_digest": digest,
                                "source": f"profiles/{tag}/{tag}{leg}_pmc.json"})
    (ROOT / "profiles" / "valu.json").write_text(json.dumps(valu, indent=1) + "\n")
    fk, wk = tot(per["a"], "FETCH_SIZE"), tot(per["a"], "WRITE_SIZE")
    traffic = {"note": "HBM bytes per bench step (the throughput launch + the latency-kernel launch that finishes the handed-off stragglers; both are "
                       "instantiations of stac::q_phase_kernel) from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), (FETCH_SIZE + "
                       "WRITE_SIZE) * 1024; narrow (dword) accesses, so the gfx950 2x FETCH correction for wide coalesced streams is not applied",
               "entries": [{"frames": 10000, "frames_per_clip": 1, "lanes": "auto", "model": "rodent", "solver": "pg", "fetch_kb": fk, "write_kb": wk,
                            "bytes_per_launch": int((fk + wk) * 1024), "lib_digest": digest, "source": f"profiles/{tag}/{tag}a_pmc.json"}]}
    (ROOT / "profiles" / "traffic.json").write_text(json.dumps(traffic, indent=1) + "\n")
    print("wrote", dst)


if __name__ == "__main__":
    main()
