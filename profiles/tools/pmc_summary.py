#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter rows per (kernel, counter) over the dispatches of a run.

usage: pmc_summary.py <dir with *_counter_collection.csv> [kernel-name substring]
Prints one JSON object {kernel: {counter: total, "_dispatches": n}}.
"""
import csv, glob, json, sys
from collections import defaultdict

def main():
    root = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    out = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if want not in k:
                continue
            out[k][row["Counter_Name"]] += float(row["Counter_Value"])
            disp[k].add(row["Dispatch_Id"])
            for extra in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size", "LDS_Block_Size", "Workgroup_Size", "Grid_Size"):
                if extra in row:
                    out[k]["_" + extra] = float(row[extra])
    res = {k: dict(v, _dispatches=len(disp[k])) for k, v in out.items()}
    print(json.dumps(res, indent=1))

if __name__ == "__main__":
    main()
