"""Kernel coverage of the GPU suite: `<calls> <kernel>` for every kernel the traced run launched.

    bash profiles/tools/suite_coverage.sh            (on the GPU box: rocprofv3 --kernel-trace --stats around pytest -m gpu)
    -> gpurun_out/gpu_suite_kernels.txt              (copy to profiles/rNN/; tests/test_isa_hazards.py reads the newest one)
"""
import csv
import re
import sys
from pathlib import Path


def main(d):
    rows = {}
    for f in Path(d).rglob("*kernel_stats.csv"):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                name = re.sub(r"^void ", "", r["Name"])
                name = re.sub(r"\(.*$", "", name).replace("stac::", "")
                rows[name] = rows.get(name, 0) + int(r["Calls"])
    for name in sorted(rows):
        print(f"{rows[name]:8d} {name}")


if __name__ == "__main__":
    main(sys.argv[1])
