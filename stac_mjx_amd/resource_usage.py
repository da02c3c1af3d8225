"""Per-kernel resource table of a built ``libstac_hip.so``: registers, spills, scratch, static instruction mix.

Reads the gfx950 code objects out of the library's fat binary (``objcopy`` + ``clang-offload-bundler``), their
metadata notes (``llvm-readelf --notes``) and their disassembly (``llvm-objdump -d``).  CPU only.

    python -m stac_mjx_amd.resource_usage [lib.so] > profiles/rNN/resource_usage.txt

Used by ``tests/test_isa_hazards.py`` (spill gate, scratch first-access check) and by the round's profile collection.
"""

from __future__ import annotations

import re
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

LLVM = Path("/opt/rocm/lib/llvm/bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def tools_available() -> bool:
    return bool(shutil.which("objcopy")) and all((LLVM / t).exists() for t in ("clang-offload-bundler", "llvm-objdump", "llvm-readelf"))


def code_objects(lib: Path, tmp: Path) -> list[Path]:
    """Unbundle every gfx950 code object of the library (one per translation unit)."""
    fat = tmp / "fat.bin"
    subprocess.run([shutil.which("objcopy"), "-O", "binary", "--only-section=.hip_fatbin", str(lib), str(fat)], check=True)
    blob = fat.read_bytes()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    if not starts:
        raise RuntimeError(f"no offload bundle in {lib}")
    out = []
    for i, a in enumerate(starts):
        chunk = tmp / f"bundle{i}.bin"
        chunk.write_bytes(blob[a:(starts[i + 1] if i + 1 < len(starts) else len(blob))])
        co = tmp / f"dev{i}.co"
        subprocess.run([str(LLVM / "clang-offload-bundler"), "--unbundle", "--type=o", f"--targets={TARGET}",
                        f"--input={chunk}", f"--output={co}"], check=True, capture_output=True)
        out.append(co)
    return out


def demangle(names: list[str]) -> list[str]:
    cf = shutil.which("c++filt")
    if not cf or not names:
        return names
    res = subprocess.run([cf], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    return [r.replace("void stac::", "").replace("(stac::QArgs)", "").replace("(stac::QArgs, stac::LmArgs)", "") for r in res]


_META_KEYS = ("vgpr_count", "agpr_count", "sgpr_count", "sgpr_spill_count", "vgpr_spill_count", "private_segment_fixed_size",
              "group_segment_fixed_size", "kernarg_segment_size")


def kernel_metadata(co: Path) -> dict[str, dict[str, int]]:
    """mangled kernel name -> {vgpr_count, sgpr_spill_count, ...} from the code object's AMDGPU metadata note."""
    txt = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(co)], check=True, capture_output=True, text=True).stdout
    out: dict[str, dict[str, int]] = {}
    for blk in txt.split("- .agpr_count:")[1:]:
        blk = ".agpr_count:" + blk
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        out[name] = {k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1)) for k in _META_KEYS}
    return out


def disassembly(co: Path) -> dict[str, list[tuple[str, list[str]]]]:
    """mangled function name -> [(mnemonic, [operands])] in address order."""
    text = subprocess.run([str(LLVM / "llvm-objdump"), "-d", "--mcpu=gfx950", str(co)], check=True, capture_output=True, text=True).stdout
    return parse_disassembly(text)


def parse_disassembly(text: str) -> dict[str, list[tuple[str, list[str]]]]:
    funcs: dict[str, list[tuple[str, list[str]]]] = {}
    cur = None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            cur = []
            funcs[m.group(1)] = cur
            continue
        if cur is None or "\t" not in line:
            continue
        body = line.split("//")[0].strip()
        if not body:
            continue
        parts = body.split(None, 1)
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        cur.append((parts[0], ops))
    return funcs


def instruction_mix(ins: list[tuple[str, list[str]]]) -> dict[str, int]:
    mix = {"insts": len(ins), "valu": 0, "salu": 0, "branch": 0, "lds": 0, "vmem": 0, "smem": 0, "readlane": 0, "writelane": 0,
           "v_mov": 0, "s_nop": 0, "scratch_ld": 0, "scratch_st": 0}
    for mn, _ in ins:
        if mn.startswith("v_readlane") or mn.startswith("v_readfirstlane"):
            mix["readlane"] += mn.startswith("v_readlane")
            mix["valu"] += 1
        elif mn.startswith("v_writelane"):
            mix["writelane"] += 1
            mix["valu"] += 1
        elif mn.startswith("v_"):
            mix["valu"] += 1
            mix["v_mov"] += mn.startswith("v_mov_b32")
        elif mn.startswith("s_cbranch") or mn == "s_branch" or mn.startswith("s_setpc") or mn.startswith("s_swappc"):
            mix["branch"] += 1
        elif mn == "s_nop":
            mix["s_nop"] += 1
        elif mn.startswith("s_load") or mn.startswith("s_buffer_load") or mn.startswith("s_store"):
            mix["smem"] += 1
        elif mn.startswith("s_"):
            mix["salu"] += 1
        elif mn.startswith("ds_"):
            mix["lds"] += 1
        elif mn.startswith("scratch_load"):
            mix["scratch_ld"] += 1
        elif mn.startswith("scratch_store"):
            mix["scratch_st"] += 1
        elif mn.startswith(("global_", "flat_", "buffer_")):
            mix["vmem"] += 1
    return mix


def table(lib: Path) -> list[dict]:
    """One row per kernel of the library: metadata + static instruction mix."""
    rows = []
    with tempfile.TemporaryDirectory() as td:
        for co in code_objects(lib, Path(td)):
            meta, dis = kernel_metadata(co), disassembly(co)
            names = list(meta)
            for mangled, nice in zip(names, demangle(names)):
                row = {"kernel": nice, "mangled": mangled, **meta[mangled]}
                row.update(instruction_mix(dis.get(mangled, [])))
                rows.append(row)
    rows.sort(key=lambda r: r["kernel"])
    return rows


def format_table(rows: list[dict]) -> str:
    cols = ("vgpr_count", "sgpr_count", "sgpr_spill_count", "vgpr_spill_count", "private_segment_fixed_size", "insts", "valu",
            "salu", "branch", "lds", "smem", "readlane", "writelane", "v_mov", "s_nop", "scratch_ld", "scratch_st")
    hdr = ("vgpr", "sgpr", "s_spill", "v_spill", "scratch_B", "insts", "valu", "salu", "branch", "lds", "smem", "readlane",
           "writelane", "v_mov", "s_nop", "scr_ld", "scr_st")
    w = max(len(r["kernel"]) for r in rows)
    lines = ["kernel".ljust(w) + " " + " ".join(h.rjust(9) for h in hdr)]
    for r in rows:
        lines.append(r["kernel"].ljust(w) + " " + " ".join(str(r[c]).rjust(9) for c in cols))
    return "\n".join(lines)


if __name__ == "__main__":
    from .build import LIB, source_digest

    lib = Path(sys.argv[1]) if len(sys.argv) > 1 else LIB
    print(f"# {lib.name}" + (f"  sources sha256 {source_digest()[:16]}" if lib == LIB else ""))
    print(format_table(table(lib)))
