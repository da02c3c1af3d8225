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
    """-> {function: [(mnemonic, [operands])]}; see parse_disassembly_addr for the form with addresses."""
    return {k: [(mn, ops) for _, mn, ops in v] for k, v in parse_disassembly_addr(text).items()}


def parse_disassembly_addr(text: str) -> dict[str, list[tuple[int, str, list[str]]]]:
    """-> {function: [(address, mnemonic, [operands])]} from llvm-objdump -d output (address from the trailing comment)."""
    funcs: dict[str, list[tuple[int, str, list[str]]]] = {}
    cur = None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            cur = []
            funcs[m.group(1)] = cur
            continue
        if cur is None or "\t" not in line:
            continue
        head, _, tail = line.partition("//")
        body = head.strip()
        if not body:
            continue
        am = re.match(r"\s*([0-9A-Fa-f]+):", tail)
        addr = int(am.group(1), 16) if am else (cur[-1][0] + 4 if cur else 0)
        parts = body.split(None, 1)
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        cur.append((addr, parts[0], ops))
    return funcs


# ---- scratch first-access analysis --------------------------------------------------------------------------------------
# A register spill slot (or any private-memory word) that a launch READS before it has WRITTEN it holds whatever the previous
# launch left in the wavefront slot's scratch backing store: results then depend on history (round 3: two latency-kernel
# shapes did exactly that).  The same goes for the lanes of the VGPRs that carry spilled SGPRs (v_writelane / v_readlane):
# vector registers are not cleared between wavefronts, so a lane read before it is written returns what the previous
# wavefront on that SIMD left there.  This is a forward "definitely stored" dataflow over the kernel's control-flow graph, rebuilt
# from the disassembly: at every scratch load with an immediate address, every byte it reads must have been stored on EVERY
# path from the kernel's entry.  Loads / stores through a VGPR or SGPR address (private arrays indexed at run time, call
# frames) cannot be resolved statically and are reported as `dynamic`.
_SCRATCH_BYTES = {"byte": 1, "ubyte": 1, "sbyte": 1, "short": 2, "ushort": 2, "sshort": 2, "dword": 4, "dwordx2": 8, "dwordx3": 12,
                  "dwordx4": 16, "ubyte_d16": 1, "ubyte_d16_hi": 1, "sbyte_d16": 1, "sbyte_d16_hi": 1, "short_d16": 2,
                  "short_d16_hi": 2, "byte_d16_hi": 1}
_BRANCHES = ("s_branch", "s_cbranch_scc0", "s_cbranch_scc1", "s_cbranch_vccz", "s_cbranch_vccnz", "s_cbranch_execz",
             "s_cbranch_execnz", "s_cbranch_cdbgsys", "s_cbranch_cdbguser", "s_cbranch_cdbgsys_or_user", "s_cbranch_cdbgsys_and_user")


def _lane_access(mn: str, ops: list[str]):
    """SGPR spills live in lanes of reserved VGPRs: v_writelane_b32 vC, sX, lane / v_readlane_b32 sX, vC, lane.
    -> (is_store, {(vC, lane)} or None when the lane is not an immediate)"""
    is_store = mn.startswith("v_writelane")
    reg = ops[0] if is_store else ops[1]
    lane = ops[2]
    if not re.fullmatch(r"\d+|0x[0-9a-fA-F]+", lane):
        return is_store, None
    return is_store, {(reg, int(lane, 0))}


def _scratch_access(mn: str, ops: list[str]):
    """-> (is_store, byte range or None when the address is not an immediate)"""
    kind = mn.split("_", 2)[2]
    nbytes = _SCRATCH_BYTES.get(kind)
    is_store = mn.startswith("scratch_store")
    # store: vaddr, vdata, saddr [offset:N]; load: vdst, vaddr, saddr [offset:N]
    vaddr = ops[0] if is_store else ops[1]
    last = ops[2].split()
    saddr = last[0]
    off = 0
    for t in last[1:]:
        if t.startswith("offset:"):
            off = int(t[7:], 0)
    if nbytes is None or vaddr != "off" or saddr != "off":
        return is_store, None
    return is_store, range(off, off + nbytes)


def scratch_first_access(ins: list[tuple[int, str, list[str]]]) -> dict:
    """Returns {"loads", "stores", "dynamic", "bad": [(address, mnemonic, offset, bytes never stored on some path)],
    "indirect": n}."""
    n = len(ins)
    addr2idx = {a: i for i, (a, _, _) in enumerate(ins)}
    leaders = {0}
    succ_of_branch: dict[int, int | None] = {}
    indirect = 0
    for i, (a, mn, ops) in enumerate(ins):
        if mn in _BRANCHES:
            simm = int(ops[0], 0)
            if simm >= 0x8000:
                simm -= 0x10000
            tgt = addr2idx.get(a + 4 + 4 * simm)
            succ_of_branch[i] = tgt
            if tgt is not None:
                leaders.add(tgt)
            if i + 1 < n:
                leaders.add(i + 1)
        elif mn == "s_endpgm" and i + 1 < n:
            leaders.add(i + 1)
        elif mn.startswith(("s_setpc", "s_swappc")):
            indirect += 1
            if i + 1 < n:
                leaders.add(i + 1)
    starts = sorted(leaders)
    block_of = {}
    blocks = []
    for bi, st in enumerate(starts):
        en = starts[bi + 1] if bi + 1 < len(starts) else n
        blocks.append((st, en))
        block_of[st] = bi
    preds: list[list[int]] = [[] for _ in blocks]
    for bi, (st, en) in enumerate(blocks):
        a, mn, ops = ins[en - 1]
        outs = []
        if mn == "s_endpgm" or mn.startswith("s_setpc"):
            pass
        elif mn == "s_branch":
            outs.append(succ_of_branch[en - 1])
        elif mn in _BRANCHES:
            outs += [succ_of_branch[en - 1], en if en < n else None]
        else:
            outs.append(en if en < n else None)
        for o in outs:
            if o is not None:
                preds[block_of[o]].append(bi)
    gen = []
    stats = {"loads": 0, "stores": 0, "dynamic": 0, "indirect": indirect, "bad": [], "lane_loads": 0, "lane_stores": 0, "lane_bad": []}

    def access(mn, ops):
        """-> (kind, is_store, set of slots or None); slots are scratch byte offsets or (carrier VGPR, lane) pairs"""
        if mn.startswith("scratch_"):
            is_store, rng = _scratch_access(mn, ops)
            return "scratch", is_store, (set(rng) if rng is not None else None)
        if mn.startswith(("v_writelane", "v_readlane")):
            is_store, slots = _lane_access(mn, ops)
            return "lane", is_store, slots
        return None, False, None

    for st, en in blocks:
        g = set()
        for a, mn, ops in ins[st:en]:
            kind, is_store, slots = access(mn, ops)
            if kind and is_store and slots is not None:
                g.update(slots)
        gen.append(g)
    universe = set().union(*gen) if gen else set()
    IN = [set(universe) for _ in blocks]
    IN[0] = set()
    changed = True
    while changed:
        changed = False
        for bi in range(1, len(blocks)):
            if not preds[bi]:
                new = set()  # unreachable from the entry by the edges we see: nothing can be assumed
            else:
                new = set(universe)
                for p in preds[bi]:
                    new &= IN[p] | gen[p]
            if new != IN[bi]:
                IN[bi] = new
                changed = True
    for bi, (st, en) in enumerate(blocks):
        have = set(IN[bi])
        for a, mn, ops in ins[st:en]:
            kind, is_store, slots = access(mn, ops)
            if kind is None:
                continue
            if slots is None:
                stats["dynamic"] += 1
                continue
            pre = "" if kind == "scratch" else "lane_"
            if is_store:
                stats[pre + "stores"] += 1
                have.update(slots)
            else:
                stats[pre + "loads"] += 1
                missing = [b for b in slots if b not in have]
                if missing:
                    if kind == "scratch":
                        stats["bad"].append((a, mn, min(slots), len(missing)))
                    else:
                        stats["lane_bad"].append((a, mn) + missing[0])
    return stats


def disassembly_addr(co: Path) -> dict[str, list[tuple[int, str, list[str]]]]:
    text = subprocess.run([str(LLVM / "llvm-objdump"), "-d", "--mcpu=gfx950", str(co)], check=True, capture_output=True, text=True).stdout
    return parse_disassembly_addr(text)


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
            meta, dis = kernel_metadata(co), disassembly_addr(co)
            names = list(meta)
            for mangled, nice in zip(names, demangle(names)):
                row = {"kernel": nice, "mangled": mangled, **meta[mangled]}
                ins = dis.get(mangled, [])
                row.update(instruction_mix([(mn, ops) for _, mn, ops in ins]))
                fa = scratch_first_access(ins)
                row.update({"scr_dyn": fa["dynamic"], "scr_bad": len(fa["bad"]), "indirect": fa["indirect"], "scr_bad_list": fa["bad"],
                            "lane_bad": len(fa["lane_bad"]), "lane_bad_list": fa["lane_bad"]})
                rows.append(row)
    rows.sort(key=lambda r: r["kernel"])
    return rows


def format_table(rows: list[dict]) -> str:
    cols = ("vgpr_count", "sgpr_count", "sgpr_spill_count", "vgpr_spill_count", "private_segment_fixed_size", "insts", "valu",
            "salu", "branch", "lds", "smem", "readlane", "writelane", "v_mov", "s_nop", "scratch_ld", "scratch_st", "scr_dyn", "scr_bad", "lane_bad")
    hdr = ("vgpr", "sgpr", "s_spill", "v_spill", "scratch_B", "insts", "valu", "salu", "branch", "lds", "smem", "readlane",
           "writelane", "v_mov", "s_nop", "scr_ld", "scr_st", "scr_dyn", "ld<st", "rdln<wr")
    rows = [dict(r, kernel=re.sub(r"^stac::(\w+)\(.*$", r"\1", r["kernel"])) for r in rows]
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
