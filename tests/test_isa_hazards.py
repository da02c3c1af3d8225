"""Static check of the built gfx950 code objects for the data hazards that hand-written DPP code can introduce.

The FK program's step (stac_mjx_amd/csrc/stac_device.hpp: quad_rotate, quad_qmul, quad_joint_fused) is inline assembly
with DPP operands.  The assembler does not see inside an asm block, so the wait states that the ISA demands between a VALU
write of a VGPR and a DPP read of it ("VALU writes VGPR -> VALU DPP reads that VGPR: 2 wait states"; "VALU writes EXEC ->
DPP: 5") are placed by hand (`s_nop`).  This test disassembles what was actually built and checks every DPP instruction of
every kernel against its linear predecessors.  CPU only: it needs the library, llvm-objdump and clang-offload-bundler, no GPU.
"""

import re
import shutil
import subprocess
from pathlib import Path

import pytest

from conftest import ROOT

LLVM = Path("/opt/rocm/lib/llvm/bin")
LIB = ROOT / "stac_mjx_amd" / "csrc" / "libstac_hip.so"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _disassemble(tmp_path):
    objcopy = shutil.which("objcopy")
    bundler, objdump = LLVM / "clang-offload-bundler", LLVM / "llvm-objdump"
    if not (objcopy and bundler.exists() and objdump.exists()):
        pytest.skip("binutils / ROCm LLVM tools not available")
    from stac_mjx_amd.build import build_extension

    build_extension()
    fat = tmp_path / "fat.bin"
    subprocess.run([objcopy, "-O", "binary", "--only-section=.hip_fatbin", str(LIB), str(fat)], check=True)
    blob = fat.read_bytes()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    assert starts, "no offload bundle in the library"
    texts = []
    for i, a in enumerate(starts):  # one bundle per translation unit
        chunk = tmp_path / f"bundle{i}.bin"
        chunk.write_bytes(blob[a:(starts[i + 1] if i + 1 < len(starts) else len(blob))])
        co = tmp_path / f"dev{i}.co"
        subprocess.run([str(bundler), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--input={chunk}", f"--output={co}"], check=True, capture_output=True)
        out = subprocess.run([str(objdump), "-d", "--mcpu=gfx950", str(co)], check=True, capture_output=True, text=True).stdout
        texts.append(out)
    return texts


_REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def _vregs(operand):
    m = _REG.search(operand)
    if not m:
        return set()
    if m.group(1) is not None:
        return {int(m.group(1))}
    return set(range(int(m.group(2)), int(m.group(3)) + 1))


def _parse(text):
    """-> list of functions, each a list of (mnemonic, [operands])"""
    funcs, cur = [], None
    for line in text.splitlines():
        if re.match(r"^[0-9a-f]+ <.*>:$", line):
            cur = []
            funcs.append(cur)
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


def _wait_states(ins):
    mn, ops = ins
    if mn == "s_nop":
        return int(ops[0], 0) + 1
    return 1


def _hazards(texts):
    n_dpp = 0
    bad = []
    for text in texts:
        for f in _parse(text):
            for i, (mn, ops) in enumerate(f):
                if not any("quad_perm" in o or "row_" in o or "wave_" in o for o in ops) and not mn.endswith("_dpp"):
                    continue
                n_dpp += 1
                src0 = _vregs(ops[1].split()[0]) if len(ops) > 1 else set()  # the operand the DPP control applies to
                ws, j = 0, i - 1
                while j >= 0 and ws < 5:
                    pm, pops = f[j]
                    if pm.startswith("v_cmpx"):
                        bad.append(("exec", mn, ops, j - i))
                    if ws < 2 and pm.startswith("v_") and pops and (_vregs(pops[0].split()[0]) & src0) and not pm.startswith("v_cmp"):
                        bad.append(("vgpr", mn, ops, f[j]))
                    ws += _wait_states(f[j])
                    j -= 1
    return n_dpp, bad


def test_the_checker_sees_a_hazard():
    fake = """
0000000000001000 <k>:
\tv_add_f32_e32 v1, v2, v3                                     // 000000001000: 02020702
\tv_mov_b32_dpp v4, v1 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1 // 000000001004: 7E0802FA FF081501
\tv_add_f32_e32 v5, v2, v3                                     // 00000000100c: 02020702
\ts_nop 1                                                      // 000000001010: BF800001
\tv_fmac_f32_dpp v6, v5, v7 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1 // 000000001014: 760C0EFA FF08AA05
"""
    n, bad = _hazards([fake])
    assert n == 2 and len(bad) == 1 and bad[0][0] == "vgpr" and bad[0][1] == "v_mov_b32_dpp"


def test_no_dpp_read_after_write_hazard(tmp_path):
    n_dpp, bad = _hazards(_disassemble(tmp_path))
    assert n_dpp > 1000, f"the FK program's DPP code was expected in the library ({n_dpp} DPP instructions found)"
    assert not bad, f"{len(bad)} DPP hazards, e.g. {bad[:3]}"
