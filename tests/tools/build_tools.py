"""Builders of the GPU suite's native test tools (not product code; used by tests/conftest.py and __graft_entry__.build())."""

from __future__ import annotations

import subprocess
from pathlib import Path

TOOLS = Path(__file__).resolve().parent
POISON_SO = TOOLS / "libpoison.so"


def build_poison_tool() -> Path:
    """tests/tools/poison_scratch.hip -> tests/tools/libpoison.so (the scratch / register poisoner of the GPU suite)."""
    from stac_mjx_amd.build import hipcc_path

    src = TOOLS / "poison_scratch.hip"
    if not POISON_SO.exists() or POISON_SO.stat().st_mtime < src.stat().st_mtime:
        subprocess.run([hipcc_path(), "--offload-arch=gfx950", "-O1", "-shared", "-fPIC", str(src), "-o", str(POISON_SO)], check=True)
    return POISON_SO
