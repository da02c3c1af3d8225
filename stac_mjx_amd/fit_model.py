"""Host-side fit set-up: model tables, bounds, part masks, trunk mask.

Restates the host work of ``Stac.__init__`` (``stac_mjx/stac.py:98-159``):
site insertion + scaling (``_build_body_spec`` :185-207, ``_init_body_sites`` :209-235),
bounds (``_align_joint_dims`` :54-88), part masks (``part_opt_setup`` :161-183, *substring* match
of config strings against per-qpos joint names), trunk mask (exact membership, :138-140),
root keypoint index (:122-127) and root dims (``compute_stac.py:51-54``).
"""

from __future__ import annotations

from dataclasses import dataclass, field
from pathlib import Path
from typing import Mapping, Sequence

import numpy as np

from .mjcf import JNT_FREE, JNT_SLIDE, ModelTables, align_joint_dims, compile_mjcf


@dataclass
class FitSetup:
    tables: ModelTables
    lb: np.ndarray  # [nq] f32
    ub: np.ndarray  # [nq] f32
    part_names: list[str]  # per-qpos joint names (names_qpos)
    part_masks: np.ndarray  # [P, nq] bool
    part_keys: list[str]
    trunk_kps: np.ndarray  # [K] bool
    root_kp_idx: int  # -1 when ROOT_OPTIMIZATION_KEYPOINT is absent
    root_dims: int  # 7 (free) or 4 (slide root), compute_stac.py:51-54
    is_regularized: np.ndarray  # [K,3] f32 in {0,1}
    kp_names: list[str]
    freejoint: bool = True
    slidejoint: bool = False
    extra: dict = field(default_factory=dict)

    @property
    def fixed_root(self) -> bool:
        return not (self.freejoint or self.slidejoint)

    @property
    def do_root_opt(self) -> bool:
        """Whether fit_offsets/ik_only run root optimisation (stac.py:277-296,397-423)."""
        return self.root_kp_idx != -1 and not self.fixed_root


def _parse_offset(pos) -> list[float]:
    if isinstance(pos, str):
        return [float(p) for p in pos.split()]
    return [float(p) for p in pos]


def build_fit_setup(
    xml_path: str | Path,
    model_cfg: Mapping,
    kp_names: Sequence[str],
    *,
    legacy_joint_pos_scale: bool = False,
    from_string: bool = False,
) -> FitSetup:
    pairs = model_cfg["KEYPOINT_MODEL_PAIRS"]
    offsets = model_cfg["KEYPOINT_INITIAL_OFFSETS"]
    sites = {k: (body, _parse_offset(offsets[k])) for k, body in pairs.items()}
    tables = compile_mjcf(
        xml_path,
        sites=sites,
        scale=float(model_cfg["SCALE_FACTOR"]),
        legacy_joint_pos_scale=legacy_joint_pos_scale,
        from_string=from_string,
    )
    return finish_fit_setup(tables, model_cfg, kp_names)


def finish_fit_setup(tables: ModelTables, model_cfg: Mapping, kp_names: Sequence[str]) -> FitSetup:
    """Everything of :func:`build_fit_setup` that does not need the XML (tables may come from a fixture)."""
    kp_names = list(kp_names)
    lb, ub, part_names = align_joint_dims(tables.jnt_type, tables.jnt_range, tables.jnt_names)

    parts = model_cfg.get("INDIVIDUAL_PART_OPTIMIZATION", None)
    part_keys, masks = [], []
    if parts:
        for key, substrings in parts.items():
            part_keys.append(key)
            masks.append([any(s in name for s in substrings) for name in part_names])
    part_masks = np.asarray(masks, dtype=bool).reshape(len(masks), tables.nq)

    trunk = list(model_cfg.get("TRUNK_OPTIMIZATION_KEYPOINTS", []) or [])
    trunk_kps = np.asarray([n in trunk for n in kp_names], dtype=bool)

    root_kp = model_cfg.get("ROOT_OPTIMIZATION_KEYPOINT", None)
    root_kp_idx = kp_names.index(root_kp) if root_kp is not None else -1

    reg = list(model_cfg.get("SITES_TO_REGULARIZE", []) or [])
    is_reg = np.asarray([[1.0] * 3 if k in reg else [0.0] * 3 for k in tables.site_names], np.float32)

    jt0 = int(tables.jnt_type[0]) if tables.njnt else -1
    return FitSetup(
        tables=tables,
        lb=lb,
        ub=ub,
        part_names=part_names,
        part_masks=part_masks,
        part_keys=part_keys,
        trunk_kps=trunk_kps,
        root_kp_idx=root_kp_idx,
        root_dims=4 if jt0 == JNT_SLIDE else 7,
        is_regularized=is_reg.reshape(tables.nsite, 3),
        kp_names=kp_names,
        freejoint=jt0 == JNT_FREE,
        slidejoint=jt0 == JNT_SLIDE,
    )
