"""Config loading that keeps the reference's ``configs/*.yaml`` surface without Hydra/OmegaConf.

The reference composes ``configs/config.yaml`` (a Hydra ``defaults:`` list selecting one file of
the ``stac/`` group and one of the ``model/`` group) and merges the result into the structured
schema ``Config(model: ModelConfig, stac: StacConfig)`` (``stac_mjx/config.py:11-88``).  This
module does the same with PyYAML: ``defaults`` list, ``group=name`` / dotted ``a.b=value``
overrides, schema check, attribute access.
"""

from __future__ import annotations

from pathlib import Path
from typing import Any, Iterable

import yaml

# Required keys (reference: stac_mjx/config.py:11-62).  MARKER_SIZE has a default there (:36).
_MODEL_REQUIRED = (
    "MJCF_PATH", "FTOL", "ROOT_FTOL", "LIMB_FTOL", "N_ITERS", "N_ITER_Q", "KP_NAMES",
    "KEYPOINT_MODEL_PAIRS", "KEYPOINT_INITIAL_OFFSETS", "ROOT_OPTIMIZATION_KEYPOINT",
    "TRUNK_OPTIMIZATION_KEYPOINTS", "INDIVIDUAL_PART_OPTIMIZATION", "KEYPOINT_COLOR_PAIRS",
    "SCALE_FACTOR", "MOCAP_SCALE_FACTOR", "SITES_TO_REGULARIZE", "RENDER_FPS", "N_SAMPLE_FRAMES",
    "M_REG_COEF",
)  # fmt: skip
_MODEL_DEFAULTS = {"MARKER_SIZE": 0.005}
_STAC_REQUIRED = (
    "fit_offsets_path", "ik_only_path", "data_path", "n_fit_frames", "skip_fit_offsets",
    "skip_ik_only", "infer_qvels", "n_frames_per_clip", "mujoco", "continuous",
)  # fmt: skip
_STAC_OPTIONAL = ("num_clips",)
_MUJOCO_REQUIRED = ("solver", "iterations", "ls_iterations")
# Engine extensions (not in the reference schema); all optional.  (lm_maxiter: accepted steps per solve of solver = lm, default 20
# -- 40 until round 3; gather: auto | rank0 | all | none, resolved per run by main.run_stac, the caller's config is never modified)
_STAC_EXTENSIONS = ("solver", "lanes_per_chain", "device", "time_indices", "fit_frames_per_clip", "reference_marker_order", "gather", "gather_max_bytes", "lm_maxiter")
_MODEL_EXTENSIONS = ("KP_NAMES_LABEL3D_PATH",)


class ConfigError(ValueError):
    pass


class ConfigNode(dict):
    """dict with attribute access and OmegaConf-like ``get``/``in`` behaviour."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def to_dict(self) -> dict:
        def conv(v):
            if isinstance(v, ConfigNode):
                return {k: conv(x) for k, x in v.items()}
            if isinstance(v, list):
                return [conv(x) for x in v]
            return v

        return conv(self)

    def to_yaml(self) -> str:
        return yaml.safe_dump(self.to_dict(), sort_keys=False)


def _wrap(v: Any) -> Any:
    if isinstance(v, dict):
        return ConfigNode({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, list):
        return [_wrap(x) for x in v]
    return v


def _load_yaml(path: Path) -> dict:
    with open(path, "r") as fh:
        return yaml.safe_load(fh) or {}


def _parse_scalar(text: str) -> Any:
    try:
        return yaml.safe_load(text)
    except yaml.YAMLError:
        return text


def compose_config(
    config_path: Path | str,
    config_name: str = "config",
    overrides: Iterable[str] | None = None,
) -> ConfigNode:
    """Compose ``<config_path>/<config_name>.yaml`` like ``stac_mjx.config.compose_config`` (config.py:73-88)."""
    config_dir = Path(config_path).resolve()
    top_path = config_dir / f"{config_name}.yaml"
    if not top_path.exists():
        raise ConfigError(f"config file not found: {top_path}")
    top = _load_yaml(top_path)
    overrides = list(overrides or [])

    group_choice: dict[str, str] = {}
    for item in top.pop("defaults", []) or []:
        if isinstance(item, dict):
            for g, name in item.items():
                group_choice[str(g)] = str(name)
        elif item != "_self_":
            raise ConfigError(f"unsupported defaults entry: {item!r}")
    dotted: list[tuple[str, Any]] = []
    for ov in overrides:
        if ov.startswith("hydra/") or ov.startswith("hydra."):
            continue  # hydra logging switches the reference adds (config.py:80)
        if "=" not in ov:
            raise ConfigError(f"override must be key=value: {ov!r}")
        key, val = ov.split("=", 1)
        key = key.lstrip("+")
        if "." not in key and key in ("stac", "model"):
            group_choice[key] = val
        else:
            dotted.append((key, _parse_scalar(val)))

    cfg: dict[str, Any] = {}
    for group, name in group_choice.items():
        gpath = config_dir / group / f"{name}.yaml"
        if not gpath.exists():
            raise ConfigError(f"config group file not found: {gpath}")
        cfg[group] = _load_yaml(gpath)
    for k, v in top.items():  # _self_ comes last in the reference's defaults lists
        if isinstance(v, dict) and isinstance(cfg.get(k), dict):
            cfg[k].update(v)
        else:
            cfg[k] = v
    for key, val in dotted:
        node = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = val
    return validate_config(cfg)


def validate_config(cfg: dict) -> ConfigNode:
    """Schema check equivalent to the OmegaConf structured merge (config.py:86-88)."""
    if "model" not in cfg or "stac" not in cfg:
        raise ConfigError("config must define both 'model' and 'stac' groups")
    model = dict(_MODEL_DEFAULTS)
    model.update(cfg["model"])
    stac = dict(cfg["stac"])
    missing = [k for k in _MODEL_REQUIRED if k not in model]
    # the reference tolerates absent optional model keys only where it probes with `in`/get
    # (ROOT_OPTIMIZATION_KEYPOINT, INDIVIDUAL_PART_OPTIMIZATION, SITES_TO_REGULARIZE: stac.py:122,174,229)
    soft = {"ROOT_OPTIMIZATION_KEYPOINT", "INDIVIDUAL_PART_OPTIMIZATION", "SITES_TO_REGULARIZE",
            "KEYPOINT_COLOR_PAIRS", "RENDER_FPS", "ROOT_FTOL", "LIMB_FTOL", "KP_NAMES"}  # fmt: skip
    hard_missing = [k for k in missing if k not in soft]
    if hard_missing:
        raise ConfigError(f"model config is missing keys: {hard_missing}")
    unknown = [k for k in model if k not in _MODEL_REQUIRED and k not in _MODEL_DEFAULTS and k not in _MODEL_EXTENSIONS]
    if unknown:
        raise ConfigError(f"model config has keys outside the schema: {unknown}")
    missing = [k for k in _STAC_REQUIRED if k not in stac]
    if missing:
        raise ConfigError(f"stac config is missing keys: {missing}")
    unknown = [k for k in stac if k not in _STAC_REQUIRED and k not in _STAC_OPTIONAL and k not in _STAC_EXTENSIONS]
    if unknown:
        raise ConfigError(f"stac config has keys outside the schema: {unknown}")
    missing = [k for k in _MUJOCO_REQUIRED if k not in stac["mujoco"]]
    if missing:
        raise ConfigError(f"stac.mujoco config is missing keys: {missing}")
    # light type coercion like OmegaConf's structured merge
    for k in ("FTOL", "SCALE_FACTOR", "MOCAP_SCALE_FACTOR", "M_REG_COEF", "MARKER_SIZE"):
        model[k] = float(model[k])
    for k in ("N_ITERS", "N_ITER_Q", "N_SAMPLE_FRAMES"):
        model[k] = int(model[k])
    for k in ("n_fit_frames", "n_frames_per_clip"):
        stac[k] = int(stac[k])
    for k in ("skip_fit_offsets", "skip_ik_only", "infer_qvels", "continuous"):
        if not isinstance(stac[k], bool):
            raise ConfigError(f"stac.{k} must be a bool")
    return _wrap({"model": model, "stac": stac})


def load_configs(config_dir: Path | str, config_name: str = "config") -> ConfigNode:
    """Same signature as ``stac_mjx.main.load_configs`` (main.py:18-30)."""
    return compose_config(config_dir, config_name=config_name)
