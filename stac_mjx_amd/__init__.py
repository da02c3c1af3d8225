"""stac_mjx_amd -- MI355X-native STAC pose-fitting engine (hot path of talmolab/stac-mjx).

Public API mirrors ``stac_mjx/__init__.py:3-6`` for the hot path: ``load_configs``, ``load_data``,
``run_stac``; plus ``Stac`` / ``StacCore`` / ``Engine`` for direct use.
"""

from .config import compose_config, load_configs  # noqa: F401
from .io import StacData, load_data, load_stac_data, save_data_to_h5  # noqa: F401

__all__ = ["compose_config", "load_configs", "load_data", "load_stac_data", "save_data_to_h5", "StacData"]


def __getattr__(name):  # lazy: these import torch / need the HIP extension
    if name in ("Engine", "StacHipError", "load_library"):
        from . import engine

        return getattr(engine, name)
    if name in ("Stac",):
        from . import stac

        return getattr(stac, name)
    if name in ("StacCore", "MOptResult"):
        from . import stac_core

        return getattr(stac_core, name)
    if name in ("run_stac",):
        from . import main

        return getattr(main, name)
    raise AttributeError(name)
