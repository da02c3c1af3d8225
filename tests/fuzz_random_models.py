"""Not collected by pytest: `python tests/fuzz_random_models.py N` runs tests/test_gpu_parity.py::test_random_models_bit_exact
for N further seeds on a GPU box (random trees with free / ball / slide joints and oriented bodies, HIP == oracle bit for
bit at 8 lanes, 16 lanes and in latency mode); combine with STAC_HIP_SPEC / STAC_HIP_SPECG / STAC_HIP_FLAGS to aim at one
kernel shape.  Round 2: 2 300 models over the shapes, no mismatch."""
import os, sys, traceback
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import pytest
import test_gpu_parity as T

if os.environ.get("STAC_FUZZ_POISON"):
    # fill the scratch memory and every vector register of every SIMD with a pattern before every launch
    # (tests/tools/poison_scratch.hip): state that a launch reads before writing it then poisons the FIRST launch instead of
    # depending on the previous one.  STAC_FUZZ_POISON=<hex pattern> (1 = 0xFFFFFFFF)
    import ctypes, torch
    from conftest import build_poison_tool
    from stac_mjx_amd import engine as _E
    _P = ctypes.CDLL(str(build_poison_tool()))
    _P.poison_scratch.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
    _pat = os.environ["STAC_FUZZ_POISON"]
    _pat = 0xFFFFFFFF if _pat == "1" else int(_pat, 16)
    _orig = _E.Engine.q_phase
    def _poisoned(self, *a, **k):
        assert _P.poison_scratch(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), _pat) == 0
        return _orig(self, *a, **k)
    _E.Engine.q_phase = _poisoned
bad = 0; ran = 0
big = len(sys.argv) > 2 and sys.argv[2] == "big"  # `N big`: the large-tree / many-chain flavour (test_random_models_large_trees_many_chains)
lean = len(sys.argv) > 2 and sys.argv[2] in ("lean", "leanwide", "leanbq", "leanwidebq")  # `N lean`: trees of the lean kernels' kind (split kinematics), several chain counts
wide = len(sys.argv) > 2 and sys.argv[2] in ("leanwide", "leanwidebq")  # `N leanwide`: 97 .. 256 coordinates (the wide lean shapes)
bq = len(sys.argv) > 2 and sys.argv[2].endswith("bq")  # `N leanbq` / `N leanwidebq`: with oriented bodies below the root (round 6: the fly's kind)
for seed in range(12, 12 + int(sys.argv[1])):
    try:
        if lean:
            T._random_lean_case(seed, chains=int((5, 40, 300)[seed % 3]), frames=1 + seed % 3, wide=wide,
                                p_oriented=(0.15, 0.4, 0.7)[(seed // 3) % 3] if bq else None)
        elif big:
            T._random_model_case(seed, 40, 110, int((7, 70, 300)[seed % 3]), 1 + seed % 3, (8, 16, 32, 64, 0), q_init=bool(seed % 2))
        else:
            T.test_random_models_bit_exact(seed)
        ran += 1
    except pytest.skip.Exception:
        pass
    except Exception as e:
        bad += 1; print("SEED", seed, "FAILED:", repr(e)[:300]); traceback.print_exc(limit=2)
print("ran", ran, "failed", bad)
