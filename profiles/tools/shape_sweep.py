import sys, os, json, subprocess
# usage: python shape_sweep.py [model [frames_per_clip [chains ...]]]  -> throughput kernel (STAC_HIP_SPEC=0) against latency kernel
# (STAC_HIP_SPEC=1, its shape chosen automatically) against the automatic choice, for short clips: where the launch heuristics switch
BENCH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "bench.py")
MODEL = sys.argv[1] if len(sys.argv) > 1 else "rodent"
FPC = int(sys.argv[2]) if len(sys.argv) > 2 else 1
CHAINS = [int(c) for c in sys.argv[3:]] or [64, 256, 512, 1024, 2048, 3072, 4096, 6144, 8192, 12288, 16384, 32768]
for chains in CHAINS:
    row = []
    for name, env in (("throughput", {"STAC_HIP_SPEC": "0"}), ("latency", {"STAC_HIP_SPEC": "1"}), ("auto", {})):
        out = subprocess.run([sys.executable, BENCH, "--steps", "3", "--warmup", "1", "--frames", str(chains * FPC), "--frames-per-clip", str(FPC),
                              "--model", MODEL, "--no-cpu-baseline", "--no-extras"], env=dict(os.environ, **env), capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            row.append("%s %.0f frames/s (%.2f ms)" % (name, d["value"], d["ms_per_step"]))
        except Exception:
            row.append(name + " FAILED " + out.stderr[-200:].replace("\n", " "))
    print(chains, "chains x", FPC, "|", " | ".join(row), flush=True)
