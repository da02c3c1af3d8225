import sys, os, json, subprocess
# usage: python lat_sweep.py [model [frames_per_clip [chains ...]]]  -> the bench at 250 frames per clip for several chain counts:
# every forced latency-mode shape (STAC_HIP_SPECG = lanes per evaluation role) and the automatic choice
BENCH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "bench.py")
MODEL = sys.argv[1] if len(sys.argv) > 1 else "rodent"
FPC = int(sys.argv[2]) if len(sys.argv) > 2 else 250
CHAINS = [int(c) for c in sys.argv[3:]] or [40, 128, 256, 500, 600, 700, 800, 900, 1000]
for chains in CHAINS:
    for sg in (8, 16, 32, 64, "auto"):
        env = dict(os.environ)
        if sg != "auto":
            env.update(STAC_HIP_SPECG=str(sg), STAC_HIP_SPEC="1")
        out = subprocess.run([sys.executable, BENCH, "--steps", "1", "--warmup", "1", "--frames", str(chains * FPC), "--frames-per-clip", str(FPC), "--model", MODEL,
                              "--no-cpu-baseline", "--no-extras"], env=env, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            print(chains, sg, "%.0f frames/s" % d["value"], "%.1f ms" % d["roofline"]["kernel_ms"],
                  "us/iter %.2f" % (d["roofline"]["kernel_ms"] * 1e3 / (FPC * d["config"]["iters_per_frame"])), flush=True)
        except Exception:
            print(chains, sg, "FAILED", out.stderr[-500:], flush=True)
