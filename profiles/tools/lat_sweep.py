import sys, os, json, time, subprocess
# usage: python lat_sweep.py  -> runs bench with F=250 for several chain counts and spec G
import itertools
res=[]
for chains in (40, 128, 256, 500, 1000):
    for sg in (8, 16, 32, 64):
        env=dict(os.environ, STAC_HIP_SPECG=str(sg), STAC_HIP_SPEC="1")
        out=subprocess.run([sys.executable,os.path.join(os.path.dirname(os.path.abspath(__file__)),"..","..","bench.py"),"--steps","1","--warmup","1","--frames",str(chains*250),"--frames-per-clip","250","--no-cpu-baseline"],env=env,capture_output=True,text=True)
        try:
            d=json.loads(out.stdout.strip().splitlines()[-1])
            print(chains, sg, "%.0f frames/s"%d["value"], "%.1f ms"%d["roofline"]["kernel_ms"], "us/iter %.2f"%(d["roofline"]["kernel_ms"]*1e3/ (250*d["config"]["iters_per_frame"])), flush=True)
        except Exception as e:
            print(chains, sg, "FAILED", out.stderr[-500:], flush=True)
