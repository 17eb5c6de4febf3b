"""Runs bench.py on the workloads named on the command line (default c3 c5) and prints one short line each."""
import sys, json, subprocess
for cfg in (sys.argv[1:] or ["c3", "c5"]):
    out = subprocess.run([sys.executable,"bench.py","--steps","20","--warmup","5","--no-cpu-baseline","--workload",cfg.lower()],capture_output=True,text=True)
    try:
        j=json.loads(out.stdout.strip().splitlines()[-1])
        print(cfg, j["value"], j["ms_per_step"], j.get("render_ms"), {k:v["ms"] for k,v in j["kernels"].items()}, flush=True)
    except Exception as e:
        print(cfg, "ERR", out.stdout[-500:], out.stderr[-1500:]); sys.exit(1)
