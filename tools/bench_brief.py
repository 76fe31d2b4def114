"""One line per workload: value, ms per step and the per-kernel milliseconds of a bench.py run (development aid)."""
import json, subprocess, sys
args = sys.argv[1:]
out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline"] + args, capture_output=True, text=True)
try:
    d = json.loads(out.stdout.strip().splitlines()[-1])
    print("%.1f %s  %.2f ms/step  %s" % (d["value"], d["unit"], d["ms_per_step"], d["roofline"].get("per_kernel_ms")))
except Exception:
    print(out.stdout[-2000:], out.stderr[-2000:])
