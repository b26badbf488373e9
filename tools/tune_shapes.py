"""tools/tune_shapes.py config2|config5|loop: times the fine kernel for every launch shape the
planner may consider (CSM_PAIR_R x CSM_PAIR_NCBX [x CSM_PAIR_GROUPS]), one bench.py child per
shape, and prints them beside the planner's own choice."""
import json
import os
import subprocess
import sys

which = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = dict(os.environ)
if which == "config2":
    args, env0 = ["--no-configs", "--no-cpu-baseline", "--steps", "3"], {"CSM_BENCH_SCANS": "512", "CSM_BENCH_WINDOWS": "64"}
    shapes = [(r, n, g) for r in (8, 6) for n in (1, 2, 3) for g in (0,)]
    pick = lambda d: (d["roofline"]["avg_launch_us"], d["ms_per_step"])
elif which == "loop":
    args, env0 = ["--workload", "loop", "--steps", "5", "--no-cpu-baseline"], {}
    shapes = [(r, n, g) for r in (8, 6) for n in (1, 2) for g in (0,)]
    pick = lambda d: (d["roofline"]["avg_launch_us"], d["ms_per_step"])
else:
    args, env0 = ["--steps", "1", "--warmup", "1", "--no-cpu-baseline"], {"CSM_BENCH_SCANS": "64", "CSM_BENCH_WINDOWS": "64", "CSM_BENCH_CONFIGS": "config5"}
    shapes = [(r, n, g) for r in (8, 6) for n in (7, 8, 9) for g in (0,)]
    pick = lambda d: (d["configs"]["config5"]["roofline"]["avg_launch_us"], d["configs"]["config5"]["ms_per_query"])
for r, n, g in [(0, 0, 0)] + shapes:
    env = dict(base, **env0)
    if r:
        env.update(CSM_PAIR_R=str(r), CSM_PAIR_NCBX=str(n))
    if g:
        env["CSM_PAIR_GROUPS"] = str(g)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True,
                         text=True, timeout=300, cwd=root)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    if out.returncode or not lines:
        print("R %d ncbx %d groups %d: failed (%s)" % (r, n, g, out.stderr.strip().splitlines()[-1][:100] if out.stderr.strip() else ""))
        continue
    us, ms = pick(json.loads(lines[-1]))
    print("R %d ncbx %d groups %d: fine kernel %.1f us, %.3f ms" % (r, n, g, us, ms), flush=True)
