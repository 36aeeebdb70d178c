"""Diagnostic: one-screen summary of bench.py JSON lines.  usage: python tools/benchsum.py file.json ..."""
import json
import sys

for f in sys.argv[1:]:
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "UNREADABLE", e)
        continue
    bd = d["config"].get("breakdown_ms", {})
    print("%s: %.1f M  %.2f ms/step  rollout %s train %s" % (f.split("/")[-1], d["value"] / 1e6, d["ms_per_step"], bd.get("rollout_100_steps_ms"), bd.get("learner_train_ms")))
    for k in d.get("kernels", [d["roofline"]]):
        print("    %-28s %6.1f us  frac %.3f %s" % (k["kernel"], k["kernel_avg_us"], k["frac"], k["bound"]))
    if "cpu_baseline" in d:
        print("    cpu %.2f M on %d threads" % (d["cpu_baseline"]["value"] / 1e6, d["cpu_baseline"]["cores"]))
