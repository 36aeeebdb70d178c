"""Diagnostic: what one bench iteration launches OUTSIDE the rollout's three kernels and the train step's kernels -- the episode's
opening and closing graphs, the replay insert -- from a rocprofv3 kernel trace of bench.py.
   box:  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/it -- python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline) && python3 tools/iter_trace.py /tmp/it"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "k_clip_adam" in r["Kernel_Name"]]
a, b = ends[-3] + 1, ends[-2] + 1                                       # one whole iteration: train(k) end .. train(k+1) end
t0 = int(rows[a]["Start_Timestamp"]); prev = t0
short = lambda n: n.replace("at::native::", "").replace("(anonymous namespace)::", "").replace("void ", "")[:110]
steps = 0
for i in range(a, b):
    r = rows[i]; s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"]); n = r["Kernel_Name"]
    rollout = any(k in n for k in ("k_inc_encode", "k_env<2", "k_head<0"))
    if rollout:
        steps += 1
        if steps > 6 and "Kernel_Name" in r and i + 8 < b and any(k in rows[i + 8]["Kernel_Name"] for k in ("k_inc_encode", "k_env<2", "k_head<0")):
            prev = e
            continue                                                     # the middle of the rollout: not listed
    print("%9.1f %7.1f %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, short(n)))
    prev = e
print("iteration span %.1f us" % ((prev - t0) / 1e3))
