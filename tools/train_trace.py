"""Diagnostic: the ordered launch list of ONE train step from a rocprofv3 kernel trace of tools/train_prof.py.
   box:  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tt -- python3 $R/tools/train_prof.py) && python3 tools/train_trace.py /tmp/tt
Prints start offset, duration and the gap to the previous kernel's end for every launch between two k_clip_adam launches."""
import csv
import glob
import sys

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
# (--memory-copy-trace: SDMA / blit copies are not kernels; show them in the same timeline)
for cf in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(cf)):
        rows.append(dict(Start_Timestamp=r["Start_Timestamp"], End_Timestamp=r["End_Timestamp"], Kernel_Name="[memory copy] %s" % r.get("Direction", "")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "k_clip_adam" in r["Kernel_Name"]]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(ends) - 3
a, b = ends[skip] + 1, ends[skip + 1] + 1
t0 = int(rows[a]["Start_Timestamp"]); prev = t0; busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"]
    n = n.replace("at::native::", "").replace("(anonymous namespace)::", "").replace("void ", "")
    print("%8.1f %7.1f %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, n[:150]))
    prev = e; busy += e - s
print("launches %d, busy %.1f us, span %.1f us" % (b - a, busy / 1e3, (prev - t0) / 1e3))
