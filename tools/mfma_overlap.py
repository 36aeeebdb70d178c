"""Diagnostic: scan the gfx950 ISA of a .hip source for MFMAs whose destination registers overlap a source operand other than the
accumulator (hipcc permits D = A / D = B for 4-register results; on gfx950 v_mfma_f32_16x16x32_f16 with D = B returned wrong low-order
bits -- DESIGN 4.3).  usage: python tools/mfma_overlap.py ssd_policy_mfma.hip [-Dflags]"""
import os, re, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(R, "homophily_marl_amd", "csrc", sys.argv[1])
out = "/tmp/mfma_overlap.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-S",
                       "--cuda-device-only", "-o", out, src] + sys.argv[2:], stderr=subprocess.DEVNULL)
def rng(t):
    m = re.match(r"[va]\[(\d+):(\d+)\]", t)
    if m: return (t[0], int(m.group(1)), int(m.group(2)))
    m = re.match(r"([va])(\d+)$", t)
    return (m.group(1), int(m.group(2)), int(m.group(2))) if m else None
def overlap(x, y):
    return x and y and x[0] == y[0] and x[1] <= y[2] and y[1] <= x[2]
kernel, counts = None, {}
for line in open(out):
    m = re.match(r"^(_Z\w+):", line)
    if m: kernel = m.group(1)
    if "v_mfma" not in line: continue
    ops = [o.strip() for o in line.split(None, 1)[1].split(",")]
    d, a, b, c = (rng(o) for o in ops[:4])
    c_ = counts.setdefault(kernel, [0, 0, 0])
    c_[0] += 1
    if overlap(d, a) and d != c: c_[1] += 1
    if overlap(d, b) and d != c: c_[2] += 1
demangle = lambda k: subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0]
bad = 0
for k, (n, da, db) in counts.items():
    if da or db:
        print("%-70s MFMAs %4d   D overlaps A %3d   D overlaps B %3d" % (demangle(k)[:70], n, da, db)); bad += db
print("kernels with MFMAs: %d; products with D over B: %d" % (len(counts), bad))
sys.exit(1 if bad else 0)
