#!/usr/bin/env python3
"""Static check of the generated gfx950 ISA for a hazard the compiler does not cover: gfx950 does NOT interlock an MFMA's result write
against a later vector instruction that reads (or overwrites) the same VGPRs -- software has to keep enough wait states between
them.  hipcc's hazard recogniser inserts them for the instructions it selects itself, but it does not look into INLINE ASM: an asm
statement whose input is an MFMA accumulator can be scheduled right behind the MFMA and read the register before it is written
(round 4: leaky_split8's `v_pk_mul_f32` in k_encode<15, 2, false, 5>).

For every instruction inside an `;;#ASMSTART ... ;;#ASMEND` region this script looks back over the previous instructions of the same basic
block(s) for a v_mfma whose destination overlaps one of the VGPRs the asm instruction reads or writes, and reports the pair when
fewer than MIN_GAP instructions lie between them (v_mfma_f32_16x16x32_f16 is a 4-pass MFMA: passes + 3, + 1 on gfx950 = 8 wait states
between its issue and a vector read of its result, as LLVM's GCNHazardRecognizer counts them for the instructions it can see; every
instruction is at least one wait state and `s_nop N` is N + 1, so MIN_GAP = 9 instructions is that bound with one to spare).

    python tools/asm_hazards.py ssd_policy_mfma.hip [ssd_gru_seq.hip ...]      (compiles each to ISA, device only)
Exit code 1 when a pair is found."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MIN_GAP = 9
SOURCES = ["ssd_policy_mfma.hip", "ssd_gru_seq.hip", "ssd_bmm.hip", "ssd_policy.hip"]      # every source with MFMAs or inline asm
LOOKBACK = 40

_vreg = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def vregs(text):
    out = set()
    for m in _vreg.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def isa_of(src, extra=()):
    """device ISA of one source with the product's flags, cached under build/isa/ by modification time"""
    from __graft_entry__ import CSRC, HIPCC_FLAGS
    flags = [f for f in HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    d = os.path.join(ROOT, "build", "isa")
    os.makedirs(d, exist_ok=True)
    out = os.path.join(d, os.path.basename(src) + ".s")
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")] + [os.path.join(ROOT, "include", "ssd_hip.h")]
    if extra or not os.path.exists(out) or any(os.path.getmtime(x) > os.path.getmtime(out) for x in deps):
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + list(extra) + ["--cuda-device-only", "-S", "-o", out, os.path.join(CSRC, src)],
                              stderr=subprocess.DEVNULL)
    return out


def scan(path):
    """[(function, line number, asm instruction, mfma instruction, instructions between)]"""
    found = []
    func = "?"
    window = []            # (text, is_mfma, dst regs) of the last LOOKBACK instructions
    in_app = False
    for ln, raw in enumerate(open(path), 1):
        line = raw.split(";", 1)[0].strip() if not raw.lstrip().startswith(";") else ""
        tag = raw.strip()
        if tag.startswith(";;#ASMSTART") or tag.startswith(";APP"):
            in_app = True
            continue
        if tag.startswith(";;#ASMEND") or tag.startswith(";NO_APP"):
            in_app = False
            continue
        if not line:
            continue
        if line.endswith(":"):
            if not line.startswith("."):
                func = line[:-1]
                window = []
            continue             # local labels: a predecessor block may end in an MFMA -- keep the window (conservative)
        if line.startswith("."):
            continue
        op = line.split()[0]
        ops = line[len(op):]
        if in_app and op.startswith("v_"):
            touched = vregs(ops)
            for back, (t, is_mfma, dst) in enumerate(reversed(window)):
                if is_mfma and (dst & touched) and back < MIN_GAP:
                    found.append((func, ln, line, t, back))
                    break
        is_mfma = op.startswith("v_mfma") or op.startswith("v_smfmac")
        dst = vregs(ops.split(",")[0]) if is_mfma else set()
        # s_nop N counts as N + 1 wait states
        reps = 1 + int(ops.strip()) if op == "s_nop" and ops.strip().isdigit() else 1
        for _ in range(reps):
            window.append((line, is_mfma, dst))
        window = window[-LOOKBACK:]
    return found


def main():
    bad = 0
    for src in sys.argv[1:] or SOURCES:
        path = src if src.endswith(".s") else isa_of(src)
        hits = scan(path)
        print("%s: %d inline-asm vector instruction(s) within %d instructions of an MFMA that writes their registers" % (src, len(hits), MIN_GAP))
        for func, ln, ins, mf, gap in hits[:20]:
            print("   %s:%d  %s   <-  %s   (%d instructions between)" % (func[:60], ln, ins, mf, gap))
        bad += len(hits)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
