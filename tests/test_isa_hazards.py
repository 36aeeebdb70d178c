"""CPU suite: static check of the generated gfx950 ISA (tools/asm_hazards.py).  gfx950 does not interlock an MFMA's result write
against a vector instruction that reads the register; hipcc inserts the wait states for the instructions it selects, but not for
inline asm.  Round 4 found the encoder's `leaky_split8` reading a conv accumulator two instructions behind its MFMA in one
instantiation (k_encode<15, 2, false, 5>: batch tile 0 of every workgroup 1e-3 off).  The check compiles every source that holds
MFMAs or inline asm to ISA (cross-compiled, no GPU) and fails on any inline-asm vector instruction closer than 9 instructions to an
MFMA that writes one of its registers; a synthetic ISA snippet proves the scanner sees the pattern."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_scanner_flags_an_asm_read_right_behind_an_mfma(tmp_path):
    import asm_hazards
    bad = tmp_path / "bad.s"
    bad.write_text("k:\n\tv_mfma_f32_16x16x32_f16 v[98:101], v[110:113], v[114:117], v[98:101]\n\tv_max_f32 v1, v2, v3\n"
                   "\t;;#ASMSTART\n\tv_pk_mul_f32 v[92:93], v[98:99], v[86:87]\n\t;;#ASMEND\n")
    hits = asm_hazards.scan(str(bad))
    assert len(hits) == 1 and hits[0][4] == 1
    ok = tmp_path / "ok.s"
    ok.write_text("k:\n\tv_mfma_f32_16x16x32_f16 v[98:101], v[110:113], v[114:117], v[98:101]\n\ts_nop 7\n\tv_max_f32 v1, v2, v3\n"
                  "\t;;#ASMSTART\n\tv_pk_mul_f32 v[92:93], v[98:99], v[86:87]\n\t;;#ASMEND\n")
    assert asm_hazards.scan(str(ok)) == []
    # the same read by a compiler-selected instruction is the hazard recogniser's business, not this scanner's
    vis = tmp_path / "vis.s"
    vis.write_text("k:\n\tv_mfma_f32_16x16x32_f16 v[98:101], v[110:113], v[114:117], v[98:101]\n\tv_mul_f32 v92, v98, v86\n")
    assert asm_hazards.scan(str(vis)) == []


def test_no_inline_asm_reads_an_mfma_result_too_early():
    import asm_hazards
    for src in asm_hazards.SOURCES:
        hits = asm_hazards.scan(asm_hazards.isa_of(src))
        assert not hits, (src, hits[:5])
