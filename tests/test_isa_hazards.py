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


def _kernel_args(isa_path, prefix):
    """{mangled kernel name: [(offset, size), ...]} from the code-object metadata of every kernel whose name starts with prefix"""
    import re
    text = open(isa_path).read()
    meta = text[text.index("amdhsa.kernels"):]
    out = {}
    for block in meta.split("  - .agpr_count:")[1:]:
        m = re.search(r"\.name:\s+(\S+)", block)
        if m and m.group(1).startswith(prefix):
            out[m.group(1)] = [(int(o), int(s)) for o, s in re.findall(r"\.offset:\s+(\d+)\n\s+\.size:\s+(\d+)", block.split(".group_segment_fixed_size")[0])]
    return out


def test_late_fetched_kernel_arguments_sit_where_the_kernels_read_them():
    """k_env, k_head and k_inc_encode read part of their arguments from the kernarg segment by OFFSET (cold_kernarg / cold_ptr /
    refetch_head_args), behind leading scalar arguments that gfx950 preloads into SGPRs.  The offsets are constants in the sources;
    this checks them against the layout the compiler actually emitted (a mismatch reads pointers from the wrong bytes: a GPU fault)."""
    import re
    import asm_hazards
    src_env = open(os.path.join(ROOT, "homophily_marl_amd", "csrc", "ssd_env.hip")).read()
    src_pol = open(os.path.join(ROOT, "homophily_marl_amd", "csrc", "ssd_policy_mfma.hip")).read()
    head_lead = eval(re.search(r"constexpr int HEAD_LEAD_BYTES = ([0-9*+ ]+);", src_pol).group(1))
    fused_lead = eval(re.search(r"constexpr int LEAD = ([0-9*+ ]+);", src_pol).group(1))
    assert "kEnvArgsOffset = (5 * 8 + 4 * 4" in src_env
    env = _kernel_args(asm_hazards.isa_of("ssd_env.hip"), "_ZN3ssd5k_envI")
    assert len(env) >= 10
    for name, args in env.items():
        assert args[-1][0] == 5 * 8 + 4 * 4 and args[-1][1] > 300, (name, args)           # EnvArgs right behind 5 pointers + 4 ints (8-aligned)
    pol = asm_hazards.isa_of("ssd_policy_mfma.hip")
    heads = _kernel_args(pol, "_ZN3ssd6k_headI")
    assert len(heads) >= 16
    for name, args in heads.items():
        (ko, ks), (co, cs) = args[-2], args[-1]
        assert ko == head_lead and co == ko + ks, (name, args)                             # HeadK, then HeadCold (HEAD_COLD_OFFSET = sizeof(HeadK))
    fused = _kernel_args(pol, "_ZN3ssd12k_inc_encodeI")
    assert len(fused) >= 8
    for name, args in fused.items():
        (ko, ks), (co, cs) = args[-3], args[-2]
        assert ko == fused_lead and co == ko + ks, (name, args)
    # the leading scalars are preloaded (14 dwords) in the three rollout kernels
    text = open(pol).read()
    assert text.count(".amdhsa_user_sgpr_kernarg_preload_length 14") >= len(heads) + len(fused)
