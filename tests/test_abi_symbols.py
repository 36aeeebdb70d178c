"""CPU suite: the C-ABI shared library loads and exports every symbol include/ssd_hip.h declares (no compute calls)."""
import ctypes
import os
import re

from homophily_marl_amd import abi

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _declared():
    src = open(os.path.join(ROOT, "include", "ssd_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ssd_[a-z_0-9]+)\s*\(", src)))


def test_header_and_ctypes_table_agree():
    assert _declared() == sorted(abi.HIP_SIGNATURES)


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    lib = ctypes.CDLL(abi.HIP_LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), name
    lib.ssd_abi_version.restype = ctypes.c_int
    assert lib.ssd_abi_version() == abi.ABI_VERSION


def test_struct_sizes_match_the_header(tmp_path):
    """sizeof() of every ABI struct as seen by a C compiler == the ctypes mirror."""
    import subprocess
    c = tmp_path / "s.c"
    c.write_text('#include <stdio.h>\n#include "ssd_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(ssd_config),'
                 'sizeof(ssd_tape), sizeof(ssd_step_out), sizeof(ssd_obs_out), sizeof(ssd_state), sizeof(ssd_info),'
                 'sizeof(ssd_store_step), sizeof(ssd_policy_head), sizeof(ssd_policy_encode_args), sizeof(ssd_block_copy), sizeof(ssd_adam_job), sizeof(ssd_clip_adam_args), sizeof(ssd_row_gather));return 0;}')
    exe = tmp_path / "s"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert sizes == [ctypes.sizeof(t) for t in (abi.SsdConfig, abi.SsdTape, abi.SsdStepOut, abi.SsdObsOut, abi.SsdState, abi.SsdInfo,
                                                    abi.SsdStoreStep, abi.SsdPolicyHead, abi.SsdPolicyEncodeArgs, abi.SsdBlockCopy, abi.SsdAdamJob, abi.SsdClipAdamArgs, abi.SsdRowGather)]


def test_oracle_is_not_reachable_from_the_product_package():
    """The product package must never import or link the checker."""
    pkg = os.path.join(ROOT, "homophily_marl_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.replace("The CPU oracle under oracle/ is a separate", "") or f in (), (dp, f)
