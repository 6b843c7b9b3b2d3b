"""CPU: the C-ABI library builds, loads, and exports exactly the symbols include/aqualora_hip.h declares, with the
argument counts the ctypes binding uses.  No compute call is made (no GPU here)."""
import os
import re

from tests.conftest import ROOT


def _header_decls():
    text = open(os.path.join(ROOT, "include", "aqualora_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(?:int|long|const char\*)\s+(aql_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("void", "") else len([a for a in args.split(",") if a.strip()])
        decls[m.group(1)] = n
    return decls


def test_header_matches_binding_and_library():
    import __graft_entry__ as ge
    ge.build()
    from aqualora_amd import _lib
    decls = _header_decls()
    lib = _lib.load()
    bound = dict((k, len(v)) for k, v in _lib.SIGNATURES.items())
    for name, n in decls.items():
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
        if name in bound:
            assert bound[name] == n, (name, bound[name], n)
    for name in bound:
        assert name in decls, f"{name} bound in _lib.py but missing from include/aqualora_hip.h"
    assert set(decls) - set(bound) <= {"aql_last_error", "aql_groupnorm_scratch_floats", "aql_bn_scratch_floats", "aql_prvl_scratch_floats"}
    # the version the header declares == the one the binding was written against == the one the built library reports
    hdr = open(os.path.join(ROOT, "include", "aqualora_abi.h")).read()   # the one definition (included by aqualora_hip.h and aql_comm.hip)
    import glob
    defs = [f for f in glob.glob(os.path.join(ROOT, "include", "*.h")) + glob.glob(os.path.join(ROOT, "aqualora_amd", "csrc", "*.*"))
            if f.endswith((".h", ".hip", ".cuh")) and re.search(r"#define\s+AQL_ABI_VERSION", open(f).read())]
    assert [os.path.basename(f) for f in defs] == ["aqualora_abi.h"], defs
    ver = int(re.search(r"#define\s+AQL_ABI_VERSION\s+(\d+)", hdr).group(1))
    assert ver == _lib.ABI_VERSION == lib.aql_abi_version()
    # the SURVEY section 8(b) minimum set of the exchange is exported
    for name in ("aql_comm_init", "aql_comm_all_reduce_f32", "aql_comm_reduce_scatter_f32", "aql_comm_all_gather", "aql_comm_destroy"):
        assert name in decls and hasattr(lib, name)


def test_product_path_fails_loudly_without_gpu():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a CPU-only box")
    from aqualora_amd import _lib, lora
    host = lora.LoRACompatibleLinear(16, 16)
    with pytest.raises(_lib.AqlError):
        host(torch.zeros(2, 4, 16))
    from aqualora_amd.watermark import MapperNet
    with pytest.raises(_lib.AqlError):
        MapperNet(8, 8)(torch.zeros(1, 8))


def test_no_kernel_spills_registers_or_uses_scratch():
    """Every gfx950 kernel of the built library: zero spilled VGPRs, no scratch segment (metadata notes of the objects the
    build leaves next to the library, tools/check_spills.py).  A spill inside a K loop is a silent 10-20 %."""
    import sys
    import __graft_entry__ as ge
    ge.build()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_spills
    n, bad = check_spills.scan()
    assert n >= 250, n
    assert not bad, bad[:5]


def test_no_mfma_under_an_exec_mask_without_a_skip_branch():
    """MFMA ignores EXEC.  hipcc may predicate a short guarded block with `s_and_saveexec` and no `s_cbranch_execz`; an MFMA in such
    a block runs for a wavefront whose guard is false, on uninitialised operands (lora_down_skinny_kernel: NaN for K < 256, fixed
    in round 3 by making the guard provably wave-uniform).  Scan the disassembly of every built gfx950 code object."""
    import sys
    import __graft_entry__ as ge
    ge.build()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_mfma_exec
    n, bad = check_mfma_exec.scan()
    assert n >= 5000, n
    assert not bad, bad[:5]
