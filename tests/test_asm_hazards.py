"""CPU check of the compiled code: no wide store of the kernels is followed by a write to its data registers without wait states.

hipcc pads that hazard for global stores but not for raw buffer stores with an SGPR soffset (LLVM's hazard recogniser exempts the form);
the persistent NT kernel of csrc/gemm8.hip stored an address instead of a value now and then before every such store got a fence
(DESIGN.md 14.3).  The files that use 12- or 16-byte raw buffer stores are compiled to gfx950 assembly here (no GPU needed) and scanned by
tools/scan_store_hazards.py; the other files contain no such builtin (checked by text)."""
import glob
import importlib.util
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "chainer-speech-recognition_amd", "csrc")


def _scanner():
    spec = importlib.util.spec_from_file_location("scan_store_hazards", os.path.join(ROOT, "tools", "scan_store_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_wide_raw_buffer_stores_are_fenced_and_compile_without_the_hazard():
    scan = _scanner()
    wide = re.compile(r"__builtin_amdgcn_raw_buffer_store_b(96|128)\s*\(")
    users = [f for f in sorted(glob.glob(os.path.join(CSRC, "*.hip"))) if wide.search(open(f).read())]
    assert [os.path.basename(f) for f in users] == ["gemm8.hip"]
    for f in users:
        lines = open(f).read().split("\n")
        for i, line in enumerate(lines):
            if wide.search(line):
                assert "ASR8_STORE_FENCE();" in lines[i + 1], (os.path.basename(f), i + 1, line.strip()[:80])
        stores, found = scan.hazards(scan.assembly(f))
        assert stores > 0 and not found, found[:3]


def test_the_scanner_sees_the_hazard():
    scan = _scanner()
    asm = """
        buffer_store_dwordx4 v[64:67], v68, s[20:23], s27 offen
        v_cndmask_b32_e64 v64, -16, v212, s[0:1]
        buffer_store_dwordx4 v[92:95], v64, s[20:23], s27 offen
        s_nop 3
        v_mov_b32_e32 v92, v1
        global_store_dwordx4 v[2:3], v[8:11], off
        v_add_u32_e32 v5, v6, v7
        v_mov_b32_e32 v9, v1
    """
    stores, found = scan.hazards(asm)
    assert stores == 3 and len(found) == 2
    assert found[0][1].startswith("v_cndmask_b32_e64 v64") and found[1][1].startswith("v_mov_b32_e32 v9")
