"""libflatgfa.so loads without a GPU and exports exactly what include/flatgfa.h declares."""
import os
import re
import subprocess

from conftest import ROOT
from pollen_amd import _lib

HEADER = os.path.join(ROOT, "include", "flatgfa.h")


def declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(flatgfa_[a-z0-9_]+)\s*\(", text))


def exported():
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    return {ln.split()[-1] for ln in out.splitlines() if " T " in ln and "flatgfa_" in ln}


def test_header_symbols_are_exported():
    decl, exp = declared(), exported()
    assert decl, "no declarations parsed"
    assert decl <= exp, f"declared but not exported: {sorted(decl - exp)}"
    assert exp <= decl, f"exported but not declared in flatgfa.h: {sorted(exp - decl)}"


def test_reference_flatgfa_c_surface_is_complete():
    # the eight functions of flatgfa-c/src/lib.rs:63-172
    ref = {"flatgfa_parse", "flatgfa_free", "flatgfa_get_segment_count", "flatgfa_get_seq", "flatgfa_path_count",
           "flatgfa_get_path_name", "flatgfa_get_path_step_count", "flatgfa_get_step"}
    assert ref <= exported()


def test_ctypes_table_covers_the_header():
    assert set(_lib.SIGNATURES) == declared()
    lib = _lib.lib()  # loads; every symbol resolves
    assert lib.flatgfa_last_error() is not None
    assert isinstance(lib.flatgfa_device_count(), int)


def test_no_oracle_in_product():
    # the product never imports, links or shells out to anything under oracle/
    bad = []
    for d, _dirs, files in os.walk(os.path.join(ROOT, "pollen_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h", "Makefile")):
                if "oracle" in open(os.path.join(d, f), errors="replace").read().replace("oracle/synth.py", ""):
                    bad.append(os.path.join(d, f))
    assert not bad, bad


def test_pinned_landing_registers_are_private():
    """k_scan's in-flight step tiles land in fixed VGPRs; nothing else in the generated ISA may
    touch them (tools/check_pinned_vgprs.py, also `make -C pollen_amd/csrc check`)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_pinned_vgprs.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
