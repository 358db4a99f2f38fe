"""The C-ABI shared library loads and exports every symbol include/racon_hip.h
declares (no compute calls: there is no GPU in the CPU test tier)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    h = open(os.path.join(ROOT, "include", "racon_hip.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(rcn_[a-z_]+)\s*\(", h)))


def test_header_declares_the_expected_entry_points():
    syms = declared_symbols()
    for s in ["rcn_engine_create", "rcn_engine_destroy", "rcn_engine_upload", "rcn_engine_run", "rcn_engine_result",
              "rcn_engine_add_window", "rcn_engine_has_windows", "rcn_engine_generate_consensus", "rcn_engine_reset"]:
        assert s in syms


def test_library_exports_every_declared_symbol(hip_lib):
    for s in declared_symbols():
        assert hasattr(hip_lib, s), s
    from racon_amd import engine
    assert set(engine.EXPORTS) == set(declared_symbols())
    assert b"gfx950" in hip_lib.rcn_version()


def test_strerror_and_argument_checks(hip_lib):
    assert hip_lib.rcn_strerror(0) == b"ok"
    assert b"no CPU fallback" in hip_lib.rcn_strerror(-1)
    assert hip_lib.rcn_engine_create(None, None) == -3          # RCN_E_ARG


def test_no_cpu_fallback(hip_lib):
    """Without a GPU the product path must fail loudly, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from racon_amd.engine import HipEngine
    with pytest.raises(RuntimeError):
        HipEngine()


def test_product_does_not_link_the_oracle():
    src = ""
    for root, _, files in os.walk(os.path.join(ROOT, "racon_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", "Makefile")):
                src += open(os.path.join(root, f), errors="ignore").read()
    assert "oracle_lib" not in src and "liboracle" not in src and "poa_oracle" not in src


def test_product_libraries_are_not_sanitizer_builds():
    """A sanitizer build of the host layer (CPU-side checks: make CXXFLAGS=-fsanitize=address) left in the tree travels to
    the GPU box like any built .so, and every process that dlopens it dies in the sanitizer's start-up check -- silently under
    pytest's capture.  __graft_entry__.build() rebuilds such a library; this test names it."""
    import __graft_entry__ as ge
    for rel in ("racon_amd/host/libracon_host.so", "racon_amd/host/racon_hip", "racon_amd/csrc/libracon_hip.so", "oracle/liboracle.so"):
        assert not ge._instrumented(os.path.join(ROOT, rel)), f"{rel} is a sanitizer build: make -C {os.path.dirname(rel)} clean all"


def test_library_holds_both_instances_of_the_consensus_kernel():
    """The consensus kernel is built twice, in two translation units (racon_amd/csrc/poa_kernel2.hpp): the instance for eight
    work-groups per CU and the one for a work-group that has a CU to itself (engine_deep.hip); and the int32 fallback kernel."""
    blob = open(os.path.join(ROOT, "racon_amd", "csrc", "libracon_hip.so"), "rb").read()
    for name in (b"_ZN3rcn18poa_window_kernel2ENS_7KParamsE", b"_ZN3rcn23poa_window_kernel2_deepENS_7KParamsE", b"_ZN3rcn17poa_window_kernelENS_7KParamsE"):
        assert name in blob, name


def test_library_holds_the_small_window_kernel():
    """poa_small.hpp's kernel (one wave per window, graph in LDS) is a third translation unit of the same library."""
    blob = open(os.path.join(ROOT, "racon_amd", "csrc", "libracon_hip.so"), "rb").read()
    assert b"_ZN3rcn23poa_window_kernel_smallENS_7KParamsE" in blob


def test_python_mirrors_have_the_c_structs_sizes(tmp_path):
    """The ctypes mirrors of the ABI's structs (racon_amd/engine.py) against the C compiler's view of include/racon_hip.h."""
    import subprocess
    from racon_amd import engine
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "racon_hip.h"\nint main(void) { printf("%zu %zu %zu %zu\\n", sizeof(rcn_run_stats), '
                   'sizeof(rcn_engine_config), sizeof(rcn_window_refs), sizeof(rcn_reserve_hint)); return 0; }\n')
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    sizes = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert sizes == [C.sizeof(engine.RcnRunStats), C.sizeof(engine.RcnEngineConfig), C.sizeof(engine.RcnWindowRefs), C.sizeof(engine.RcnReserveHint)]


def test_engine_switches_are_read_once_at_creation():
    """No getenv on the launch path: the engine's RCN_* switches are read by read_knobs() when an engine is created, behind
    RCN_EXPERIMENT=1."""
    src = open(os.path.join(ROOT, "racon_amd", "csrc", "engine.hip")).read()
    body = src[src.index("static Knobs read_knobs()"):]
    body = body[:body.index("\n}\n") + 3]
    assert src.count("getenv") == body.count("getenv") and "RCN_EXPERIMENT" in body
    for h in ("poa_small.hpp", "poa_kernel2.hpp", "poa_band.hpp", "pair_align.hpp", "window_build.hpp"):
        assert "getenv" not in open(os.path.join(ROOT, "racon_amd", "csrc", h)).read(), h
