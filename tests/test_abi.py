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


def test_library_holds_both_instances_of_the_consensus_kernel():
    """The consensus kernel is built twice, in two translation units (racon_amd/csrc/poa_kernel2.hpp): the instance for eight
    work-groups per CU and the one for a work-group that has a CU to itself (engine_deep.hip); and the int32 fallback kernel."""
    blob = open(os.path.join(ROOT, "racon_amd", "csrc", "libracon_hip.so"), "rb").read()
    for name in (b"_ZN3rcn18poa_window_kernel2ENS_7KParamsE", b"_ZN3rcn23poa_window_kernel2_deepENS_7KParamsE", b"_ZN3rcn17poa_window_kernelENS_7KParamsE"):
        assert name in blob, name
