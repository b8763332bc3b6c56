"""The C-ABI shared library loads and exports every symbol include/trt.h declares
(no compute calls: there is no GPU where these run)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT
from toroidal_ray_tracing_amd import abi, lib


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "trt.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(trt_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert _declared_functions() == sorted(lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    L = lib.load()
    for name in _declared_functions():
        assert hasattr(L, name), name
    assert L.trt_version() == 3   # TRT_VERSION_MAJOR*1000 + TRT_VERSION_MINOR


def test_release_library_reads_no_environment():
    """Every TRT_* knob and the timing ablations are compiled out of libtrt.so (they exist in the
    -DTRT_TUNING build libtrt_tuning.so that tools/ load): the release library imports no getenv."""
    import subprocess
    so = os.path.join(ROOT, "toroidal_ray_tracing_amd", "libtrt.so")
    dyn = subprocess.run(["nm", "-D", "--undefined-only", so], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in dyn
    assert not hasattr(lib.load(), "trt_debug_reload_tuning")
    tuning = os.path.join(ROOT, "toroidal_ray_tracing_amd", "libtrt_tuning.so")
    if os.path.exists(tuning):
        assert "getenv" in subprocess.run(["nm", "-D", "--undefined-only", tuning], capture_output=True, text=True, check=True).stdout


def test_struct_layouts_match_header_comments():
    assert C.sizeof(abi.trt_stats) == 64
    assert C.sizeof(abi.trt_globals) == 204   # 3 mat4 + vec3
    assert C.sizeof(abi.trt_push) == 44       # BEF PushConstantRay
    assert C.sizeof(abi.trt_material) == 80   # WaveFrontMaterial, scalar layout
    assert C.sizeof(abi.trt_torus) == 24
    assert C.sizeof(abi.trt_rendered_data) == 64
    assert abi.trt_push.rho.offset == 40 and abi.trt_push.maxDepth.offset == 36
    assert abi.trt_material.illum.offset == 72 and abi.trt_material.shininess.offset == 60
    assert abi.trt_globals.center.offset == 192


def test_header_compiles_as_c_and_cxx(tmp_path):
    import subprocess
    hdr = os.path.join(ROOT, "include", "trt.h")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", hdr], check=True)
    subprocess.run(["g++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", hdr], check=True)


def test_no_cpu_fallback_without_device():
    """Product path fails loudly when no HIP device is usable."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from toroidal_ray_tracing_amd.tracer import Tracer, TrtError
    with pytest.raises(TrtError) as e:
        Tracer(0)
    assert e.value.code == abi.TRT_E_NO_DEVICE and "no CPU fallback" in str(e.value)
    L = lib.load()
    assert L.trt_set_solver(None, 0) == abi.TRT_E_INVALID
    L.trt_destroy(None)  # harmless


def test_product_never_imports_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, "toroidal_ray_tracing_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "trt_oracle" not in txt and "oracle/" not in txt, f
