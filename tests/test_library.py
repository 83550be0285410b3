"""
CPU tests of the C-ABI shared library: it loads, exports every symbol include/edhip.h declares,
and its argument validation (the checks of _deform_grid.c:121-255) answers with the right status
codes -- all before any HIP call, so no GPU is needed.  No compute is launched here.
"""
import ctypes
import os
import re

import numpy as np
import pytest

from elasticdeform_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(not os.path.exists(_lib.LIB_PATH),
                                reason="libedhip.so not built (run __graft_entry__.build())")


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "edhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(edhip_[a-z0-9_]+)\s*\(", text)))


def test_header_and_library_agree():
    names = _declared_functions()
    assert set(names) == set(_lib.EXPORTS)
    L = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), n
    lib = _lib.load()
    assert lib.edhip_version() == 100
    assert lib.edhip_status_string(2) == b"data type not supported"


def test_header_constants_match_the_python_binding():
    """The flags, limits and dtype codes of include/edhip.h are the ones _lib.py passes."""
    text = open(os.path.join(ROOT, "include", "edhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    enums = {k: int(v) for k, v in re.findall(r"\b(EDHIP_[A-Z0-9_]+)\s*=\s*(\d+)", text)}
    defs = {k: int(v) for k, v in re.findall(r"#define\s+(EDHIP_[A-Z0-9_]+)\s+(\d+)", text)}
    assert enums["EDHIP_FLAG_AUTO"] == _lib.FLAG_AUTO and enums["EDHIP_FLAG_EXACT"] == _lib.FLAG_EXACT
    assert enums["EDHIP_FLAG_FAST"] == _lib.FLAG_FAST
    assert enums["EDHIP_FLAG_RAW_DISPLACEMENT"] == _lib.FLAG_RAW_DISPLACEMENT
    assert enums["EDHIP_FLAG_KEEP_BOXES"] == _lib.FLAG_KEEP_BOXES
    assert enums["EDHIP_FLAG_USE_BOXES"] == _lib.FLAG_USE_BOXES
    assert enums["EDHIP_FLAG_ZERO_GRADIENT"] == _lib.FLAG_ZERO_GRADIENT
    assert enums["EDHIP_FLAG_GRID_STAYS"] == _lib.FLAG_GRID_STAYS
    assert enums["EDHIP_FLAG_SCRATCH_INPUT"] == _lib.FLAG_SCRATCH_INPUT
    assert enums["EDHIP_FLAG_STRONG_FIELD"] == _lib.FLAG_STRONG_FIELD
    flags = [v for k, v in enums.items() if k.startswith("EDHIP_FLAG_") and v]
    assert len(set(flags)) == len(flags) and all(f & (f - 1) == 0 for f in flags)     # distinct bits
    assert defs["EDHIP_MAX_DIMS"] == _lib.MAX_DIMS and defs["EDHIP_MAX_AXES"] == _lib.MAX_AXES
    assert _lib.MAX_AXES == _lib.MAX_DIMS - 1          # the control grid has one dimension more
    for name, code in _lib.DTYPE_CODES.items():
        key = {"bool": "BOOL", "uint8": "U8", "int8": "I8", "uint16": "U16", "int16": "I16", "uint32": "U32",
               "int32": "I32", "uint64": "U64", "int64": "I64", "float32": "F32", "float64": "F64",
               "float16": "F16", "bfloat16": "BF16"}[name]
        assert enums["EDHIP_" + key] == code, name


def _desc(shape, dtype="float32", ptr=0x1000):
    a = np.empty(shape, dtype=dtype)
    return _lib.describe(ptr, a.dtype.name, a.shape, a.strides)


def _call(ins, disp, outs, axis, orders=None, modes=None, cvals=None, off=None, aff=None, grad=False):
    n = len(ins)
    _lib.deform(grad, ins, disp, off, outs, axis, orders or [3] * n, modes or [4] * n,
                cvals or [0.0] * n, aff, _lib.FLAG_AUTO, 0)


def test_validation_errors_map_to_reference_exceptions():
    d2 = _desc((2, 3, 3), "float64")
    x = _desc((8, 9))
    # input / output rank mismatch (_deform_grid.c:137-140)
    with pytest.raises(RuntimeError, match="dimensions should match"):
        _call([x], d2, [_desc((8, 9, 1))], [(0, 1)])
    # axis out of range (:161-165)
    with pytest.raises(RuntimeError, match="invalid axis"):
        _call([x], d2, [x], [(0, 2)])
    # inputs of different deformed size (:166-169)
    with pytest.raises(RuntimeError, match="same size"):
        _call([x, _desc((8, 10))], d2, [x, _desc((8, 10))], [(0, 1), (0, 1)])
    # displacement shape (:178-183)
    with pytest.raises(RuntimeError, match="invalid displacement shape"):
        _call([x], _desc((3, 3, 3), "float64"), [x], [(0, 1)])
    with pytest.raises(RuntimeError, match="invalid displacement shape"):
        _call([x], _desc((2, 3), "float64"), [x], [(0, 1)])
    # spline order / mode ranges
    with pytest.raises(RuntimeError, match="spline order"):
        _call([x], d2, [x], [(0, 1)], orders=[6])
    with pytest.raises(RuntimeError, match="boundary mode"):
        _call([x], d2, [x], [(0, 1)], modes=[9])
    # unsupported dtypes never reach the library: the descriptor builder refuses them with the
    # reference's message (deform.c:744,891)
    with pytest.raises(RuntimeError, match="data type not supported"):
        _lib.describe(0x1000, "complex64", (4, 4), (32, 8))
    # float16 / bfloat16 are storage types of the C ABI (an extension); the public API hands them
    # over only after set_reduced_precision(True) and otherwise answers like the reference
    assert _lib.describe(0x1000, "float16", (4, 4), (8, 2)).dtype == 11
    assert _lib.describe(0x1000, "bfloat16", (4, 4), (8, 2)).dtype == 12
    import importlib
    dgm = importlib.import_module("elasticdeform_amd.deform_grid")
    torch = pytest.importorskip("torch")
    assert not dgm._reduced
    with pytest.raises(RuntimeError, match="data type not supported"):
        dgm._dtype_name(torch.zeros(2, dtype=torch.float16))
    with pytest.raises(RuntimeError, match="data type not supported"):
        dgm._dtype_name(torch.zeros(2, dtype=torch.bfloat16))


def test_filter_validation():
    x = _desc((8, 9))
    with pytest.raises(RuntimeError, match="spline order not supported"):
        _lib.spline_filter1d(x, x, 0, 6, False, 0, 0)
    with pytest.raises(RuntimeError, match="invalid axis"):
        _lib.spline_filter1d(x, x, 2, 3, False, 0, 0)
    with pytest.raises(RuntimeError, match="shapes should match"):
        _lib.spline_filter1d(x, _desc((8, 10)), 0, 3, False, 0, 0)


def test_empty_work_is_ok_without_a_gpu():
    # zero output voxels: validated, nothing launched, EDHIP_OK
    d2 = _desc((2, 3, 3), "float64")
    _call([_desc((8, 9))], d2, [_desc((0, 9))], [(0, 1)])


def test_shipped_library_has_no_environment_switches():
    """profiling / ablation switches must not ship.  The default build never calls
    getenv (the symbol is not even imported), so no environment variable -- EDHIP_HOT_ABL used to select
    kernels that skip the gather -- can change what the library launches or returns; the switches
    exist only behind EDHIP_EXPERIMENTS (make EXPERIMENTS=1), through the one helper ed_env()."""
    import glob
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "elasticdeform_amd", "csrc")
    calls = 0
    for path in glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")):
        text = open(path).read()
        calls += len(re.findall(r"(?<![A-Za-z_])getenv\s*\(", text))
    assert calls <= 1, calls          # the one call inside ed_env()'s EDHIP_EXPERIMENTS branch
    text = open(os.path.join(csrc, "ed_device.h")).read()
    i = text.index("#ifdef EDHIP_EXPERIMENTS")
    assert "getenv" in text[i:text.index("#else", i)] and "getenv" not in text[text.index("#else", i):text.index("#endif", i)]
    lib = os.path.join(root, "elasticdeform_amd", "libedhip.so")
    syms = subprocess.run(["nm", "-D", lib], capture_output=True, text=True).stdout
    if "edhip_experiments_build" in syms:
        pytest.skip("the library in the tree is a profiling build (make EXPERIMENTS=1)")
    assert "getenv" not in syms, "the shipped library imports getenv"
    blob = open(lib, "rb").read()
    for name in (b"EDHIP_HOT_ABL", b"EDHIP_TILE_DBG", b"EDHIP_GRAD_DUO", b"EDHIP_WAVE"):
        assert name not in blob, name


def test_spill_feedback_bookkeeping_host_logic(tmp_path):
    """SpillHint (csrc/ed_workspace.h): the per-stream ring of calls and table of geometries behind the
    level-1 spill feedback, driven from a host-only C++ program (no GPU, no HIP call): reports are matched
    by sequence number, consumed once, garbage and stale reports ignored, old geometries forgotten."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "elasticdeform_amd", "csrc")
    exe = str(tmp_path / "spill_hint_test")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O1", "-std=c++17", "-w", "-I" + csrc,
                    "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cxx", "spill_hint_test.cpp"),
                    os.path.join(csrc, "ed_workspace.hip"), "-o", exe], check=True, timeout=300)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr
