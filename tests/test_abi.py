"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and
exports every symbol include/wxhip.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "wxhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(wx_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(built_lib):
    lib = ctypes.CDLL(built_lib)
    names = _declared_symbols()
    assert "wx_euler3d_rhs" in names and "wx_last_error" in names
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/wxhip.h but not exported"


def test_binding_covers_header(built_lib):
    from wxfactory_amd import _lib

    assert sorted(_lib.SIGNATURES) == _declared_symbols()
    lib = _lib.load()
    assert lib.wx_version().decode().startswith("wxhip")


def test_argument_errors_are_reported_without_a_gpu(built_lib):
    """Validation happens before any HIP call, with a message (reference kernels fail silently:
    pde/interface.cu:190-210)."""
    from wxfactory_amd import _lib

    lib = _lib.load()
    h = ctypes.c_void_p()
    st = lib.wx_euler3d_plan_create(ctypes.byref(h), 9, 4, 2, 31, 0, 0, None, None)
    assert st == 1 and b"null" in lib.wx_last_error()
    ops = _lib.DfrOps()
    m = _lib.Euler3DMetric()
    st = lib.wx_euler3d_plan_create(ctypes.byref(h), 9, 4, 2, 31, 0, 0, ctypes.byref(ops), ctypes.byref(m))
    assert st == 2 and b"num_solpts" in lib.wx_last_error()
    st = lib.wx_euler3d_rhs(None, None, None, None, 0, None)
    assert st == 1
    with pytest.raises(_lib.WxError):
        _lib.check(st, "wx_euler3d_rhs")


def test_no_cpu_fallback_in_product():
    """The product package never imports the oracle."""
    pkg = os.path.join(ROOT, "wxfactory_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "oracle/" not in src, f


def test_header_is_plain_c_and_the_c_example_links(built_lib, tmp_path):
    """include/wxhip.h compiles as C99 (no C++, no torch types) and examples/c_abi_rhs.c links against the library:
    the boundary is usable from C as it stands (the program itself runs under -m gpu)."""
    import shutil
    import subprocess

    gcc, rocm = shutil.which("gcc"), os.environ.get("ROCM_PATH", "/opt/rocm")
    if gcc is None or not os.path.exists(os.path.join(rocm, "include", "hip", "hip_runtime_api.h")):
        pytest.skip("no C toolchain / HIP headers here")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "only_header.c"
    src.write_text('#include "wxhip.h"\nint main(void) { return wx_version() == 0; }\n')
    libdir = os.path.dirname(built_lib)
    # (-pedantic for the header alone: the HIP runtime's own headers, which the example includes, are not pedantic C)
    for source, out, extra in ((str(src), "only_header", ["-pedantic"]),
                               (os.path.join(root, "examples", "c_abi_rhs.c"), "c_abi_rhs", [])):
        subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", *extra, "-D__HIP_PLATFORM_AMD__", f"-I{rocm}/include",
                        f"-I{root}/include", source, f"-L{libdir}", "-lwxhip", f"-L{rocm}/lib", "-lamdhip64", "-lm", "-o",
                        str(tmp_path / out)], check=True, capture_output=True, text=True)
