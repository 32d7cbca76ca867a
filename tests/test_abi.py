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


@pytest.mark.parametrize("world,k,loopback", [(1, 1, False), (1, 1, True), (2, 1, False), (3, 1, False), (6, 1, False),
                                              (4, 2, False), (8, 2, False), (8, 1, False), (1, 2, True), (5, 1, False),
                                              (24, 2, False), (54, 3, False)])
def test_native_exchange_layout_equals_the_host_mirror(built_lib, world, k, loopback):
    """The C-ABI exchange (csrc/exchange.hip: tile graph, ownership and slot order from csrc/wx_panels.h) and the Python
    mirror (exchange.PanelExchange, pinned against the halos the reference delivered on 6 and 24 ranks by
    tests/test_exchange_gloo.py) name the same tiles, buffer sizes, per-rank message sizes and slot of every tile edge.
    Host-only calls: no GPU, no communicator."""
    import torch

    from wxfactory_amd import _lib
    from wxfactory_amd.exchange import PanelExchange

    lib = _lib.load()
    ec = 40
    for rank in range(world):
        py = PanelExchange(ec, "cpu", rank=rank, world_size=world, loopback=loopback, tiles_per_side=k)
        h = ctypes.c_void_p()
        _lib.check(lib.wx_exchange_create(ctypes.byref(h), None, rank, world, k, ec, int(loopback)), "wx_exchange_create")
        try:
            n = lib.wx_exchange_local_tiles(h, None, 0)
            tiles = (ctypes.c_int * max(n, 1))()
            assert lib.wx_exchange_local_tiles(h, tiles, n) == n and list(tiles[:n]) == py.local
            assert lib.wx_exchange_needs_comm(h) == int(py.needs_comm)
            assert lib.wx_exchange_send_doubles(h) == py.send_buf.numel()
            assert lib.wx_exchange_recv_doubles(h) == py.recv_buf.numel()
            sc, rc = (ctypes.c_size_t * world)(), (ctypes.c_size_t * world)()
            _lib.check(lib.wx_exchange_peer_counts(h, sc, rc), "wx_exchange_peer_counts")
            assert list(sc) == py.send_splits and list(rc) == py.recv_splits
            # fake base addresses (never dereferenced on the host): the slots are offsets from them
            sbase, rbase = 1 << 40, 1 << 41
            _lib.check(lib.wx_exchange_bind(h, sbase, rbase), "wx_exchange_bind")
            for p in py.local:
                for e in range(4):
                    want_s = sbase + 8 * py._send_slot[(p, e)] * ec
                    kind, slot = py._halo_src[(p, e)]
                    want_h = (sbase if kind == "send" else rbase) + 8 * slot * ec
                    assert lib.wx_exchange_send_ptr(h, p, e) == want_s and lib.wx_exchange_halo_ptr(h, p, e) == want_h
            assert lib.wx_exchange_send_ptr(h, 6 * k * k, 0) is None       # not a tile of this rank / not a tile at all
            if py.needs_comm:   # no communicator: the exchange refuses to start, with a message
                assert lib.wx_exchange_start(h, None, None) == 1 and b"communicator" in lib.wx_last_error()
            else:
                assert lib.wx_exchange_start(h, None, None) == 0 and lib.wx_exchange_wait(h, None) == 0
        finally:
            lib.wx_exchange_destroy(h)
    del torch
