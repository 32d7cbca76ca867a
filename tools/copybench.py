#!/usr/bin/env python3
"""Development benchmark: the streaming copy behind bench.py's roofline.ceiling in every launch shape (loads in flight per
lane, temporal / non-temporal, workgroups per CU), 2 GiB, far beyond the Infinity Cache.  Prints GB/s (read + written)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from wxfactory_amd import _lib  # noqa: E402

lib = _lib.load()
fn = lib.wx_stream_copy_variant
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = "cuda:0"
nbytes = 2 << 30
src = torch.empty(nbytes // 8, dtype=torch.float64, device=dev).normal_()
dst = torch.empty_like(src)
st = torch.cuda.current_stream(dev).cuda_stream
best = (0.0, None)
for unroll in (1, 2, 4, 8):
    for nt in (0, 1):
        for wg in (2, 4, 8, 16, 32):
            ts = []
            for it in range(7):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                _lib.check(fn(src.data_ptr(), dst.data_ptr(), nbytes, unroll, nt, wg, st), "copy")
                b.record()
                torch.cuda.synchronize()
                if it >= 2:
                    ts.append(a.elapsed_time(b) * 1e-3)
            gbs = 2 * nbytes / (sum(ts) / len(ts)) / 1e9
            assert torch.equal(src[:4096], dst[:4096]) and torch.equal(src[-4096:], dst[-4096:])
            best = max(best, (gbs, (unroll, nt, wg)))
            print(f"unroll {unroll} nt {nt} wg/CU {wg:2d}: {gbs:7.1f} GB/s", flush=True)
# torch's own device-to-device copy, for scale
ts = []
for it in range(7):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    dst.copy_(src)
    b.record()
    torch.cuda.synchronize()
    if it >= 2:
        ts.append(a.elapsed_time(b) * 1e-3)
print(f"torch Tensor.copy_: {2 * nbytes / (sum(ts) / len(ts)) / 1e9:7.1f} GB/s")
print("best", best)
