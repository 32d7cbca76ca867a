#!/usr/bin/env python3
"""Does data a kernel has just WRITTEN come back faster than from HBM when the next kernel reads it, and how much streaming
traffic in between does that survive (the 256 MB memory-side cache of the MI355X)?  Development tool behind DESIGN section 7's note
on interleaving the extrapolation kernel's slabs with the fused kernel's: read rate of a buffer (wx_stream_read) cold, right after
it was written, and after X MB of other data streamed through."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from wxfactory_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda", 0)
big = torch.empty(3 << 27, dtype=torch.float64, device=dev).normal_()   # 3 GiB
sink = torch.zeros(lib.wx_stream_read_sink_doubles(), dtype=torch.float64, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream


def read(t):
    _lib.check(lib.wx_stream_read(t.data_ptr(), t.numel() * 8, sink.data_ptr(), st), "wx_stream_read")


def timed_read(buf, prepare, reps=15):
    ts = []
    for _ in range(reps):
        prepare()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        read(buf)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e-3)
    ts.sort()
    return ts[len(ts) // 2]


for mb in (14, 55, 110, 220):
    n = mb * (1 << 20) // 8
    buf = torch.empty(n, dtype=torch.float64, device=dev)
    src = torch.randn(n, dtype=torch.float64, device=dev)
    cold = timed_read(buf, lambda: read(big))
    line = [f"{mb:4d} MB buffer: cold {mb * 1.048576e-3 / cold:7.1f}"]
    hot = timed_read(buf, lambda: (read(big), buf.copy_(src)))
    line.append(f"just written {mb * 1.048576e-3 / hot:7.1f}")
    reread = timed_read(buf, lambda: (read(big), read(buf)))
    line.append(f"just read {mb * 1.048576e-3 / reread:7.1f}")
    for x in (64, 128, 192, 256, 512):
        part = big[: x * (1 << 20) // 8]
        t = timed_read(buf, lambda: (read(big), buf.copy_(src), read(part)))
        line.append(f"written, then {x} MB streamed {mb * 1.048576e-3 / t:7.1f}")
    print("; ".join(line) + "   (GB/s)", flush=True)
