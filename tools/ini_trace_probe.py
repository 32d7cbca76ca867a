#!/usr/bin/env python3
"""The shipped .ini sizes' steps (EPI2 + KIOPS, EPI2 + PMEX, Ros2 + FGMRES) for a kernel trace (development tool):
rocprofv3 --kernel-trace -- python3 tools/ini_trace_probe.py, then tools/trace_gaps.py on the trace."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from benchlib.extras import ini_size_extras  # noqa: E402

print(ini_size_extras(torch.device("cuda", 0), 1), flush=True)
