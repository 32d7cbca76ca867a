#!/usr/bin/env python3
"""Registers / LDS / scratch of the kernels in a device-only assembly file (hipcc --cuda-device-only -S): development tool.
usage: kres.py file.s [substring ...]"""
import re
import sys

txt = open(sys.argv[1]).read()
keys = ("group_segment_fixed_size", "private_segment_fixed_size", "sgpr_count", "vgpr_count", "vgpr_spill_count", "agpr_count")
for blk in txt.split("  - .agpr_count:")[1:]:
    blk = ".agpr_count:" + blk
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if len(sys.argv) > 2 and not any(s in name for s in sys.argv[2:]):
        continue
    vals = {k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1)) for k in keys if re.search(r"\." + k + r":\s+(\d+)", blk)}
    print(f"{name[:110]:110s} vgpr {vals.get('vgpr_count'):4d} agpr {vals.get('agpr_count', 0):3d} spill {vals.get('vgpr_spill_count'):4d} "
          f"scratch {vals.get('private_segment_fixed_size'):5d} lds {vals.get('group_segment_fixed_size'):6d} sgpr {vals.get('sgpr_count')}")
