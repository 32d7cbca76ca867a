#!/usr/bin/env python3
"""Average every counter of the rocprofv3 --pmc passes under <dir> per kernel and launch (development tool).

    python tools/sq_summary.py <dir> <out.json> [note]
"""
import collections
import csv
import glob
import json
import sys


def main():
    d, out = sys.argv[1], sys.argv[2]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "wx::" not in r["Kernel_Name"]:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for k, cs in sorted(acc.items()):
        res[k] = {c: sum(v) / len(v) for c, v in sorted(cs.items())}
        res[k]["launches"] = max(len(v) for v in cs.values())
    if len(sys.argv) > 3:
        res["_note"] = sys.argv[3]
    json.dump(res, open(out, "w"), indent=1)
    for k, cs in res.items():
        if k.startswith("_"):
            continue
        print(k)
        for c, v in cs.items():
            print(f"   {c:34s} {v:16.1f}")


if __name__ == "__main__":
    main()
