#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output: mean counter value per kernel over its
dispatches.  python tools/pmc_summary.py <dir with *_counter_collection.csv> [out.json]"""
import csv
import glob
import json
import os
import sys


def main():
    root = sys.argv[1]
    acc = {}
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"]
                c = row["Counter_Name"]
                v = float(row["Counter_Value"])
                d = acc.setdefault(k, {}).setdefault(c, [0.0, 0])
                d[0] += v
                d[1] += 1
    out = {k: {c: {"mean": s / n, "dispatches": n} for c, (s, n) in cs.items()} for k, cs in acc.items()}
    text = json.dumps(out, indent=1)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            f.write(text)
    else:
        print(text)


if __name__ == "__main__":
    main()
