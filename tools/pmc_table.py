#!/usr/bin/env python3
"""Collapse rocprofv3 --pmc CSVs (p_counter_collection.csv) into one row per (kernel, counter):
mean value per dispatch of the render kernels.  Usage: python tools/pmc_table.py DIR [DIR...]"""
import csv
import collections
import sys


def main(dirs):
    for d in dirs:
        acc = collections.defaultdict(list)
        with open(f"{d}/p_counter_collection.csv") as f:
            for row in csv.DictReader(f):
                name = row["Kernel_Name"]
                if "render_kernel" not in name:
                    continue
                acc[(name.split("(")[0][-40:], row["Counter_Name"], row["Dispatch_Id"])].append(float(row["Counter_Value"]))
        per = collections.defaultdict(list)
        for (k, c, disp), v in acc.items():
            per[(k, c)].append(sum(v))
        print(f"# {d}")
        for (k, c), v in sorted(per.items()):
            print(f"{k:42s} {c:26s} dispatches={len(v):2d} mean={sum(v) / len(v):.6g}")


if __name__ == "__main__":
    main(sys.argv[1:])
