#!/usr/bin/env python3
"""Condenses rocprofv3 output directories into the markdown tables kept under profiles/.

    python tools/prof_summary.py <title> <dir> [<dir> ...] > profiles/rNN_<what>_summary.md

For every directory given: `*kernel_stats.csv` (from `--kernel-trace --stats`) becomes a table of the top kernels;
`*counter_collection.csv` (from a `--pmc` pass) becomes mean counter values per dispatch of each kernel.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name, n=90):
    return name if len(name) <= n else name[:n]


def main():
    title, dirs = sys.argv[1], sys.argv[2:]
    print("# %s\n" % title)
    for d in dirs:
        stats = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True))
        pmc = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))
        if stats:
            print("## `%s` — `rocprofv3 --kernel-trace --stats`\n" % os.path.basename(d.rstrip("/")))
            print("| kernel | calls | avg ns | min ns | max ns | % |\n|---|---|---|---|---|---|")
            for f in stats:
                rows = list(csv.DictReader(open(f)))
                for r in rows[:8]:
                    print("| %s | %s | %s | %s | %s | %s |" % (short(r["Name"]), r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"],
                                                            r["Percentage"]))
            print()
        if pmc:
            acc = defaultdict(lambda: [0.0, 0])
            for f in pmc:
                for r in csv.DictReader(open(f)):
                    k = (short(r["Kernel_Name"], 110), r["Counter_Name"])
                    acc[k][0] += float(r["Counter_Value"])
                    acc[k][1] += 1
            print("## `%s` — `rocprofv3 --pmc` (mean per dispatch)\n" % os.path.basename(d.rstrip("/")))
            print("| kernel | counter | mean per dispatch | dispatches |\n|---|---|---|---|")
            for (k, c), (tot, n) in sorted(acc.items()):
                print("| %s | %s | %.6g | %d |" % (k, c, tot / n, n))
            print()


if __name__ == "__main__":
    main()
