#!/usr/bin/env python3
"""Condenses rocprofv3 output directories into the markdown tables kept under profiles/ — and into the one JSON record
`bench.py` reads its recorded (not live) figures from.

    python tools/prof_summary.py <title> <dir> [<dir> ...] > profiles/rNN_<what>_summary.md
    python tools/prof_summary.py --json profiles/rNN_pmc.json --head <git sha> --from-md profiles/rNN_solver_summary.md [more.md ...]

For every directory given: `*kernel_stats.csv` (from `--kernel-trace --stats`) becomes a table of the top kernels;
`*counter_collection.csv` (from a `--pmc` pass) becomes mean counter values per dispatch of each kernel.

`--json`: the dominant kernel of every section (`c2_stats`, `c2_pmc_FETCH_SIZE`, … — the section name's first token is the
bench workload) with its launch statistics and counter means, read back from the committed markdown summaries, so that every
recorded number `bench.py` prints greps to a file under profiles/.  `--head` is the commit the profiled build was made from
(the GPU box receives a snapshot without .git: the caller passes `git rev-parse --short HEAD`).
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name, n=90):
    return name if len(name) <= n else name[:n]


# the dominant kernel of a workload: the first of these that a section's kernel names contain
DOMINANT = {"c2": ["cilqr_solve_share_kernel", "cilqr_solve_kernel<false, 1, false", "cilqr_solve_kernel<false"],
            "c3": ["cilqr_solve_split_kernel", "cilqr_solve_kernel<false, 2, false", "cilqr_solve_kernel<false"],
            "c5": ["cilqr_solve_groups_fast", "cilqr_solve_kernel<false, 0, false"],
            "warp": ["warp_batch_kernel", "warp_kernel"], "warp16": ["warp_batch_kernel", "warp"],
            "occ": ["layer_to_occ_steps_kernel", "layer_to_occ"]}


def parse_md(path):
    """{section: {"kind": "stats"|"pmc", "rows": [...]}} from a summary this tool wrote."""
    sections, cur = {}, None
    for line in open(path):
        m = re.match(r"## `([^`]+)` — `rocprofv3 (--kernel-trace --stats|--pmc)`", line)
        if m:
            cur = {"kind": "stats" if "stats" in m.group(2) else "pmc", "rows": []}
            sections[m.group(1)] = cur
            continue
        if cur is None or not line.startswith("| ") or line.startswith("| kernel") or line.startswith("|---"):
            continue
        cells = [c.strip() for c in line.strip().strip("|").split("|")]
        cur["rows"].append(cells)
    return sections


def to_json(md_files, head):
    out = {"recorded_head": head, "sources": [os.path.relpath(f) for f in md_files],
           "units": {"FETCH_SIZE": "KB per dispatch (x2 on gfx950, MI355X_MICROARCH.md)", "WRITE_SIZE": "KB per dispatch",
                     "stats": "ns", "SQ_*/TCC_*": "events per dispatch"},
           "workloads": {}}
    for f in md_files:
        for name, sec in parse_md(f).items():
            wl = name.split("_")[0]
            pats = DOMINANT.get(wl)
            if not pats:
                continue
            rec = out["workloads"].setdefault(wl, {"kernel": None, "stats": None, "counters": {}, "source": os.path.relpath(f)})
            rows = None
            for p in pats:
                rows = [r for r in sec["rows"] if p in r[0]]
                if rows:
                    break
            if not rows:
                continue
            rec["kernel"] = rec["kernel"] or rows[0][0]
            if sec["kind"] == "stats":
                r = rows[0]
                rec["stats"] = {"calls": int(r[1]), "avg_ns": float(r[2]), "min_ns": float(r[3]), "max_ns": float(r[4])}
            else:
                for r in rows:
                    rec["counters"][r[1]] = float(r[2])
    for wl, rec in out["workloads"].items():
        c = rec["counters"]
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            rec["hbm_bytes_per_launch"] = 2.0 * c["FETCH_SIZE"] * 1e3 + c["WRITE_SIZE"] * 1e3
    return out


def main():
    if "--json" in sys.argv:
        a = sys.argv[1:]
        dst = a[a.index("--json") + 1]
        head = a[a.index("--head") + 1] if "--head" in a else "unknown"
        mds = a[a.index("--from-md") + 1:]
        json.dump(to_json(mds, head), open(dst, "w"), indent=1, sort_keys=True)
        open(dst, "a").write("\n")
        return
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
