#!/usr/bin/env python3
"""Which host API calls were in flight while the GPU sat idle between two queued kernels?
    python benchmarks/stall_probe.py <rocprofv3 --kernel-trace --hip-trace results.db> [min_gap_ms]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    min_gap = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 5e6
    views = [r[0] for r in db.execute("select name from sqlite_master where type='view'")]
    print("views:", views)
    k = list(db.execute("select name,start,end from kernels order by start"))
    marks = [i for i, r in enumerate(k) if "bbox_feat" in r[0]]
    first = k[marks[3]][1] if len(marks) > 3 else k[0][1]          # skip the warm-up steps
    cols = [d[1] for d in db.execute("pragma table_info(regions)")]
    print("regions columns:", cols)
    for a, b in zip(k[:-1], k[1:]):
        if a[2] < first or b[1] - a[2] < min_gap:
            continue
        print("GPU idle %.1f ms after %s before %s" % ((b[1] - a[2]) / 1e6, a[0][:50], b[0][:50]))
        q = "select name,start,end,tid from regions where end > ? and start < ? and (end - start) > 200000 order by start"
        for r in db.execute(q, (a[2], b[1])):
            print("     %-40s %9.2f ms  (starts %+.2f ms into the gap) tid %s" % (r[0][:40], (r[2] - r[1]) / 1e6, (r[1] - a[2]) / 1e6, r[3]))


if __name__ == "__main__":
    main()
