#!/usr/bin/env python3
"""GPU idle time inside one step, from a rocprofv3 --kernel-trace database (rocpd sqlite):
    python benchmarks/step_gaps.py gpurun_out/train_prof/tr_results.db [marker-kernel-substring]
Steps are delimited by launches of the marker kernel (default: bbox_feat, the first kernel of a BIG-C forward)."""
import collections
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    marker = sys.argv[2] if len(sys.argv) > 2 else "bbox_feat"
    rows = list(db.execute("select name,start,end from kernels order by start"))
    starts = [i for i, r in enumerate(rows) if marker in r[0]]
    i0, i1 = starts[-3], starts[-2]
    seg = rows[i0:i1]
    wall = (seg[-1][2] - seg[0][1]) / 1e6
    busy = sum(r[2] - r[1] for r in seg) / 1e6
    print("step: wall %.2f ms, kernels %.2f ms (%.0f%%), %d launches" % (wall, busy, 100 * busy / wall, len(seg)))
    per = collections.defaultdict(lambda: [0, 0])
    for r in seg:
        per[r[0][:70]][0] += 1
        per[r[0][:70]][1] += r[2] - r[1]
    for k, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:18]:
        print("  %-72s %4d %8.3f ms" % (k, n, t / 1e6))
    big = []
    for a, b in zip(seg[:-1], seg[1:]):
        g = b[1] - a[2]
        if g > 50000:
            big.append((g / 1e3, a[0][:50], b[0][:50]))
    print("idle gaps > 50 us:")
    for g in sorted(big, reverse=True)[:16]:
        print("  %8.1f us  after %-52s before %s" % g)


if __name__ == "__main__":
    main()
