"""CPU restatement of the device matching of the BIG-C training step (TEST INFRASTRUCTURE ONLY: imported by tests/, never by
the product).

The reference matches queries to GT predicates with ``scipy.optimize.linear_sum_assignment`` on the host
(models/model_0v10.py:637) after re-indexing the GT adjacency by proposals on the host (``enti_viou_align``,
models/model_0v10.py:583-602).  scipy (third-party, not in /root/reference; this image: see ``scipy.__version__``, any 1.4+) solves
the rectangular assignment problem with the shortest-augmenting-path algorithm of D. F. Crouse, "On implementing 2D rectangular
assignment algorithms", IEEE Trans. Aerospace and Electronic Systems 52(4), 2016, in float64, on the transposed matrix when there
are more rows than columns.  ``csrc/loss.hip: hungarian_kernel`` restates that algorithm for one wave per video; this module is the
same restatement in plain Python, written the way the KERNEL is written (the candidate choice as an order-free rule over all
not-yet-visited columns instead of scipy's sequential scan), so that the rule itself can be pinned against scipy on the CPU:
tests/test_host_cpu.py compares ``assign`` with scipy on random, tie-heavy integer, duplicated-column and partly infinite matrices.

    assign(cost (nq, n_gt) array, n_gt < nq) -> match (nq,) int: GT index per query or -1       (raises ValueError like scipy)
    align(vi (n, t) float32, adj (2, g, t), th) -> (2, g, n)                                       (the loop form of adj_from_viou)
"""
import numpy as np


def _better(a, b):
    """candidate a = (value, column unassigned, position in `remaining`) wins over b.  scipy scans positions 0 .. n-1 keeping the
    first lowest value, except that a later UNASSIGNED column with an equal value replaces it: over the whole scan that is the
    lowest value; among equals an unassigned column before an assigned one; the LAST unassigned one; the FIRST assigned one."""
    if a[0] != b[0]:
        return a[0] < b[0]
    if a[1] != b[1]:
        return a[1] > b[1]
    return a[2] > b[2] if a[1] else a[2] < b[2]


def assign(cost):
    cost = np.asarray(cost, dtype=np.float64)
    nc, nr = cost.shape                       # columns = queries, rows = GT predicates (scipy transposes the tall matrix)
    assert nr < nc
    if np.isnan(cost).any() or np.isneginf(cost).any():
        raise ValueError("matrix contains invalid numeric entries")
    c = cost.T
    u, v = np.zeros(nr), np.zeros(nc)
    row4col, col4row = -np.ones(nc, dtype=np.int64), -np.ones(nr, dtype=np.int64)
    path = -np.ones(nc, dtype=np.int64)
    for cur in range(nr):
        min_val, i, sink = 0.0, cur, -1
        remaining = [nc - it - 1 for it in range(nc)]
        sr, sc = np.zeros(nr, dtype=bool), np.zeros(nc, dtype=bool)
        sp = np.full(nc, np.inf)
        while sink == -1:
            sr[i] = True
            best = None
            for it, j in enumerate(remaining):                  # (any order: `_better` does not depend on it)
                r = min_val + c[i, j] - u[i] - v[j]
                if r < sp[j]:
                    path[j], sp[j] = i, r
                cand = (sp[j], row4col[j] == -1, it)
                if best is None:
                    if cand[0] < np.inf or cand[1]:
                        best = cand
                elif _better(cand, best):
                    best = cand
            if best is None or not best[0] < np.inf:
                raise ValueError("cost matrix is infeasible")
            min_val = best[0]
            j = remaining[best[2]]
            if row4col[j] == -1:
                sink = j
            else:
                i = row4col[j]
            sc[j] = True
            remaining[best[2]] = remaining[-1]
            remaining.pop()
        u[cur] += min_val
        for r in range(nr):
            if sr[r] and r != cur:
                u[r] += min_val - sp[col4row[r]]
        for j in range(nc):
            if sc[j]:
                v[j] -= min_val - sp[j]
        j = sink
        while True:
            r = path[j]
            row4col[j] = r
            col4row[r], j = j, col4row[r]
            if r == cur:
                break
    return row4col


def align(vi, adj, th):
    """models/model_0v10.py:583-602 with explicit loops (first maximum on ties, as torch.argmax)"""
    vi = np.asarray(vi, dtype=np.float32)
    adj = np.asarray(adj, dtype=np.float32)
    n, t = vi.shape
    has = np.zeros(n, dtype=bool)
    pick = np.zeros(n, dtype=np.int64)
    for p in range(n):
        pick[p] = int(np.argmax(vi[p]))
        has[p] = bool((vi[p] > th).any())
    for g in range(t):
        if not (vi[:, g] > th).any():
            has[int(np.argmax(vi[:, g]))] = True
    return adj[:, :, pick] * has[None, None, :].astype(np.float32)
