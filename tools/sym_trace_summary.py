#!/usr/bin/env python3
"""Summary of a DLCO_SYM_TRACE dump (skinny_sym_kernel<3, NS, true, TRACE>): where a chunk iteration's cycles go, per wave
role, averaged over the steady-state iterations of the sampled workgroups.  Stamps are s_memtime (shader clock)."""
import collections
import sys

rows = [l.split() for l in open(sys.argv[1]) if not l.startswith("#")]
d = collections.defaultdict(dict)
for wg, w, c, s0, s1, s2, s3 in rows:
    d[(int(wg), int(w))][int(c)] = tuple(int(x) for x in (s0, s1, s2, s3))
wgs = sorted({k[0] for k in d})
lo, hi = 6, 26
print("workgroups sampled:", wgs)
for wg in wgs:
    it = []
    for c in range(lo, hi):
        ends = [d[(wg, w)][c][3] for w in range(12) if c in d[(wg, w)] and d[(wg, w)][c][3]]
        prev = [d[(wg, w)][c - 1][3] for w in range(12) if c - 1 in d[(wg, w)] and d[(wg, w)][c - 1][3]]
        if ends and prev:
            it.append(max(ends) - max(prev))
    def avg(ws, a, b):
        v = [d[(wg, w)][c][b] - d[(wg, w)][c][a] for w in ws for c in range(lo, hi) if c in d[(wg, w)] and d[(wg, w)][c][a] and d[(wg, w)][c][b]]
        return sum(v) / max(len(v), 1)
    comp, load = range(0, 8), range(8, 12)
    last = collections.Counter()
    for c in range(lo, hi):
        arr = {w: d[(wg, w)][c][2] for w in range(12) if c in d[(wg, w)] and d[(wg, w)][c][2]}
        if arr:
            last["loader" if max(arr, key=arr.get) >= 8 else "compute"] += 1
    print("wg %3d: iteration %6.0f cyc | compute: fetch %5.0f  mfma %5.0f  barrier wait %5.0f | loader: wait+store %5.0f  issue %5.0f  barrier wait %5.0f | last at barrier: %s"
          % (wg, sum(it) / max(len(it), 1), avg(comp, 0, 1), avg(comp, 1, 2), avg(comp, 2, 3), avg(load, 0, 1), avg(load, 1, 2), avg(load, 2, 3), dict(last)))
