#!/usr/bin/env python3
"""Summary of a DLCO_JACOBI_TRACE dump (jacobi_blk_kernel<3, TR>): cycles of the four phases of an inner rotation step of
one wave - top -> partner column in -> dot product reduced -> rotation and updates done -> next top (store, loop)."""
import sys

rows = [list(map(int, l.split())) for l in open(sys.argv[1]) if not l.startswith("#")]
print(open(sys.argv[1]).readline().strip())
ph = {"partner column from LDS": [r[1] - r[0] for r in rows], "dot product + 8-lane reduction": [r[2] - r[1] for r in rows],
      "rotation + column updates": [r[3] - r[2] for r in rows], "store, bookkeeping, loop (to the next top)": [rows[i + 1][0] - rows[i][3] for i in range(len(rows) - 1)],
      "whole step": [rows[i + 1][0] - rows[i][0] for i in range(len(rows) - 1)]}
print("%d inner steps (8 per block meeting; the 8th carries the round's end: store of the wave's own columns, workgroup barrier)" % len(rows))
for k, v in ph.items():
    w = sorted(v)
    print("%-46s median %5d   p10 %5d   p90 %5d   mean %6.0f cycles" % (k, w[len(w) // 2], w[len(w) // 10], w[9 * len(w) // 10], sum(v) / len(v)))
tot = ph["whole step"]
print("median step by position in the meeting:", " ".join(str(sorted(tot[i] for i in range(len(tot)) if i % 8 == kk)[len(tot) // 16]) for kk in range(8)))
