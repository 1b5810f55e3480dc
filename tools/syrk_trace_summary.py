"""Summarises a DLCO_SYRK_TRACE file (phase time stamps of one SYRK launch): mean phase lengths per workgroup,
the timeline of one CU, and how many workgroups are in which phase over the launch."""
import collections, sys
import numpy as np
rows = [l.split() for l in open(sys.argv[1]) if not l.startswith('#')]
a = np.array(rows, dtype=np.int64)
wg, xcc, hw, t0, t1, t2, t3, tile = a.T
base = t0.min()
t0, t1, t2, t3 = [(x - base) / 100.0 for x in (t0, t1, t2, t3)]          # us
print("launch span %.1f us, %d workgroups" % (t3.max(), len(wg)))
print("mean per workgroup: start->first MFMA %.1f us, K loop %.1f us, epilogue %.1f us, total %.1f us"
      % ((t1 - t0).mean(), (t2 - t1).mean(), (t3 - t2).mean(), (t3 - t0).mean()))
cuid = (xcc << 12) | ((hw >> 8) & 0xff) | (((hw >> 13) & 7) << 8)
c0 = collections.Counter(cuid.tolist()).most_common(1)[0][0]
idx = np.nonzero(cuid == c0)[0]
idx = idx[np.argsort(t0[idx])]
print("one CU (%d workgroups):" % len(idx))
for i in idx:
    print("  wg %4d  start %6.1f  K loop %6.1f .. %6.1f  end %6.1f" % (wg[i], t0[i], t1[i], t2[i], t3[i]))
for t in np.linspace(0, t3.max(), 12):
    print("t = %5.0f us: %3d before their K loop, %3d in it, %3d in the epilogue"
          % (t, ((t0 <= t) & (t < t1)).sum(), ((t1 <= t) & (t < t2)).sum(), ((t2 <= t) & (t < t3)).sum()))
