#!/usr/bin/env python3
"""Registers, scratch and LDS of every kernel in a device assembly file (hipcc --cuda-device-only -S): what to look at
after touching a kernel - a private segment > 0 means spills or a dynamically indexed register array."""
import re
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("- .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\.%s:\s*(\S+)" % k, blk) or [None, "?"])[1]
    name = g("name")
    if pat and pat not in name:
        continue
    print("%-100s vgpr %s agpr %s scratch %s spill %s lds %s" % (name[:100], g("vgpr_count"), blk.split()[0], g("private_segment_fixed_size"),
                                                                   g("vgpr_spill_count"), g("group_segment_fixed_size")))
