#!/bin/bash
# usage: vmcnt_pattern.sh file.s mangled-substring
# The sequence of s_waitcnt vmcnt(N) / s_barrier / loop headers of one kernel in a device assembly file: a store in a
# software-pipelined loop that waits with vmcnt(0) while younger prefetches are in flight shows up here at a glance.
awk -v k="$2" 'index($0, k) && /^_Z.*:/ {f=1} f{print} f&&/^\.Lfunc_end/{f=0}' "$1" | grep "s_waitcnt vmcnt\|s_barrier\|Loop Header" | sed 's/^\s*//; s/s_waitcnt //; s/;.*Loop Header.*/LOOP{/; s/s_barrier/BAR/' | tr '\n' ' '; echo
