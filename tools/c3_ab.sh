#!/bin/bash
# C3 (sequential_line_search_nd 32 30): mean / median ms per submit without the first, three runs, for the library in place
cd "$(dirname "$0")/.."
for r in 1 2 3; do
  ./sequential-line-search_amd/bin/sequential_line_search_nd 32 30 1 | python3 -c "
import re, sys, statistics
ms = [float(v) for v in re.findall(r' ms ([-\d.e]+)', sys.stdin.read())]
print('submits', len(ms), 'mean w/o first %.4f ms' % statistics.mean(ms[1:]), 'median %.4f' % statistics.median(ms), 'last %.3f' % ms[-1])"
done
