#!/usr/bin/env python3
"""One line per view of a lib_ab.py log: medians of A and B, their ratio, frames equal or not."""
import re, sys
for l in open(sys.argv[1]):
    m = re.search(r'^(.*?)(\S+) (\S+): A .*? med ([\d.]+) ms \| B .*? med ([\d.]+) ms \| B/A ([\d.]+) \| frames (\w+)', l)
    if m: print(*m.groups())
