#!/usr/bin/env python3
"""Timeline of the launches of bench.py's first timed region from a rocprofv3 --kernel-trace database (where do the
microseconds of a 20-step region go?).  usage: short_trace.py <dir with *.db> <precondition> <warmup> <steps>"""
import glob, os, sqlite3, sys
d, pre, warm, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
for f in sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True)):
    db = sqlite3.connect(f)
    rows = list(db.execute("select start, end, name from kernels where name like '%vrt_trace_kernel<%, false,%' order by start"))
    a = 10 + 2 + pre + warm          # counting contexts (2 x 5 views), 2 probe frames, pre-conditioning, warm-up
    reg = rows[a:a + steps]
    prev = rows[a - 1]
    t0 = reg[0][0]
    print(f"{len(rows)} product launches; timed region = launches {a}..{a + steps - 1}; gap after the last warm-up launch: {(t0 - prev[1]) / 1e3:.1f} us")
    for i, (s, e, n) in enumerate(reg):
        print(f"  {i:2d}: start {(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f} us  end {(e - t0) / 1e3:8.1f}")
    print(f"region: first start -> last end {(max(r[1] for r in reg) - t0) / 1e3:.1f} us = {(max(r[1] for r in reg) - t0) / 1e3 / steps:.2f} us per step")
