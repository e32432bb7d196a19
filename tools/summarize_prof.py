#!/usr/bin/env python3
"""Condense rocprofv3 (rocpd sqlite) output dirs — one kernel-trace/stats run plus PMC passes — into a
small text summary to commit under profiles/.
usage: summarize_prof.py <dir> [pmc.json] [--bench-line FILE] [--require-phases]
  the bench line of the profiled command is looked for in FILE, <dir>/bench.json, <dir>/stats.log (first found);
  --require-phases: exit 2 when the launches of the traversal kernel cannot be split by bench.py phase
  (no bench line, or a launch count the line does not explain) — a round's summaries must not lose that section silently."""
import glob
import os
import sqlite3
import sys
from collections import defaultdict

_args = sys.argv[1:]
REQUIRE_PHASES = "--require-phases" in _args
_args = [a for a in _args if a != "--require-phases"]
BENCH_LINE_FILE = None
if "--bench-line" in _args:
    i = _args.index("--bench-line")
    BENCH_LINE_FILE = _args[i + 1]
    del _args[i:i + 2]
sys.argv = [sys.argv[0]] + _args
out = sys.argv[1]
phases_done = False
print(f"# rocprofv3 summary of {os.path.basename(out.rstrip('/'))}")
for f in sorted(glob.glob(os.path.join(out, "stats", "**", "*.db"), recursive=True)):
    db = sqlite3.connect(f)
    print(f"\n## kernel stats (--kernel-trace --stats)  [{os.path.relpath(f, out)}]")
    print(f"{'kernel':70s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'%':>7s}")
    for name, calls, total, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print(f"{name[:70]:70s} {calls:6d} {total:12.1f} {avg:10.2f} {pct:7.2f}")
    rows = list(db.execute("select name, min(duration), max(duration), avg(duration), count(*), vgpr_count, accum_vgpr_count, sgpr_count, "
                           "scratch_size, lds_size, grid_x, workgroup_x from kernels where name like '%vrt_%' group by name"))
    for r in rows:
        print(f"   {r[0][:60]}: n={r[4]} min={r[1] / 1e3:.1f}us max={r[2] / 1e3:.1f}us avg={r[3] / 1e3:.1f}us vgpr={r[5]} agpr={r[6]} sgpr={r[7]} "
              f"scratch={r[8]} lds={r[9]} grid={r[10]} wg={r[11]}")
# The launches of the traversal kernel by bench.py phase.  bench.py's launch sequence of the PRODUCT kernel is fixed:
# 2 counting contexts x 5 views x 1 frame | 2 probe frames, warm-up, timed region | (single-stream leg) warm-up, timed
# region | per view: settle, `reps` back to back (-> roofline.kernel_ms_per_view), min(reps, 512) with an event pair each.
import json as _json
line = None
for log in [BENCH_LINE_FILE, os.path.join(out, "bench.json"), os.path.join(out, "stats.log")]:
    if not log or not os.path.exists(log) or line is not None:
        continue
    for ln in open(log, errors="replace"):
        if ln.startswith("{") and '"roofline"' in ln:
            line = _json.loads(ln)
if line is None:
    print("\n## NO bench line found (looked in --bench-line, bench.json, stats.log): launches NOT split by bench.py phase")
if line is not None:
    for f in sorted(glob.glob(os.path.join(out, "stats", "**", "*.db"), recursive=True)):
        if not line or not line.get("roofline"):
            break
        db = sqlite3.connect(f)
        d = [r[0] / 1e3 for r in db.execute("select duration from kernels where name like '%vrt_trace_kernel<%, false,%' or name like '%vrt_path_kernel<%' or name like '%vrt_pool_kernel<%' order by start")]
        steps, warmup = line["steps"], line["warmup"]
        pre = line.get("precondition_frames", 0)
        views = line["config"]["views"] + line["config"].get("views_reported_only", [])
        settle = line["roofline"]["settle_frames"]
        reps = line["roofline"].get("timed_frames_per_view") or max(8, steps // len(line["config"]["views"]))
        timed = min(reps, 512)
        # bench.py's launch sequence of the PRODUCT kernel (round 4): 2 counting contexts x views x 1 frame | 2 probe frames, pre-conditioning
        # (0 by default), warm-up, timed region | `value_sustained`: ~150 ms of frames, warm-up, timed region | (single-stream leg) warm-up,
        # timed region | per view: settle, `reps` back to back, min(reps, 512) timed | present pass: 64 frames
        sus = (line.get("value_sustained") or {}).get("precondition_frames")
        sus_n = (sus + warmup + steps) if sus is not None else 0
        tail = 64 if line.get("present_pass") and "error" not in line["present_pass"] else 0
        head = 2 * len(views) + 2 + pre + (warmup + steps) + sus_n + (warmup + steps)
        want = head + len(views) * (settle + reps + timed) + tail
        print(f"\n## {len(d)} launches of the traversal kernel by bench.py phase (us)")
        if len(d) != want:
            print(f"   (expected {want} launches from the bench line = {2 * len(views)} + 2 + {pre} + {warmup + steps} + {sus_n} + {warmup + steps} + "
                  f"{len(views)} x ({settle} + {reps} + {timed}) + {tail}; phase breakdown skipped; durations -> launch_durations.json)")
            with open(os.path.join(out, "launch_durations.json"), "w") as fh:
                _json.dump({"want": want, "got": len(d), "durations_us": [round(x, 1) for x in d]}, fh)
            continue
        phases_done = True
        a = 2 * len(views) + 2 + pre
        main = d[a + warmup:a + warmup + steps]
        single = d[a + warmup + steps + sus_n + warmup:a + sus_n + 2 * (warmup + steps)]
        two = "1 frame(s) in flight" not in line["config"]["parallelism"]
        print(f"   timed region ({steps} launches): avg {sum(main) / steps:.2f}"
              + ("  (two frames in flight: two kernels share the GPU, a kernel's duration is not a frame's cost)" if two else ""))
        print(f"   single-stream timed region ({steps} launches): avg {sum(single) / steps:.2f}   "
              f"(bench line ms_per_step_single_stream, wall clock incl. launch gaps: {line['ms_per_step_single_stream'] * 1e3:.2f})")
        leg = d[head:len(d) - tail]
        per = settle + reps + timed
        tot = []
        for v, name in enumerate(views):
            seg = leg[v * per:(v + 1) * per]
            b2b = seg[settle:settle + reps]
            ev = sorted(seg[settle + reps:])
            print(f"   {name}: {reps} back-to-back launches avg {sum(b2b) / reps:.2f} (HIP events: {line['roofline']['kernel_ms_per_view'][name] * 1e3:.2f}); "
                  f"{timed} individually timed launches median {ev[len(ev) // 2]:.2f} "
                  f"(HIP events: {line['roofline']['frame_ms_percentiles_per_view'][name]['median'] * 1e3:.2f})")
            if name in line["config"]["views"]:
                tot += b2b
        print(f"   roofline leg, views {'/'.join(line['config']['views'])}: avg {sum(tot) / len(tot):.2f}  "
              f"(bench line roofline.kernel_ms_avg by HIP events incl. the schedule kernels between the launches: {line['roofline']['kernel_ms_avg'] * 1e3:.2f})")
if REQUIRE_PHASES and not phases_done:
    print("\n## ERROR: --require-phases and no phase split was printed", flush=True)
    sys.exit(2)
for d in sorted(glob.glob(os.path.join(out, "pmc*"))):
    if not os.path.isdir(d):
        continue
    for f in sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True)):
        db = sqlite3.connect(f)
        agg = defaultdict(lambda: defaultdict(float))
        disp = defaultdict(set)
        for k, c, v, did in db.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection where kernel_name like '%vrt_%'"):
            agg[k][c] += v
            disp[k].add(did)
        print(f"\n## PMC pass [{os.path.relpath(f, out)}]")
        for k, cs in sorted(agg.items()):
            n = max(1, len(disp[k]))
            print(f"kernel {k}  dispatches={n}")
            for c, v in sorted(cs.items()):
                print(f"   {c:34s} per_dispatch={v / n:.6g}")

# machine-readable per-kernel counters (bench.py reads the HBM traffic of the traversal kernel from here)
if len(sys.argv) > 2:
    import json
    result = {}
    for d in sorted(glob.glob(os.path.join(out, "pmc*"))):
        for f in sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True)):
            db = sqlite3.connect(f)
            for k, c, v, n in db.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                                         "where kernel_name like '%vrt_%' group by kernel_name, counter_name"):
                result.setdefault(k, {})[c] = v
    for f in sorted(glob.glob(os.path.join(out, "stats", "**", "*.db"), recursive=True)):
        db = sqlite3.connect(f)
        for name, calls, total, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            if "vrt_" in name:
                result.setdefault(name, {})["kernel_trace_avg_us"] = avg
                result[name]["kernel_trace_calls"] = calls
    with open(sys.argv[2], "w") as fh:
        json.dump(result, fh, indent=1, sort_keys=True)
