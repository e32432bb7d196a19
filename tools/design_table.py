#!/usr/bin/env python3
"""DESIGN.md 6's table rows, the driver-protocol sentence and the fly-through rows from the committed files profiles/<tag>_bench_*.json,
<tag>_flythrough_*.json (so that the document's numbers are the files' numbers).   usage: design_table.py [tag]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
ROWS = [("cfg0_256x256_64c_b4", "cfg0 256², 64³ b4, primary"), ("cfg1_1080p_256c_b4", "cfg1 1080p, 256³ b4, primary"),
        ("cfg2_1080p_512c_b8", "**cfg2 1080p, 512³ b8, primary + shadow (headline)**"), ("cfg2_1080p_512c_b4", "cfg2 with 4³ bricks"),
        ("cfg3_4k_1024c_b8", "cfg3 4K, 1024³ b8, 2 spp × (primary + shadow)"), ("cfg4_4k_2048c_b8_sparse", "cfg4 4K, 2048³ sparse, 16 spp, 3 bounces"),
        ("refapp_1024x576_128x64x128_b4", "the reference app's own run: 1024×576, 128×64×128 bricks of 4³, 2 spp, 2 bounces, sun")]


def load(name):
    with open(os.path.join(ROOT, "profiles", f"{tag}_{name}.json")) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def sig(v, n=3):
    return f"{v:.{n}g}" if v < 1 else (f"{v:.3f}" if v < 100 else f"{v:.1f}")


for name, label in ROWS:
    d = load("bench_" + name)
    r = d["roofline"]
    kv = r["kernel_ms_per_view"]
    traffic = r.get("traffic")
    tr = "n/a" if not traffic else (f"{traffic / 1e9:.0f} GB" if traffic > 5e9 else f"{traffic / 1e6:.1f} MB")
    frac = f"{r['frac']:.2f}" + ("" if r.get("frac_model_valid", True) else f" (model void: `bound` = issue, {r['governing_frac']:.2f})")
    pp = d.get("present_pass") or {}
    cpu = d["cpu_baseline"]
    cpus = f"{cpu['value']:.1f} ({cpu['kind']}, {cpu['cores']} thr)"
    port = d.get("cpu_baseline_port")
    if port and cpu["kind"] == "reference":
        cpus += f" / {port['value']:.1f} (port, {port['cores']})"
    print(f"| {label} | {d['value'] / 1e3:.2f} | {d['ms_per_step']:.4f} / {d['ms_per_step_single_stream']:.4f} | "
          + " / ".join(sig(kv[v]) for v in ("V0", "V1", "V2", "V1x", "VG") if v in kv)
          + f" | {frac} | {r['frac_issued']:.2f} | {r.get('lane_util', 0):.2f} | {r['valu_frac']:.2f} | {r['issue_slots_frac']:.2f} | {tr} | "
          + (f"{pp['us_median']:.0f} µs (issue-bound: byte model void)" if pp and not pp.get("frac_model_valid", True) else (f"{pp['us_median']:.0f} µs (frac {pp['frac']:.2f})" if pp else "n/a"))
          + f" | {cpus} | `{r['kernel']}` |")
d = load("bench_driver_protocol_steps20")
print(f"\ndriver protocol: {d['value'] / 1e3:.1f} Grays/s by the host's clock, {d['value_device_events'] / 1e3:.1f} by device events, "
      f"{d['value_sustained']['value'] / 1e3:.1f} sustained; ms_per_step {d['ms_per_step']:.4f}")
for name in ("flythrough_refapp", "flythrough_headline"):
    f = load(name)
    t, p, fr = f["trace"], f["present"], f["frame_trace_plus_present"]
    print(f"{name}: trace {t['min_ms']:.3f} / {t['max_ms']:.3f} / {t['avg_ms']:.3f} | present {p['avg_ms']:.3f} | frame {fr['min_ms']:.3f} / {fr['max_ms']:.3f} / "
          f"{fr['avg_ms']:.3f} | {f['frames']} frames, {f['Mrays_per_s_trace'] / 1e3:.2f} Grays/s over the path"
          + (f" | pipelined (two frames in flight, host clock): {f['pipelined_avg_frame_ms']}" if f.get("pipelined_avg_frame_ms") else ""))
