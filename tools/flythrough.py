#!/usr/bin/env python3
"""The reference's own benchmark protocol on the HIP path: the scripted 60 s fly-through of
src/modules/voxel_rt/Benchmark.zig:141-172 replayed at a fixed simulated frame rate; reports the
min / max / avg frame time like Benchmark.Report.print (Benchmark.zig:109-136), from HIP events.
A frame of the app is the trace (ComputePipeline.zig:417-463) followed by the present / denoise pass at the
window resolution (image.frag:18-78, GraphicsPipeline.zig:27-39; Pipeline.draw, Pipeline.zig:432-541): both
are timed, per frame, and reported apart and summed.

    python tools/flythrough.py [workload] [simulated_fps] [present_w present_h] [--out profiles/x.json]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zig_vulkan_amd import workloads as W  # noqa: E402

argv = [a for a in sys.argv[1:]]
out_path = None
if "--out" in argv:
    i = argv.index("--out")
    out_path = argv[i + 1]
    del argv[i:i + 2]
name = argv[0] if len(argv) > 0 else W.HEADLINE
fps = float(argv[1]) if len(argv) > 1 else 30.0
w = W.WORKLOADS[name]
# the reference presents its 1024x576 internal image in a 1920x1080 window (src/main.zig:23,58-75); other workloads present 1:1
pw = int(argv[2]) if len(argv) > 3 else (1920 if w.width == 1024 else w.width)
ph = int(argv[3]) if len(argv) > 3 else (1080 if w.height == 576 else w.height)
grid = W.build_grid(w)


def stats(v):
    return {"min_ms": min(v), "max_ms": max(v), "avg_ms": sum(v) / len(v)}


# pass 1 (counting build, slow): rays per frame along the path
rt = W.make_renderer(w, grid, enable_counters=True)
bench = rt.create_benchmark()
rays, done = 0, False
while not done:
    rt.draw()
    rays += rt.counters()["rays"]
    done = bench.update(1.0 / fps)
rt.deinit()
# pass 2: the same path on the product kernels, trace + present per frame, HIP events around each
rt = W.make_renderer(w, grid)
bench = rt.create_benchmark()
trace, present, done = [], [], False
while not done:
    rt.draw()
    rt.present(pw, ph)
    trace.append(rt.last_kernel_ms())
    present.append(rt.last_denoise_ms())
    done = bench.update(1.0 / fps)
kernel = rt.kernel_name()
rt.deinit()
# pass 3 (round 6): the same path PIPELINED, as a renderer runs — two frames in flight, nothing waited for until the end; host clock over
# the whole path.  The present pass on the stream of the frame it reads (the default: the next frame traces on the other stream underneath
# it), on a stream of its own behind an event of that frame (VRT_TUNE_PRESENT_OWN_STREAM, the reference's two queues joined by a semaphore,
# Pipeline.zig:494-517), and the trace alone (no present pass) for scale.
import time  # noqa: E402
from zig_vulkan_amd import _lib as L  # noqa: E402


def pipelined(flags, with_present=True):
    r = W.make_renderer(w, grid, frames_in_flight=2, tuning_flags=flags)
    b = r.create_benchmark()
    for _ in range(8):       # (code objects, buffers, clocks)
        r.draw()
        if with_present:
            r.present(pw, ph)
    r.wait()
    n, fin = 0, False
    t0 = time.perf_counter()
    while not fin:
        r.draw()
        if with_present:
            r.present(pw, ph)
        n += 1
        fin = b.update(1.0 / fps)
    r.wait()
    dt = time.perf_counter() - t0
    r.deinit()
    return dt / n * 1e3


pipe = {"two_in_flight_present_on_the_frames_stream_ms": pipelined(0), "two_in_flight_present_on_its_own_stream_ms": pipelined(L.TUNE_PRESENT_OWN_STREAM),
        "two_in_flight_trace_only_ms": pipelined(0, with_present=False)}
frame = [a + b for a, b in zip(trace, present)]
rec = {"workload": w.name, "protocol": "Benchmark.zig:141-172 scripted path, 60 s at a fixed simulated frame rate; Report = min / max / avg frame ms (Benchmark.zig:109-136)",
       "frames": len(trace), "simulated_fps": fps, "present_size": [pw, ph], "timing": "hipEventElapsedTime around each launch",
       "trace": stats(trace), "present": stats(present), "frame_trace_plus_present": stats(frame),
       "pipelined_avg_frame_ms": pipe, "pipelined_note": "host clock over the whole path / frames, nothing waited for until the end; the serial figures above are per-frame HIP events, one frame at a time",
       "min_frame_ms": min(trace), "max_frame_ms": max(trace), "avg_frame_ms": sum(trace) / len(trace),
       "rays": rays, "Mrays_per_s_trace": rays / (sum(trace) * 1e-3) / 1e6, "kernel": kernel}
line = json.dumps(rec)
print(line)
if out_path:
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    with open(out_path, "w") as f:
        f.write(line + "\n")
