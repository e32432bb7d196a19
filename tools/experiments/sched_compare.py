"""Tile order A/B on the headline workload: per-view kernel time (single stream, HIP events) and
throughput with two frames in flight, for the kernel_variants given on the command line.

    python tools/experiments/sched_compare.py 0 0x70000 0x50000
"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401  (HIP runtime first)
from zig_vulkan_amd import workloads as W


def main():
    variants = [int(a, 0) for a in sys.argv[1:]] or [0, 0x70000]
    w = W.WORKLOADS[W.HEADLINE]
    grid = W.build_grid(w)
    for var in variants:
        rt = W.make_renderer(w, grid, kernel_variant=var)
        per = {}
        for v in ["V0", "V1", "V2"]:
            W.set_view(rt, v)
            rt.draw(frames=40)
            rt.draw(frames=200)
            per[v] = rt.last_kernel_ms()
        rt.deinit()
        rt = W.make_renderer(w, grid, kernel_variant=var, frames_in_flight=2)
        thr = {}
        for v in ["V0", "V1", "V2"]:
            W.set_view(rt, v)
            for _ in range(60):
                rt.draw()
            rt.wait()
            t0 = time.perf_counter()
            for _ in range(600):
                rt.draw()
            rt.wait()
            thr[v] = (time.perf_counter() - t0) / 600 * 1e3
        rt.deinit()
        print(f"variant {var:#x}: kernel ms " + " ".join(f"{v} {per[v]:.4f}" for v in per) + f" mean {sum(per.values())/3:.4f}"
              + " | 2 in flight ms/frame " + " ".join(f"{v} {thr[v]:.4f}" for v in thr) + f" mean {sum(thr.values())/3:.4f}", flush=True)


if __name__ == "__main__":
    main()
