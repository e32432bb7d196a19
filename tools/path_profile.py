#!/usr/bin/env python3
"""Phase profile of vrt_path_kernel (frames with bounces).  Needs the library built with the development profile:
    make -C zig_vulkan_amd/csrc -B EXTRA=-DVRT_DEV_PROFILE
usage: VRT_HIP_LIB=zig_vulkan_amd/libvrt_hip_prof.so path_profile.py <workload> <view> [variant] [tuning_flags]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zig_vulkan_amd import workloads as W

w = W.WORKLOADS[sys.argv[1]]
view = sys.argv[2]
variant = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0
flags = int(sys.argv[4], 0) if len(sys.argv) > 4 else 0
grid = W.build_grid(w)
rt = W.make_renderer(w, grid, kernel_variant=variant, tuning_flags=flags)
rc = W.make_renderer(w, grid, kernel_variant=variant, enable_counters=True)
W.set_view(rt, view); W.set_view(rc, view)
rt.draw(); rt.wait()   # (behind a finished frame the library knows the box of the occupied cells)
rt.draw(); rt.wait()
rt_raw_int = rt.wave_timeline(raw=True).reshape(-1)
raw = rt_raw_int.astype(float)
kernel = rt.kernel_name()
pr = raw[:12]
bw = raw[12:20]
ms = rt.last_kernel_ms()
rc.draw(); c = rc.counters()
t_trans, t_walk, t_brick = pr[0:3]
n_tr, n_wait, n_calls, n_alive_in, n_alive_out, n_brick, n_parked, n_nostage, waves = pr[3], pr[4], pr[5], pr[6], pr[7], pr[8], pr[9], pr[10], pr[11]
tot = t_trans + t_walk + t_brick
print(f"{w.name} {view} variant {variant:#x} {kernel}: {ms:.2f} ms, {int(waves)} waves, rays {c['rays']/1e6:.1f}M, grid steps/ray {c['grid_steps']/c['rays']:.1f}, bricks/ray {c['bricks_entered']/c['rays']:.2f}, "
      f"voxel steps/ray {c['voxel_steps']/c['rays']:.1f}")
print(f"  cycles: transitions {100*t_trans/tot:.1f} %, walk loop {100*t_walk/tot:.1f} %, bricks {100*t_brick/tot:.1f} %  (sum over waves {tot/1e9:.2f} G cycles)")
print(f"  transitions: {n_tr/1e6:.2f} M rounds, {n_wait/n_tr:.1f} waiting lanes per round, {t_trans/n_tr:.0f} cycles per round")
print(f"  walk loop: {n_calls/1e6:.2f} M calls, {n_alive_in/n_calls:.1f} lanes at entry, {n_alive_out/n_calls:.1f} still moving at exit, {t_walk/n_calls:.0f} cycles per call; "
      f"lane-trips {c['grid_steps']/1e6:.0f} M -> {c['grid_steps']/n_calls:.1f} lane-trips per call")
print(f"  bricks: {n_brick/1e6:.2f} M rounds, {n_parked/n_brick:.1f} parked lanes per round, {t_brick/n_brick:.0f} cycles per round")
rt.deinit(); rc.deinit()
if "pool" in kernel and len(raw) > 23 and raw[22]:
    rawi = rt_raw_int
    inv = lambda v: (~int(v)) & 0xFFFFFFFFFFFFFFFF
    begin, first_dry, last_dry, end = inv(rawi[20]), inv(rawi[21]), int(rawi[23]), int(rawi[22])
    span = (end - begin) / 100.0
    print(f"  drain: kernel {span / 1e3:.2f} ms by the wall clock; the first wave finds the pixel counter exhausted at {100 * (first_dry - begin) / (end - begin):.1f} % of it, "
          f"the last at {100 * (last_dry - begin) / (end - begin):.1f} %; the last wave ends {(end - first_dry) / 100.0 / 1e3:.2f} ms after the counter ran out")
if "pool" in kernel:
    print(f"  brick rounds that found both staging areas of the workgroup taken (the wave served another queue): {n_nostage/1e6:.2f} M against {n_brick/1e6:.2f} M rounds run")
    print(f"  vrt_pool_kernel (wave-cycles, share of all): phase rule {100*bw[3]/tot:.1f} %, exchange before a walk call {100*bw[0]/tot:.1f} % ({bw[0]/n_calls:.0f} cycles per call), "
          f"the walk loop itself {100*bw[1]/tot:.1f} % ({bw[1]/n_calls:.0f} cycles per call), exchange before a brick round {100*bw[6]/tot:.1f} % ({bw[6]/n_brick:.0f} per round), "
          f"before a transition round {100*bw[7]/tot:.1f} % ({bw[7]/n_tr:.0f} per round)")
print(f"  inside the brick rounds (wave-cycles, share of all): voxel loops {100*bw[2]/tot:.1f} %, material test + hit record {100*bw[4]/tot:.1f} %, "
      f"brick staged in LDS / first word requested {100*bw[5]/tot:.1f} %, rest of the round (cell -> brick index, walk set-up, hit record) {100*(t_brick-bw[2]-bw[4]-bw[5])/tot:.1f} %")
