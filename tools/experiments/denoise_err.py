import numpy as np, sys
sys.path.insert(0, '.')
from tests.helpers import O
from zig_vulkan_amd import workloads as W
for frame, out, kw in [((320,200),(320,200),{}), ((640,400),(640,400),{}), ((320,200),(400,260),{}), ((320,200),(320,200),dict(inverse_hue_tolerance=7.0))]:
    w = W.Workload("t", frame[0], frame[1], 64, 4, 2, 2, False, 0.0)
    grid = W.build_grid(w)
    rt = W.make_renderer(w, grid)
    W.set_view(rt, "V2"); rt.draw(); traced = rt.read_rgba8()
    u, f = rt.denoise(out[0], out[1], want_float=True, **kw); rt.deinit()
    fo, uo = O.denoise(traced, out[0], out[1], **kw)
    ok = ~np.isnan(fo[..., :3]).any(axis=-1)
    print(frame, out, kw, "max err", np.abs(f[ok] - fo[ok]).max(), "u8 diff", np.abs(u.astype(int) - uo.astype(int))[ok].max())
