"""A second, independent restatement of the traversal (GridHit / BrickHit / AdvNormIntersect), written straight
from the shader's control flow in slow scalar Python with float32 rounding after every operation.

TEST INFRASTRUCTURE ONLY.  Purpose: cross-check oracle/vrt_oracle.c (the C restatement every parity test
leans on) against a restatement that shares no code with it and keeps the shader's own structure — nested
branches for the min-axis step, integer cell positions, the per-word status cache, byte-wise occupancy.  It
covers the traversal only (no RNG, no shading); tests/test_literal_port.py feeds both with the same random
rays and demands identical hit flag, distance, point, normal and material index, bit for bit.

Follows assets/shaders/brick_raytracer.comp: safeInverse :267, GridHit :271-376, BrickHit :378-471,
indexOfMaxComponent :501-503, AdvNormIntersect :522-536, RayAt :192-195.
Conventions shared with the oracle because GLSL leaves them open (oracle header): float->int
conversion clamps, min/max are the GLSL `y < x ? y : x` / `x < y ? y : x` forms, fma is a*b + c with two roundings and dot is
reduced from the last channel (Mesa llvmpipe's lowering: the reference as it executes).
"""
from __future__ import annotations

import numpy as np

f32 = np.float32
INF = f32(np.inf)


def fma(a, b, c) -> np.float32:
    """GLSL fma as the reference's executable implementation lowers it (Mesa llvmpipe, nir lower_ffma32): a*b + c, two roundings —
    the arithmetic contract of the oracle and of the kernels since round 3."""
    with np.errstate(all="ignore"):
        return f32(f32(f32(a) * f32(b)) + f32(c))


def gl_min(x, y):
    return y if y < x else x


def gl_max(x, y):
    return y if x < y else x


def sign(x):
    return f32(1.0) if x > 0 else (f32(-1.0) if x < 0 else f32(0.0))


def to_int(x) -> int:
    """int(float) with the conversion clamped (the contract shared with kernel and oracle)."""
    x = f32(x)
    if np.isnan(x):
        return -(2 ** 31)
    return int(min(max(float(x), -2147483648.0), 2147483520.0))


def safe_inverse(x):
    return f32(1e12) if x == 0 else f32(1.0) / f32(x)


class Scene:
    """The bound buffers as plain numpy arrays (bindings 1-7) and the specialisation constants."""

    def __init__(self, grid_min, grid_max, scale, dims, brick_dimension, status, brick_index, occupancy, start_index, material_index, materials):
        self.g_min = [f32(v) for v in grid_min]
        self.g_max = [f32(v) for v in grid_max]
        self.scale = f32(scale)
        self.dims = [int(d) for d in dims]
        self.b = int(brick_dimension)
        self.brick_bytes = self.b ** 3 // 8
        self.brick_voxel_scale = f32(1.0) / f32(self.b)
        self.status, self.brick_index, self.occupancy = status, brick_index, occupancy
        self.start_index, self.material_index, self.materials = start_index, material_index, materials


def ray_at(origin, direction, t):
    return [fma(t, direction[i], origin[i]) for i in range(3)]


def adv_norm_intersect(sc: Scene, origin, inv_dir, t_min, t_max):
    t_lower = [(sc.g_min[i] - origin[i]) * inv_dir[i] for i in range(3)]
    t_upper = [(sc.g_max[i] - origin[i]) * inv_dir[i] for i in range(3)]
    t_mins = [gl_min(t_lower[i], t_upper[i]) for i in range(3)]
    t_maxes = [gl_max(t_lower[i], t_upper[i]) for i in range(3)]
    index = int(t_mins[1] > t_mins[0] and t_mins[1] > t_mins[2]) + 2 * int(t_mins[2] > t_mins[0] and t_mins[2] > t_mins[1])
    normal = [f32(0.0)] * 3
    normal[index] = sign(inv_dir[index])
    t_min = gl_max(t_min, t_mins[index])
    t_max = gl_min(t_max, gl_min(gl_min(t_maxes[0], t_maxes[1]), t_maxes[2]))
    return bool(t_min <= t_max), normal, t_min, t_max


def _dda_step(side, delta, step, pos, scale, normal_axis):
    """The shader's nested branches; returns the new t_value and normal, updates side/pos in place."""
    if side[0] < side[1]:
        a = 0 if side[0] < side[2] else 2
    else:
        a = 1 if side[1] < side[2] else 2
    t_value = side[a] * scale
    side[a] = side[a] + delta[a]
    pos[a] += step[a]
    normal = [f32(0.0)] * 3
    normal[a] = normal_axis[a]
    return t_value, normal


def _side_dist(step, fposition, delta):
    out = []
    for i in range(3):
        fs = f32(step[i])
        inter = f32(np.floor(fposition[i])) - fposition[i]
        out.append(fma(fs, inter, fs * f32(0.5) + f32(0.5)) * delta[i])
    return out


def brick_hit(sc: Scene, origin, direction, ignore_type, internal_reflection, t_max, delta, step, brick, brick_min, hit):
    voxel_scale = sc.scale * sc.brick_voxel_scale
    base = brick * sc.brick_bytes
    p = ray_at(origin, direction, hit["t"])
    fposition = [(p[i] - brick_min[i]) / voxel_scale for i in range(3)]
    side = _side_dist(step, fposition, delta)
    normal_axis = [f32(1.0) if s < 0 else f32(-1.0) for s in step]
    pos = [to_int(np.floor(fposition[i] + f32(0.0))) for i in range(3)]
    local_t_max = t_max - hit["t"]
    t_value = f32(0.0)
    trips = 0
    while all(0 <= pos[i] < sc.b for i in range(3)) and t_value <= local_t_max:
        trips += 1
        assert trips < 10000
        voxel = pos[0] + sc.b * (pos[2] + sc.b * pos[1])
        byte = int(sc.occupancy[base + voxel // 8])
        if (byte >> (voxel % 8)) & 1:
            start = int(sc.start_index[brick]) & 0x7FFFFFFF
            hit["index"] = int(sc.material_index[start + voxel])
            m = sc.materials[hit["index"]]
            ignore = int(m["type"]) == ignore_type and f32(internal_reflection) == f32(m["type_data"])
            if not ignore:
                t_offset = voxel_scale * f32(0.05)
                hit["t"] = hit["t"] + (t_value - t_offset)
                q = ray_at(origin, direction, hit["t"])
                hit["point"] = [q[i] + hit["normal"][i] * t_offset for i in range(3)]
                return True
        t_value, hit["normal"] = _dda_step(side, delta, step, pos, voxel_scale, normal_axis)
    return False


def grid_hit(sc: Scene, origin, direction, ignore_type=3, internal_reflection=1.0, t_min=f32(0.00001), t_max=INF):
    """Returns (hit?, record) with record = dict(t, point, normal, index)."""
    origin = [f32(v) for v in origin]
    direction = [f32(v) for v in direction]
    hit = {"t": f32(0.0), "point": [f32(0.0)] * 3, "normal": [f32(0.0)] * 3, "index": 0}
    inv = [safe_inverse(d) for d in direction]
    ok, hit["normal"], grid_t_min, grid_t_max = adv_norm_intersect(sc, origin, inv, f32(t_min), f32(t_max))
    if not ok:
        return False, hit
    global_t = grid_t_min + f32(0.0001) * sc.scale
    delta = [f32(abs(v)) for v in inv]
    step = [to_int(sign(d)) for d in direction]
    p = ray_at(origin, direction, global_t)
    fposition = [(p[i] - sc.g_min[i]) / sc.scale for i in range(3)]
    side = _side_dist(step, fposition, delta)
    cached_word_index, cached_word = None, 0
    normal_axis = [f32(1.0) if s < 0 else f32(-1.0) for s in step]
    t_value = f32(0.0)
    pos = [to_int(np.floor(fposition[i] + f32(0.0))) for i in range(3)]
    trips = 0
    while all(0 <= pos[i] < sc.dims[i] for i in range(3)) and global_t <= t_max:
        trips += 1
        assert trips < 100000
        cell = pos[0] + sc.dims[0] * (pos[2] + sc.dims[2] * pos[1])
        if cached_word_index != cell // 32:
            cached_word_index = cell // 32
            cached_word = int(sc.status[cached_word_index])
        if cached_word & (1 << (cell % 32)):
            brick_min = [fma(f32(pos[i]), sc.scale, sc.g_min[i]) for i in range(3)]
            global_t = t_value + grid_t_min + f32(0.01) * sc.scale
            hit["t"] = global_t
            if brick_hit(sc, origin, direction, ignore_type, internal_reflection, grid_t_max, delta, step, int(sc.brick_index[cell]), brick_min, hit):
                return True, hit
        t_value, hit["normal"] = _dda_step(side, delta, step, pos, sc.scale, normal_axis)
    return False, hit


# ---- one pixel of a deterministic frame (samples_per_pixel 1, device max_bounce <= 1, sun radius 0) -----------------
# main :153-178, CameraGetRay :474-477, CreateRay/CreateShadowRay :180-190, BackgroundColor :197-201, RayColor :203-265.
# With one sample the jitter is hash12(0) = 0; with sun radius 0 RandVec3 is the zero vector; with max_bounce <= 1 the
# scatter functions' only products (the next ray, the continue flag) are never read.  normalize(v) = v * (1 / sqrt(dot))
# with dot reduced from the last channel, (z*z + y*y) + x*x: the contract of DESIGN.md §3.

def normalize(v):
    d = f32(f32(f32(v[2] * v[2]) + f32(v[1] * v[1])) + f32(v[0] * v[0]))
    inv = f32(1.0) / f32(np.sqrt(d))
    return [c * inv for c in v]


def pixel(sc: Scene, cam: dict, sun: dict, px: int, py: int):
    """cam: image_width, image_height, horizontal, vertical, lower_left_corner, origin (lists of float32), max_bounce;
    sun: position, enabled, color.  Returns (rgb float32 list, rgba8 list)."""
    u = f32(px) / f32(cam["image_width"] - 1)
    v = f32(py) / f32(cam["image_height"] - 1)
    ray_dir = [fma(cam["horizontal"][i], u, cam["lower_left_corner"][i]) + fma(v, cam["vertical"][i], -cam["origin"][i]) for i in range(3)]
    origin, direction = cam["origin"], normalize(ray_dir)
    sun_on = sun["enabled"] > 0
    color = [f32(0.0)] * 3
    loop_count = 0
    if cam["max_bounce"] > 0:
        ok, hit = grid_hit(sc, origin, direction, 3, 1.0)
        if ok:
            loop_count = 1
            m = sc.materials[hit["index"]]
            attenuation = [f32(m["albedo_r"]), f32(m["albedo_g"]), f32(m["albedo_b"])]
            if int(m["type"]) > 2:
                loop_count = 0
            if sun_on:
                shadow_dir = normalize([f32(sun["position"][i]) - hit["point"][i] for i in range(3)])
                shadowed, _ = grid_hit(sc, hit["point"], shadow_dir, 3, 1.0)
                if not shadowed:
                    color = [color[i] + attenuation[i] * f32(sun["color"][i]) for i in range(3)]
            else:
                color = [color[i] + attenuation[i] for i in range(3)]
    if loop_count == 0:
        t = f32(0.5) * (direction[1] + f32(1.0))
        bg = [fma(f32(1.0) - t, f32(1.0), t * c) for c in (f32(0.5), f32(0.7), f32(1.0))]
        k = [f32(c) for c in sun["color"]] if sun_on else [f32(1.0)] * 3
        color = [color[i] + bg[i] * k[i] for i in range(3)]
    color = [c / (c + f32(1.0)) for c in color]
    color = [f32(np.sqrt(c / f32(1.0))) for c in color]
    # RGBA8 UNORM store: round-to-nearest-even of clamp(c, 0, 1) * 255 in float32 (NaN -> 0), SURVEY.md §8 quirk 10
    rgba8 = [int(np.rint(f32(min(c, f32(1.0))) * f32(255.0))) if c > 0 else 0 for c in color] + [255]
    return color, rgba8
