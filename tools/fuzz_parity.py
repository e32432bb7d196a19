#!/usr/bin/env python3
"""Randomised parity fuzz (usage: fuzz_parity.py [cases] [seed] [big|pow2|pool]; pow2: grids and frames that take vrt_path_kernel's block-skipping walk;
pool: the pow2 draw narrowed to what vrt_pool_kernel takes — both brick sizes (round 5), three power-of-two dimensions, two or three
bounces; voxels in two opposite corners so that the occupied cells' box is the grid, or (three cases in ten) any box with
VRT_TUNE_GRID_EXIT_ANY_BOX; half of the cases with ONE material per brick (the byte-per-cell material, TraceParams::cell_material)): random small grids (odd dimensions, both brick sizes, any
scale, sparse allocation), random materials incl. glass / metal / unknown types, random cameras inside and outside the
box, samples 1-3, bounces 0-2, sun on/off with and without jitter — product kernel against the oracle, whole frames,
float target bit for bit.  The committed tests pin chosen cases; this looks for the ones nobody chose.  A mismatching case is
dumped whole (seven buffers, push constants, both frames) to gpurun_out/fuzz_fail_seed<seed>_case<case>.npz."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from zig_vulkan_amd import BrickGrid, Config, CameraConfig, SunConfig, VoxelRT, default_materials
from helpers import O, oracle_scene_from_grid

def fuzz(cases: int, seed: int, big: bool = False, verbose: bool = True, pow2: bool = False, library=None, pool: bool = False) -> int:
    """Returns the number of mismatching cases.  library: path of the development build (make dev) — the variants that lost their
    A/B measurement then take part in the draw; without it only the kernels of the product build are drawn."""
    dev = library is not None
    PATH = 1 << 23
    big_variants = ([0, 0, 6, 1, 0x10070000, 0x10070000, PATH, PATH | (1 << 22), PATH | (5 << 8), PATH | (2 << 24)] if dev
                    else [0, 0, 9, 5, 0x10070000, 0x10070000, PATH, PATH | (3 << 24), PATH | (5 << 8), PATH | (2 << 24)])
    pow2_variants = [PATH | (1 << 22), PATH | (1 << 22) | (5 << 8), PATH] if dev else [PATH, PATH | (5 << 8), PATH | (2 << 24)]
    rng = np.random.default_rng(seed)
    pow2 = pow2 or pool
    bad = 0
    used = {}
    for case in range(cases):
        b = int(rng.choice([4, 8]))
        dims = [int(rng.integers(1, 25 if big else 9)) for _ in range(3)]
        if pow2:  # grids the path kernel's block filter accepts: x, z powers of two >= 4, y a multiple of 4
            dims = [int(rng.choice([4, 8, 16, 32])), int(rng.choice([4, 8, 12, 16, 20])), int(rng.choice([4, 8, 16, 32]))]
        any_box = False
        if pool:
            dims = [int(rng.choice([4, 8, 16, 32])), int(rng.choice([4, 8, 16])), int(rng.choice([4, 8, 16, 32]))]
            any_box = bool(rng.random() < 0.3)
        scale = float(rng.choice([0.5, 1.0, 2.0, 0.3, 1.7, 4.0]))
        min_point = [float(-0.5 * d * scale + rng.normal() * 0.3) for d in dims]
        cells = dims[0] * dims[1] * dims[2]
        grid = BrickGrid(*dims, min_point=min_point, scale=scale, brick_dimension=b)
        n = int(rng.integers(1, max(2, int(0.2 * cells * b ** 3))))
        xyz = np.stack([rng.integers(0, b * d, n) for d in dims], axis=-1)
        if pow2 and rng.random() < 0.7:  # mostly empty space with a few clumps: whole 4x4x4 blocks of cells without a voxel
            k = int(rng.integers(1, 6))
            centres = np.stack([rng.integers(0, b * d, k) for d in dims], axis=-1)
            xyz = np.clip(centres[rng.integers(0, k, n)] + rng.integers(-2 * b, 2 * b + 1, (n, 3)), 0, np.array(dims) * b - 1)
        if (not pool or any_box) and rng.random() < 0.5:  # the occupied cells fill only a sub-box of the grid: rays enter the grid in front of it (skip_to_box)
            lo = [int(rng.integers(0, d)) for d in dims]
            hi = [int(rng.integers(l, d)) for l, d in zip(lo, dims)]
            xyz = np.stack([b * l + rng.integers(0, b * (h - l + 1), n) for l, h in zip(lo, hi)], axis=-1)
        if rng.random() < 0.5:  # clumps: whole columns
            xyz[:, 1] = rng.integers(0, b * dims[1], n) // 2 * 2
        if pool and not any_box:  # the box of the occupied cells is the grid
            xyz = np.concatenate([xyz, np.array([[0, 0, 0], [b * d - 1 for d in dims]])])
            n += 2
        voxel_mats = rng.integers(0, 14, n)
        if pool and rng.random() < 0.5:  # one material per brick (insert() flips y brick-wise: bricks stay bricks)
            c = xyz // b
            voxel_mats = (c[:, 0] * 7 + c[:, 1] * 13 + c[:, 2] * 31 + int(rng.integers(0, 14))) % 14
        grid.insert_many(xyz, voxel_mats)
        mats = default_materials(256)
        mats[8] = (2, 0.9, 0.95, 1.0, 1.52)
        mats[9] = (7, 0.9, 0.2, 0.9, 1.0)
        mats[10] = (3, 0.3, 0.9, 0.3, 1.0)
        if pool and rng.random() < 0.6:
            # (no record of the type MAT_NONE: vrt_pool_kernel then leaves a hit's material to the round of transitions that shades it,
            # TraceParams::materials_plain — with the record above in the table every case took the other path)
            mats[10] = (0, 0.3, 0.9, 0.3, 0.0)
        mats[11] = (1, 0.8, 0.8, 0.8, 0.05)
        mats[12] = (2, 0.9, 0.9, 1.0, 1.0)
        mats[13] = (1, 0.7, 0.6, 0.5, 0.6)
        w, h = int(rng.integers(1, 400 if big else 90)), int(rng.integers(1, 260 if big else 70))
        spp, bounce = int(rng.integers(1, 4)), int(rng.integers(0, 3))
        if pow2:
            bounce = int(rng.integers(2 if pool else 1, 4))
        sun_on, radius = bool(rng.random() < 0.7), float(rng.choice([0.0, 5.0, 40.0]))
        rt = VoxelRT(grid, Config(internal_resolution_width=w, internal_resolution_height=h, camera=CameraConfig(samples_per_pixel=spp, max_bounce=bounce),
                                  sun=SunConfig(enabled=sun_on, radius=radius), want_float_output=True,
                                  library=library, tuning_flags=(1 << 19) if any_box else 0,
                                  kernel_variant=PATH if pool else (int(rng.choice(big_variants)) if big
                                  else int(rng.choice(pow2_variants) if pow2 else rng.choice([0, 0, PATH])))))
        rt.push_materials(mats)
        size = np.array(dims) * scale
        centre = np.array(min_point) + 0.5 * size
        origin = centre + (rng.random(3) - 0.5) * size * (3.0 if rng.random() < 0.7 else 0.9)
        rt.camera.look_at(origin.tolist(), (centre + (rng.random(3) - 0.5) * size * 0.5).tolist())
        if (rt.config.kernel_variant >> 16) & 0xF == 7:
            # cost-ordered launch, re-sorted every 2 frames: the compared frame comes after three re-sorts, with split tiles
            rt.draw(frames=7)
        if pow2:
            # (the library learns the box of the occupied cells behind the upload and never waits for it: with a frame and a wait
            # behind it, a scene that fills its grid takes the dilated-index walk without steps-left counters)
            rt.draw()
            rt.wait()
        rt.draw()
        f, u = rt.read_rgba32f(), rt.read_rgba8()
        pc = O.push_constants(rt.camera.blob(), rt.sun.blob())
        name = rt.kernel_name()
        used[name] = used.get(name, 0) + 1
        rt.deinit()
        fo, uo, co = O.render(oracle_scene_from_grid(grid, mats), pc)
        # bit for bit, except that any NaN equals any NaN (a 1-pixel-wide image divides 0 by 0 in comp:168-170; the sign and
        # payload of the resulting NaN differ between x86 and gfx950, its RGBA8 value 0 does not)
        both_nan = np.isnan(f) & np.isnan(fo)
        ok = np.array_equal(f.view(np.uint32)[~both_nan], fo.view(np.uint32)[~both_nan]) and np.array_equal(u, uo)
        if not ok:
            bad += 1
            # the whole case, for a bit-level post-mortem off the box (an unexplained mismatch must leave its inputs)
            out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
            os.makedirs(out_dir, exist_ok=True)
            sc = oracle_scene_from_grid(grid, mats)
            np.savez_compressed(os.path.join(out_dir, f"fuzz_fail_seed{seed}_case{case}.npz"), kernel=np.array(name), brick_dimension=np.int32(b),
                                grid_state=sc.grid_state, materials=sc.materials.view(np.uint8).reshape(-1), brick_status=sc.brick_status,
                                brick_index=sc.brick_index, brick_occupancy=sc.brick_occupancy, brick_start_index=sc.brick_start_index,
                                material_index=sc.material_index, push_constants=pc, hip_rgba32f=f, oracle_rgba32f=fo, hip_rgba8=u, oracle_rgba8=uo,
                                kernel_variant=np.int64(rt.config.kernel_variant))
            du = np.argwhere(u != uo)
            db = np.argwhere((f.view(np.uint32) != fo.view(np.uint32)) & ~both_nan)
            print(f"   float bits differ at {len(db)} places, first {[(i.tolist(), float(f[tuple(i)]), float(fo[tuple(i)]), hex(int(f.view(np.uint32)[tuple(i)])), hex(int(fo.view(np.uint32)[tuple(i)]))) for i in db[:4]]}")
            print(f"case {case}: MISMATCH  {name} b={b} dims={dims} scale={scale} {w}x{h} spp={spp} bounce={bounce} sun={sun_on}/{radius}  {np.count_nonzero(f != fo)} floats differ, "
                  f"{len(du)} RGBA8 bytes differ, first {[(i.tolist(), int(u[tuple(i)]), int(uo[tuple(i)])) for i in du[:4]]}")
        elif verbose and case % 10 == 0:
            print(f"case {case}: ok  b={b} dims={dims} scale={scale} {w}x{h} spp={spp} bounce={bounce} sun={sun_on}/{radius} rays={co['rays']} hits={co['hits']} {name.split('<')[0]}")
    if verbose:
        print(f"{cases} cases, {bad} mismatching; kernels: " + ", ".join(f"{k} x{v}" for k, v in sorted(used.items())))
    return bad


if __name__ == "__main__":
    sys.exit(1 if fuzz(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 7,
                       len(sys.argv) > 3 and sys.argv[3] in ("big", "pow2", "pool"), pow2=len(sys.argv) > 3 and sys.argv[3] == "pow2",
                       pool=len(sys.argv) > 3 and sys.argv[3] == "pool") else 0)
