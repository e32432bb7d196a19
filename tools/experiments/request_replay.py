#!/usr/bin/env python3
"""CPU replay: how often would a lane of vrt_path_kernel's walk loop have to ask for a status word, per structure?
Walks rays (a third from the grid's centre, two thirds from cells next to occupied ones, random directions) through the status
bits of a workload's scene with the plain DDA and counts requests per lane-trip for: the shader's word per cell, half-block
words (4 x 4 x 2 cells), 4 x 4 x 4-cell words, the L1 distance field as vrt_path_kernel<DIST> pipelines it, asked one trip
early, and without pipelining (the ideal), and a Chebyshev-distance cube (ideal).  No GPU needed.  DESIGN.md 4 / 9 quote it.
usage: request_replay.py [workload] [rays]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from scipy import ndimage
from zig_vulkan_amd import _lib as L
from zig_vulkan_amd import workloads as W

w = W.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg4_4k_2048c_b8_sparse"]
n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
grid = W.build_grid(w)
nx = ny = nz = w.voxels // w.brick_dimension
if w.dims:
    nx, ny, nz = w.dims
st = grid.array(L.BUF_BRICK_STATUS)
occ = np.unpackbits(st.view(np.uint8), bitorder="little")[:nx * ny * nz].astype(bool).reshape(ny, nz, nx)   # [y][z][x]
D = ndimage.distance_transform_cdt(~occ, metric="taxicab").astype(np.int32)
Dc = ndimage.distance_transform_cdt(~occ, metric="chessboard").astype(np.int32)
print(f"{w.name}: {nx} x {ny} x {nz} cells, {occ.mean() * 100:.2f} % occupied, mean L1 distance of an empty cell {D[~occ].mean():.2f} (Chebyshev {Dc[~occ].mean():.2f})")
rng = np.random.default_rng(1)
dims = np.array([nx, ny, nz])


def rays(n):
    out = []
    ys, zs, xs = np.nonzero(occ)
    while len(out) < n:
        if rng.random() < 0.33:
            p = dims / 2.0 + rng.random(3)
            d = rng.normal(size=3)
            d[2] = -abs(d[2]) * 2
        else:
            i = rng.integers(len(xs))
            p = np.array([xs[i], ys[i], zs[i]], float) + rng.random(3)
            d = rng.normal(size=3)
            a = np.argmax(abs(d))
            p[a] += np.sign(d[a])
        d /= np.linalg.norm(d)
        c = np.floor(p).astype(int)
        if (c < 0).any() or (c >= dims).any() or occ[c[1], c[2], c[0]]:
            continue
        out.append((p, d))
    return out


def walk(p, d):
    c = np.floor(p).astype(int)
    step = np.where(d >= 0, 1, -1)
    inv = 1 / np.maximum(abs(d), 1e-9)
    sd = np.where(d >= 0, (c + 1 - p), (p - c)) * inv
    cells = [tuple(c)]
    while True:
        a = int(np.argmin(sd))
        sd[a] += inv[a]
        c[a] += step[a]
        if c[a] < 0 or c[a] >= dims[a]:
            return cells
        cells.append(tuple(c))
        if occ[c[1], c[2], c[0]]:
            return cells


tot = 0
req = dict(word_per_cell=0, halfblock_4x4x2=0, block_4x4x4=0, distance_pipelined=0, distance_asked_early=0, distance_ideal=0, chebyshev_ideal=0)
for p, d in rays(n_rays):
    cells = walk(p, d)
    n = len(cells)
    tot += n
    req["word_per_cell"] += n
    hb = b64 = None
    for (x, y, z) in cells:
        k = (x >> 2, z >> 2, y >> 1)
        if k != hb:
            req["halfblock_4x4x2"] += 1
            hb = k
        k = (x >> 2, z >> 2, y >> 2)
        if k != b64:
            req["block_4x4x4"] += 1
            b64 = k
    for key, early in (("distance_pipelined", 0), ("distance_asked_early", 1)):
        K, asked_prev, r = 0, True, 1          # (the word of the first cell is loaded outside the loop)
        for i in range(n - 1):
            need = K <= early
            K -= 1
            r += need
            if asked_prev:
                x, y, z = cells[i]
                dd = D[y, z, x]
                if dd == 0:
                    break
                K = max(K, dd - 2) if early else dd - 2
            asked_prev = need
        req[key] += r
    for key, F in (("distance_ideal", D), ("chebyshev_ideal", Dc)):
        i = r = 0
        while i < n:
            x, y, z = cells[i]
            dd = F[y, z, x]
            r += 1
            if dd == 0:
                break
            if F is D:
                i += dd
            else:
                c0, j = cells[i], i + 1
                while j < n and max(abs(cells[j][0] - c0[0]), abs(cells[j][1] - c0[1]), abs(cells[j][2] - c0[2])) < dd:
                    j += 1
                i = j
        req[key] += r
print(f"{n_rays} rays, {tot / n_rays:.1f} trips per ray")
for k, v in req.items():
    print(f"  {k:24s} {v / tot:.3f} requests per lane-trip")
