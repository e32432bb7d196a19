"""Recipe of `oracle/_ref`: the reference's own compute shader under Mesa llvmpipe.

TEST INFRASTRUCTURE (oracle side; see oracle/ref_gl/glref.c).  Run in the build container, where
/root/reference exists:

    python -m oracle.ref_gl.recipe            # builds oracle/_ref/libglref.so + oracle/_ref/*.glbin

What it does
  1. compiles the runner oracle/ref_gl/glref.c -> oracle/_ref/libglref.so (our code; a GL context on llvmpipe
     through the DRI swrast loader interface, no display needed);
  2. reads /root/reference/assets/shaders/brick_raytracer.comp and rand.comp WHERE THEY LIE, applies the dialect
     edits below IN MEMORY (the text is never written to disk), has Mesa 23.2.1's GLSL compiler compile it, and
     stores Mesa's program binary (serialised NIR — no source text) as oracle/_ref/brick_raytracer.b<B>.<fmt>.glbin
     for B in {4, 8} and fmt in {rgba8, rgba32f}; likewise image.vert + image.frag (the present / denoise pass) as
     oracle/_ref/image_present.glbin.  oracle/_ref/ is git-ignored and travels to the GPU box, where the
     same image holds the same Mesa build, so the binaries load there without /root/reference.

Why edits are needed at all: the shader is Vulkan GLSL (built by the reference through a network-fetched glslang
wrapper, build.zig:123-158); Mesa's OpenGL front end compiles standard GLSL 4.50.  Every edit replaces a
Vulkan-only declaration by its OpenGL spelling; none touches a statement of main(), RayColor, GridHit, BrickHit,
AdvNormIntersect, the scatter functions or rand.comp apart from the two 8-bit array reads:

  E1  `#include "rand.comp"` (GL_GOOGLE_include_directive) -> the text of rand.comp, inline.
  E2  `#extension` lines for GL_EXT_debug_printf, *_int8, 8bit_storage, include_directive: removed (unknown to OpenGL).
  E3  `local_size_x_id = 0, local_size_y_id = 1` -> literal `local_size_x = 32, local_size_y = 32`
      (specialization constants 0/1; the reference passes floor(sqrt(maxComputeWorkGroupInvocations)),
      ComputePipeline.zig:588-597 — 32 on a 1024-invocation device; the value cannot influence a pixel).
  E4  `layout (constant_id = N) const T name = <default>;` -> `const T name = <value>;` with the values the
      reference specialises with (Pipeline.zig:293-315): brick_bits = b^3, brick_bytes = b^3/8,
      brick_dimensions = b, brick_voxel_scale = 1/b.
  E5  `layout (push_constant) uniform PushConstants` -> `layout (std140, binding = 0) uniform PushConstants`
      (same offsets: scalars at 0/4, vec3s at 16/32/48/64, float at 76, ints at 80/84, vec3 96, uint 108, vec3 112,
      float 124 — the 128 bytes pushed at ComputePipeline.zig:488-505).
  E6  uniform / buffer blocks get the layouts Vulkan implies: `layout (binding = 1) uniform` -> std140,
      `layout (binding = N) buffer` -> std430 (OpenGL's default would be `shared`).
  E7  8-bit storage (GL_EXT_shader_8bit_storage is Vulkan-only): `readonly uint8_t NAME[];` -> `readonly uint
      NAME_w[];` plus `int NAME_u8(uint i)` = byte i of the same memory; the two reads `NAME[expr]` become
      `NAME_u8(expr)`; the type `uint8_t` -> `int` and the constructor `uint8_t(x)` -> `(int(x) & 0xFF)` (every
      such value is a byte; same integers).
  E8  (rgba32f variant only) image format `Rgba8` -> `rgba32f`, so that the colour can be compared before the UNORM8
      rounding of the store (the rgba8 variant keeps the reference's format).
"""
from __future__ import annotations

import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(os.path.dirname(HERE), "_ref")
REFERENCE_SHADERS = "/root/reference/assets/shaders"
WORKGROUP = 32          # E3
BRICK_DIMENSIONS = (4, 8)
FORMATS = ("rgba8", "rgba32f")


def reference_available() -> bool:
    return os.path.exists(os.path.join(REFERENCE_SHADERS, "brick_raytracer.comp"))


def build_runner(force: bool = False) -> str:
    os.makedirs(REF_DIR, exist_ok=True)
    src = os.path.join(HERE, "glref.c")
    out = os.path.join(REF_DIR, "libglref.so")
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall", "-Wextra",
                               "-o", out, src, "-ldl"])
    return out


def _sub_once(pattern: str, repl, text: str, what: str, count: int = 1, flags: int = 0) -> str:
    new, n = re.subn(pattern, repl, text, flags=flags)
    if n != count:
        raise RuntimeError(f"dialect edit {what}: expected {count} match(es), found {n} — the reference shader changed")
    return new


def opengl_dialect(brick_dimension: int, fmt: str) -> str:
    """The reference shader's text with edits E1..E8 applied (in memory only)."""
    with open(os.path.join(REFERENCE_SHADERS, "brick_raytracer.comp")) as fh:
        src = fh.read()
    with open(os.path.join(REFERENCE_SHADERS, "rand.comp")) as fh:
        rand = fh.read()
    b = brick_dimension
    src = _sub_once(r'^#include "rand\.comp"[ \t]*$', lambda m: rand, src, "E1", flags=re.M)
    src = _sub_once(r"^#extension GL_(EXT_debug_printf|EXT_shader_explicit_arithmetic_types_int8|EXT_shader_8bit_storage|"
                    r"GOOGLE_include_directive)[^\n]*$", "", src, "E2", count=4, flags=re.M)
    src = _sub_once(r"local_size_x_id = 0, local_size_y_id = 1", f"local_size_x = {WORKGROUP}, local_size_y = {WORKGROUP}", src, "E3")
    spec = {"brick_bits": f"{b ** 3}U", "brick_bytes": f"{b ** 3 // 8}U", "brick_dimensions": f"{b}", "brick_voxel_scale": repr(1.0 / b)}
    for name, value in spec.items():
        src = _sub_once(r"layout \(constant_id = \d\) const (\w+) " + name + r" = [^;]*;",
                        lambda m, name=name, value=value: f"const {m.group(1)} {name} = {value};", src, f"E4 {name}")
    src = _sub_once(r"layout \(push_constant\) uniform", "layout (std140, binding = 0) uniform", src, "E5")
    src = _sub_once(r"layout \(binding = 1\) uniform", "layout (std140, binding = 1) uniform", src, "E6 uniform")
    src = _sub_once(r"layout \(binding = (\d)\) buffer", r"layout (std430, binding = \1) buffer", src, "E6 buffer", count=6)
    for name in ("brick_solid_mask", "material_indices"):
        src = _sub_once(r"readonly uint8_t " + name + r"\[\];\s*\};",
                        f"readonly uint {name}_w[];\n}};\n"
                        f"int {name}_u8(uint i) {{ return int(({name}_w[i >> 2] >> ((i & 3u) * 8u)) & 0xFFu); }}",
                        src, f"E7 decl {name}")
        src = _sub_once(name + r"\[([^\]]*)\];", name + r"_u8(\1);", src, f"E7 read {name}")
    src = _sub_once(r"uint8_t\(([^;]*)\);", r"(int(\1) & 0xFF);", src, "E7 constructor", count=2)
    src = _sub_once(r"\buint8_t\b", "int", src, "E7 type", count=3)
    if fmt == "rgba32f":
        src = _sub_once(r"layout\(Rgba8, binding = 0\)", "layout(rgba32f, binding = 0)", src, "E8")
    elif fmt != "rgba8":
        raise ValueError(fmt)
    return src


def present_dialect():
    """(vertex, fragment) text of the reference's present pass, image.vert / image.frag (assets/shaders), for OpenGL: the ONE
    edit is E5 (`layout (push_constant) uniform PushConstant` -> `layout (std140, binding = 0) uniform PushConstant`: int,
    float, float, float at offsets 0, 4, 8, 12 either way).  In memory only."""
    with open(os.path.join(REFERENCE_SHADERS, "image.vert")) as fh:
        vs = fh.read()
    with open(os.path.join(REFERENCE_SHADERS, "image.frag")) as fh:
        fs = fh.read()
    fs = _sub_once(r"layout \(push_constant\) uniform", "layout (std140, binding = 0) uniform", fs, "E5 image.frag")
    return vs, fs


PRESENT_BINARY = os.path.join(REF_DIR, "image_present.glbin")


def binary_path(brick_dimension: int, fmt: str) -> str:
    return os.path.join(REF_DIR, f"brick_raytracer.b{brick_dimension}.{fmt}.glbin")


def build(force: bool = False) -> None:
    """Everything under oracle/_ref/.  Needs /root/reference for the program binaries."""
    build_runner(force)
    if not reference_available():
        return
    from . import GlRef
    gl = GlRef()
    shader_mtime = max(os.path.getmtime(os.path.join(REFERENCE_SHADERS, f)) for f in ("brick_raytracer.comp", "rand.comp"))
    for b in BRICK_DIMENSIONS:
        for fmt in FORMATS:
            out = binary_path(b, fmt)
            if not force and os.path.exists(out) and os.path.getmtime(out) >= max(shader_mtime, os.path.getmtime(__file__)):
                continue
            prog = gl.compile(opengl_dialect(b, fmt))
            gl.save_binary(prog, out)
            gl.delete(prog)
    present_mtime = max(os.path.getmtime(os.path.join(REFERENCE_SHADERS, f)) for f in ("image.vert", "image.frag"))
    if force or not os.path.exists(PRESENT_BINARY) or os.path.getmtime(PRESENT_BINARY) < max(present_mtime, os.path.getmtime(__file__)):
        prog = gl.compile_raster(*present_dialect())
        gl.save_binary(prog, PRESENT_BINARY)
        gl.delete(prog)


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print("oracle/_ref:", ", ".join(sorted(os.listdir(REF_DIR))))
