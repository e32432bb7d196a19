"""ctypes wrapper of oracle/_ref: the reference's compute shader run by Mesa llvmpipe on the host cores.

TEST INFRASTRUCTURE.  Used by tests/ (to pin oracle/vrt_oracle.c against the real shader), by
tests/golden/make_ref_golden.py (fixtures) and by bench.py's cpu_baseline leg (`kind: "reference"`).
Never imported by zig_vulkan_amd.  See recipe.py for how oracle/_ref is made and what is edited.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Tuple

import numpy as np

from . import recipe

_FORMAT_FILE_MAGIC = b"GLRF"


class GlRefUnavailable(RuntimeError):
    pass


class GlRef:
    """One llvmpipe GL 4.5 context per process (thread-affine: use from the thread that made it)."""

    _lib = None

    def __init__(self):
        if GlRef._lib is None:
            path = os.path.join(recipe.REF_DIR, "libglref.so")
            if not os.path.exists(path):
                try:
                    recipe.build_runner()
                except Exception as e:  # noqa: BLE001
                    raise GlRefUnavailable(f"oracle/_ref/libglref.so missing and not buildable: {e}") from e
            L = C.CDLL(path)
            L.glref_last_error.restype = C.c_char_p
            L.glref_info.restype = C.c_char_p
            L.glref_compile.argtypes = [C.c_char_p, C.c_int]
            L.glref_get_binary.restype = C.c_int64
            L.glref_get_binary.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_uint32)]
            L.glref_load_binary.argtypes = [C.c_void_p, C.c_int64, C.c_uint32]
            L.glref_delete_program.argtypes = [C.c_int]
            _bufs = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
            L.glref_buffers_set.argtypes = _bufs + _bufs
            L.glref_ubo_update.argtypes = [C.c_int, C.c_void_p, C.c_int64]
            L.glref_run.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
            L.glref_compile_raster.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]
            L.glref_present.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]
            L.glref_dispatch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                                         C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                                         C.c_int, C.c_void_p]
            if L.glref_init() != 0:
                raise GlRefUnavailable("llvmpipe context: " + L.glref_last_error().decode(errors="replace"))
            GlRef._lib = L
        self.L = GlRef._lib

    def info(self) -> str:
        return self.L.glref_info().decode()

    def _err(self) -> str:
        return self.L.glref_last_error().decode(errors="replace")

    def compile(self, source: str) -> int:
        s = source.encode()
        prog = self.L.glref_compile(s, len(s))
        if prog <= 0:
            raise RuntimeError("Mesa GLSL compiler: " + self._err())
        return prog

    def compile_raster(self, vertex: str, fragment: str) -> int:
        v, f = vertex.encode(), fragment.encode()
        prog = self.L.glref_compile_raster(v, len(v), f, len(f))
        if prog <= 0:
            raise RuntimeError("Mesa GLSL compiler: " + self._err())
        return prog

    def present(self, prog: int, image_rgba8: np.ndarray, out_w: int, out_h: int, ubo: np.ndarray) -> np.ndarray:
        img = np.ascontiguousarray(image_rgba8, dtype=np.uint8)
        ubo = np.ascontiguousarray(ubo)
        out = np.zeros((out_h, out_w, 4), dtype=np.float32)
        if self.L.glref_present(prog, img.ctypes.data, img.shape[1], img.shape[0], 0, ubo.ctypes.data, ubo.nbytes, out_w, out_h, out.ctypes.data) != 0:
            raise RuntimeError(self._err())
        return out

    def save_binary(self, prog: int, path: str) -> None:
        n = self.L.glref_get_binary(prog, None, 0, None)
        if n <= 0:
            raise RuntimeError(self._err())
        buf = (C.c_uint8 * n)()
        fmt = C.c_uint32()
        if self.L.glref_get_binary(prog, buf, n, C.byref(fmt)) != n:
            raise RuntimeError(self._err())
        with open(path, "wb") as fh:
            fh.write(_FORMAT_FILE_MAGIC + int(fmt.value).to_bytes(4, "little") + bytes(buf))

    def load_binary(self, path: str) -> int:
        with open(path, "rb") as fh:
            blob = fh.read()
        if blob[:4] != _FORMAT_FILE_MAGIC:
            raise RuntimeError(f"{path}: not a glref program binary")
        fmt = int.from_bytes(blob[4:8], "little")
        body = blob[8:]
        prog = self.L.glref_load_binary(body, len(body), fmt)
        if prog <= 0:
            # Mesa keys a program binary by a SHA-1 of the driver build AND the host's CPU feature bits
            # (llvmpipe: lp_disk_cache_create -> update_cache_sha1_cpu), so a binary made in the build container is
            # refused on a box with another CPU although its payload — serialised NIR, made before any machine code —
            # is the same.  Header (src/mesa/main/program_binary.c): u32 internal_format, u8 sha1[20], u32 size,
            # u32 crc32 of the payload.  Re-key: take the SHA-1 this host's driver writes into a binary of its own.
            body = body[:4] + self._local_driver_sha1() + body[24:]
            prog = self.L.glref_load_binary(body, len(body), fmt)
        if prog <= 0:
            raise RuntimeError(self._err())
        return prog

    def _local_driver_sha1(self) -> bytes:
        probe = self.compile("#version 450\nlayout(local_size_x = 1) in;\nvoid main() {}\n")
        n = self.L.glref_get_binary(probe, None, 0, None)
        buf = (C.c_uint8 * n)()
        fmt = C.c_uint32()
        self.L.glref_get_binary(probe, buf, n, C.byref(fmt))
        self.delete(probe)
        return bytes(buf[4:24])

    def delete(self, prog: int) -> None:
        self.L.glref_delete_program(prog)

    @staticmethod
    def _pack(d):
        keys = sorted(d)
        arrs = [np.ascontiguousarray(d[k]) for k in keys]
        return (len(keys), (C.c_int * len(keys))(*keys), (C.c_void_p * len(keys))(*[a.ctypes.data for a in arrs]),
                (C.c_int64 * len(keys))(*[a.nbytes for a in arrs]), arrs)

    def set_buffers(self, ubos: Dict[int, np.ndarray], ssbos: Dict[int, np.ndarray]) -> None:
        """Create, fill and bind the blocks (copied: the arrays need not outlive the call); they stay bound."""
        nu, ub, up, un, keep_u = self._pack(ubos)
        ns, sb, sp, sn, keep_s = self._pack(ssbos)
        rc = self.L.glref_buffers_set(nu, ub, up, un, ns, sb, sp, sn)
        del keep_u, keep_s
        if rc != 0:
            raise (GlRefUnavailable if rc == -2 else RuntimeError)(self._err())

    def update_ubo(self, binding: int, data: np.ndarray) -> None:
        data = np.ascontiguousarray(data)
        if self.L.glref_ubo_update(binding, data.ctypes.data, data.nbytes) != 0:
            raise RuntimeError(self._err())

    def run(self, prog: int, width: int, height: int, out_float: bool, read: bool = True,
            workgroup: Tuple[int, int] = (recipe.WORKGROUP, recipe.WORKGROUP)) -> Optional[np.ndarray]:
        out = np.zeros((height, width, 4), dtype=np.float32 if out_float else np.uint8) if read else None
        rc = self.L.glref_run(prog, width, height, workgroup[0], workgroup[1], 1 if out_float else 0,
                              out.ctypes.data if read else None)
        if rc != 0:
            raise RuntimeError(self._err())
        return out

    def clear_buffers(self) -> None:
        self.L.glref_buffers_clear()

    def worker_threads(self) -> int:
        return int(self.L.glref_worker_threads())

    def dispatch(self, prog: int, width: int, height: int, ubos: Dict[int, np.ndarray], ssbos: Dict[int, np.ndarray],
                 out_float: bool, workgroup: Tuple[int, int] = (recipe.WORKGROUP, recipe.WORKGROUP)) -> np.ndarray:
        self.set_buffers(ubos, ssbos)
        try:
            return self.run(prog, width, height, out_float, True, workgroup)
        finally:
            self.clear_buffers()


class ReferenceShader:
    """brick_raytracer.comp (reference) for one brick dimension, as built by recipe.py.

    render(scene, pc) takes the same inputs as oracle.render: an OracleScene (bindings 1..7) and the 128
    push-constant bytes; returns (rgba32f or None, rgba8)."""

    def __init__(self, brick_dimension: int, want_float: bool = True):
        self.gl = GlRef()
        self.brick_dimension = brick_dimension
        self.progs = {}
        for fmt in (("rgba8", "rgba32f") if want_float else ("rgba8",)):
            path = recipe.binary_path(brick_dimension, fmt)
            if os.path.exists(path):
                try:
                    self.progs[fmt] = self.gl.load_binary(path)
                    continue
                except RuntimeError:
                    if not recipe.reference_available():
                        raise
            if not recipe.reference_available():
                raise GlRefUnavailable(f"{path} missing and /root/reference absent: run `python -m oracle.ref_gl.recipe` "
                                       "in the build container")
            self.progs[fmt] = self.gl.compile(recipe.opengl_dialect(brick_dimension, fmt))

    @staticmethod
    def _blocks(scene, pc: np.ndarray):
        ubos = {0: np.frombuffer(pc.tobytes(), dtype=np.uint8), 1: scene.grid_state}
        ssbos = {2: scene.materials.view(np.uint8).reshape(-1), 3: scene.brick_status, 4: scene.brick_index,
                 5: scene.brick_occupancy, 6: scene.brick_start_index, 7: scene.material_index}
        return ubos, ssbos

    def render(self, scene, pc: np.ndarray, want_float: bool = True, want_u8: bool = True):
        assert scene.brick_dimension == self.brick_dimension
        w, h = (int(v) for v in np.frombuffer(pc[:8].tobytes(), dtype=np.uint32))
        ubos, ssbos = self._blocks(scene, pc)
        f = self.gl.dispatch(self.progs["rgba32f"], w, h, ubos, ssbos, True) if want_float else None
        u = self.gl.dispatch(self.progs["rgba8"], w, h, ubos, ssbos, False) if want_u8 else None
        return f, u

    # -- resident scene (bench.py's cpu_baseline leg): buffers uploaded once, 128 constant bytes per frame --
    def bind(self, scene, pc: np.ndarray) -> None:
        assert scene.brick_dimension == self.brick_dimension
        ubos, ssbos = self._blocks(scene, pc)
        self.gl.set_buffers(ubos, ssbos)

    def frame(self, pc: np.ndarray, read: bool = False, want_float: bool = False) -> Optional[np.ndarray]:
        """One dispatch over the bound scene (the Rgba8 build, or the rgba32f build of the same shader); returns when it has
        finished, with the frame if `read`."""
        w, h = (int(v) for v in np.frombuffer(pc[:8].tobytes(), dtype=np.uint32))
        self.gl.update_ubo(0, np.frombuffer(pc.tobytes(), dtype=np.uint8))
        return self.gl.run(self.progs["rgba32f" if want_float else "rgba8"], w, h, want_float, read)

    def unbind(self) -> None:
        self.gl.clear_buffers()


class ReferencePresent:
    """The reference's present / denoise pass (image.vert + image.frag) as built by recipe.py: one fullscreen quad into an
    out_w x out_h float target, sampling the traced RGBA8 image with a linear / repeat sampler."""

    def __init__(self):
        self.gl = GlRef()
        if os.path.exists(recipe.PRESENT_BINARY):
            try:
                self.prog = self.gl.load_binary(recipe.PRESENT_BINARY)
                return
            except RuntimeError:
                if not recipe.reference_available():
                    raise
        if not recipe.reference_available():
            raise GlRefUnavailable(f"{recipe.PRESENT_BINARY} missing and /root/reference absent")
        self.prog = self.gl.compile_raster(*recipe.present_dialect())

    def render(self, image_rgba8: np.ndarray, out_w: int, out_h: int, samples: int = 20, distribution_bias: float = 0.6,
               pixel_multiplier: float = 1.5, inverse_hue_tolerance: float = 20.0) -> np.ndarray:
        pc = np.zeros(1, dtype=np.dtype([("samples", np.int32), ("b", np.float32), ("m", np.float32), ("t", np.float32)]))
        pc[0] = (samples, distribution_bias, pixel_multiplier, inverse_hue_tolerance)   # GraphicsPipeline.PushConstant, GraphicsPipeline.zig:27-32
        return self.gl.present(self.prog, image_rgba8, out_w, out_h, pc.view(np.uint8))


def available() -> Optional[str]:
    """None when the reference shader can be run here, else the reason."""
    try:
        GlRef()
    except (GlRefUnavailable, OSError) as e:
        return str(e)
    if recipe.reference_available():
        return None
    missing = [recipe.binary_path(b, f) for b in recipe.BRICK_DIMENSIONS for f in recipe.FORMATS
               if not os.path.exists(recipe.binary_path(b, f))]
    if not os.path.exists(recipe.PRESENT_BINARY):
        missing.append(recipe.PRESENT_BINARY)
    return ("missing " + ", ".join(os.path.basename(m) for m in missing)) if missing else None
