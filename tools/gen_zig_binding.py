#!/usr/bin/env python3
"""Regenerates the `extern fn` block of bindings/vrt_hip.zig from include/vrt_hip.h, so that the Zig binding declares
every entry point of the C ABI with the header's own argument lists.  zig is not in this image: the binding cannot be
compiled here, so it is kept in step mechanically instead (tests/test_abi.py re-runs this and fails on drift).

    python tools/gen_zig_binding.py            # rewrites the block between the BEGIN/END markers
    python tools/gen_zig_binding.py --check    # exit 1 if the file is out of date
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vrt_hip.h")
ZIG = os.path.join(ROOT, "bindings", "vrt_hip.zig")
BEGIN = "// BEGIN GENERATED extern declarations (tools/gen_zig_binding.py from include/vrt_hip.h) — do not edit by hand"
END = "// END GENERATED"

STRUCTS = {"vrt_ctx": "Ctx", "vrt_grid": "Grid", "vrt_vox": "Vox", "vrt_benchmark": "Benchmark", "vrt_config": "Config",
           "vrt_grid_state": "GridState", "vrt_material": "Material", "vrt_camera_device": "CameraDevice", "vrt_sun_device": "SunDevice",
           "vrt_shard_info": "ShardInfo", "vrt_counters": "Counters", "vrt_grid_config": "GridConfig", "vrt_camera_config": "CameraConfig",
           "vrt_sun_config": "SunConfig", "vrt_denoise_config": "DenoiseConfig", "vrt_vox_xyzi": "VoxXyzi", "vrt_vox_rgba": "VoxRgba", "vrt_dist_options": "DistOptions"}
OPAQUE = {"Ctx", "Grid", "Vox", "Benchmark"}
SCALARS = {"int": "c_int", "uint32_t": "u32", "uint64_t": "u64", "int32_t": "i32", "int64_t": "i64", "uint8_t": "u8", "float": "f32",
           "double": "f64", "void": "void", "vrt_buffer_id": "BufferId"}


def prototypes(text: str):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)
    text = re.sub(r"\b(typedef\s+)?(struct|enum)\b[^;{]*\{[^}]*\}[^;]*;", " ", text, flags=re.S)
    text = re.sub(r"typedef[^;]*;", " ", text)
    text = text.replace('extern "C" {', " ").replace("}", " ")
    out = []
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(vrt_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        out.append((ret, name, [] if args in ("", "void") else [a.strip() for a in args.split(",")]))
    return out


def zig_type(ctype: str, array: str = "", is_return: bool = False) -> str:
    ctype = " ".join(ctype.split())
    const = "const " if re.search(r"\bconst\b", ctype) else ""
    base = re.sub(r"\bconst\b", "", ctype).replace("*", " ").split()
    base = " ".join(base)
    stars = ctype.count("*")
    zbase = STRUCTS.get(base) or SCALARS.get(base)
    if base == "char":
        assert stars == 1
        return "[*:0]const u8" if is_return else "?[*:0]const u8"
    if zbase is None:
        raise ValueError(f"unknown C type {ctype!r}")
    if array:
        return f"*{const}[{array}]{zbase}"
    if stars == 0:
        return zbase
    if base == "void":
        return f"?*{const}anyopaque" if stars == 1 else f"[*c]?*{const}anyopaque"
    if zbase in OPAQUE:
        return f"?*{const}{zbase}" if stars == 1 else f"*?*{zbase}"
    if base in STRUCTS:
        assert stars == 1
        return f"[*c]{const}{zbase}"      # one struct or an array of them: the C pointer type covers both
    assert stars == 1
    return f"[*c]{const}{zbase}"


def zig_decl(ret, name, args) -> str:
    zargs = []
    for a in args:
        m = re.match(r"(.*?)(\w+)\s*(?:\[(\d+)\])?$", a)
        ctype, pname, arr = m.group(1), m.group(2), m.group(3) or ""
        if pname in ("type", "error", "align", "test", "fn", "var"):
            pname += "_"
        zargs.append(f"{pname}: {zig_type(ctype, arr)}")
    return f"pub extern fn {name}({', '.join(zargs)}) {zig_type(ret, is_return=True)};"


def generated_block() -> str:
    with open(HEADER) as fh:
        protos = prototypes(fh.read())
    return "\n".join([BEGIN] + [zig_decl(*p) for p in protos] + [END])


def main() -> int:
    with open(ZIG) as fh:
        zig = fh.read()
    a, b = zig.index(BEGIN), zig.index(END) + len(END)
    new = zig[:a] + generated_block() + zig[b:]
    if "--check" in sys.argv:
        return 0 if new == zig else 1
    with open(ZIG, "w") as fh:
        fh.write(new)
    return 0


if __name__ == "__main__":
    sys.exit(main())
