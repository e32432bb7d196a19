// host_vox.cpp — MagicaVoxel .vox parsing for the scene side of the traversal path.
// Semantics of the reference loader (src/modules/voxel_rt/vox/loader.zig, types.zig) with bounds
// checking; see include/vrt_hip.h.
#include <cstring>
#include <new>
#include <vector>
#include "host_brick_grid.hpp"

struct vrt_vox {
    int32_t num_models = 1; // Chunk.Pack.num_models
    std::vector<int32_t> sizes;                      // 3 per model (Chunk.Size)
    std::vector<std::vector<vrt_vox_xyzi>> voxels;   // per model (Chunk.XyziElement)
    vrt_vox_rgba palette[256];
};

namespace {

struct Reader {
    const uint8_t *p;
    uint64_t n;
    bool has(uint64_t pos, uint64_t len) const { return pos <= n && len <= n - pos; }
    int32_t i32(uint64_t pos) const { // loader.zig parseI32: little-endian i32 at pos
        uint32_t v;
        std::memcpy(&v, p + pos, 4);
        return (int32_t)v;
    }
    bool tag(uint64_t pos, const char *t) const { return has(pos, 4) && std::memcmp(p + pos, t, 4) == 0; }
};

// The MagicaVoxel default palette (0xAABBGGRR words, byte order r,g,b,a in memory): entry 0 is
// empty, 1..215 walk the 6x6x6 cube {ff,cc,99,66,33,00} with blue fastest, then green, then red
// (black left out), followed by ten-step ramps of red, green, blue and grey.
void default_palette(vrt_vox_rgba *out) {
    static const uint8_t lv[6] = {0xff, 0xcc, 0x99, 0x66, 0x33, 0x00};
    static const uint8_t ramp[10] = {0xee, 0xdd, 0xbb, 0xaa, 0x88, 0x77, 0x55, 0x44, 0x22, 0x11};
    out[0] = vrt_vox_rgba{0, 0, 0, 0};
    uint32_t i = 1;
    for (uint32_t n = 0; n < 215; n++, i++) out[i] = vrt_vox_rgba{lv[n / 36], lv[(n / 6) % 6], lv[n % 6], 0xff};
    for (uint32_t k = 0; k < 10; k++, i++) out[i] = vrt_vox_rgba{ramp[k], 0, 0, 0xff};
    for (uint32_t k = 0; k < 10; k++, i++) out[i] = vrt_vox_rgba{0, ramp[k], 0, 0xff};
    for (uint32_t k = 0; k < 10; k++, i++) out[i] = vrt_vox_rgba{0, 0, ramp[k], 0xff};
    for (uint32_t k = 0; k < 10; k++, i++) out[i] = vrt_vox_rgba{ramp[k], ramp[k], ramp[k], 0xff};
}

} // namespace

extern "C" {

int vrt_vox_validate_header(const void *buffer, uint64_t nbytes) { // loader.zig:231-245
    if (!buffer || nbytes < 12) return VRT_VOX_E_INVALID_FILE_CONTENT;
    const uint8_t *b = static_cast<const uint8_t *>(buffer);
    if (std::memcmp(b, "VOX ", 4) != 0) return VRT_VOX_E_INVALID_ID;
    if (b[4] != 150) return VRT_VOX_E_UNEXPECTED_VERSION;
    if (std::memcmp(b + 8, "MAIN", 4) != 0) return VRT_VOX_E_INVALID_FILE_CONTENT;
    return VRT_OK;
}

int vrt_vox_parse(const void *buffer, uint64_t nbytes, int strict, vrt_vox **out) { // loader.zig:41-229
    if (!out) return VRT_E_INVALID_ARG;
    *out = nullptr;
    if (!buffer) return VRT_E_INVALID_ARG;
    if (strict) {
        const int rc = vrt_vox_validate_header(buffer, nbytes);
        if (rc != VRT_OK) return rc;
    }
    Reader r{static_cast<const uint8_t *>(buffer), nbytes};
    const uint64_t chunk_stride = 12; // id + chunk size + child size
    uint64_t pos = 8 + chunk_stride;  // skip the header and the MAIN chunk
    if (!r.has(pos, 1)) return VRT_VOX_E_INVALID_FILE_CONTENT;

    vrt_vox *v = new (std::nothrow) vrt_vox();
    if (!v) return VRT_E_OOM;
    int rc = VRT_OK;
    try {
        if (r.p[pos] == 'P') { // PACK, loader.zig:62-76
            pos += chunk_stride;
            if (!r.has(pos, 4)) throw (int)VRT_VOX_E_INVALID_FILE_CONTENT;
            v->num_models = r.i32(pos);
            pos += 4;
        } else {
            v->num_models = 1;
        }
        if (v->num_models < 0 || (uint64_t)v->num_models > nbytes) throw (int)VRT_VOX_E_INVALID_FILE_CONTENT;
        v->sizes.resize((size_t)v->num_models * 3);
        v->voxels.resize((size_t)v->num_models);
        for (int32_t model = 0; model < v->num_models; model++) {
            if (strict && !r.tag(pos, "SIZE")) throw (int)VRT_VOX_E_EXPECTED_SIZE_HEADER;
            pos += chunk_stride;
            if (!r.has(pos, 12)) throw (int)VRT_VOX_E_INVALID_FILE_CONTENT;
            for (int k = 0; k < 3; k++) v->sizes[(size_t)model * 3 + k] = r.i32(pos + 4 * k);
            pos += 12;
            if (strict && !r.tag(pos, "XYZI")) throw (int)VRT_VOX_E_EXPECTED_XYZI_HEADER;
            pos += chunk_stride;
            if (!r.has(pos, 4)) throw (int)VRT_VOX_E_INVALID_FILE_CONTENT;
            const int32_t count = r.i32(pos);
            pos += 4;
            if (count < 0 || !r.has(pos, (uint64_t)count * 4u)) throw (int)VRT_VOX_E_INVALID_FILE_CONTENT;
            std::vector<vrt_vox_xyzi> &dst = v->voxels[(size_t)model];
            dst.resize((size_t)count);
            if (count) std::memcpy(dst.data(), r.p + pos, (size_t)count * 4u);
            pos += (uint64_t)count * 4u;
        }
        bool rgba_set = false;
        while (pos < nbytes) { // loader.zig:157-196
            if (r.p[pos] == 'R') {
                if (strict && !r.tag(pos, "RGBA")) throw (int)VRT_VOX_E_EXPECTED_RGBA_HEADER;
                pos += chunk_stride;
                // entry 0 is (0,0,0,1); the reference then reads 254 colours into entries 1..254
                // (`while (i < 255)`, loader.zig:175) and leaves entry 255 unset — it is zero here
                if (!r.has(pos, 254u * 4u)) throw (int)VRT_VOX_E_INVALID_FILE_CONTENT;
                std::memset(v->palette, 0, sizeof v->palette);
                v->palette[0] = vrt_vox_rgba{0, 0, 0, 1};
                std::memcpy(&v->palette[1], r.p + pos, 254u * 4u);
                pos += 254u * 4u;
                rgba_set = true;
            } else {
                pos += 4; // skip bytes
            }
        }
        if (!rgba_set) default_palette(v->palette);
    } catch (int code) {
        rc = code;
    } catch (const std::bad_alloc &) {
        rc = VRT_E_OOM;
    }
    if (rc != VRT_OK) {
        delete v;
        return rc;
    }
    *out = v;
    return VRT_OK;
}

void vrt_vox_destroy(vrt_vox *v) { delete v; }

uint32_t vrt_vox_num_models(const vrt_vox *v) { return v ? (uint32_t)v->num_models : 0u; }

int vrt_vox_model_size(const vrt_vox *v, uint32_t model, int32_t size_xyz[3]) {
    if (!v || !size_xyz || model >= (uint32_t)v->num_models) return VRT_E_INVALID_ARG;
    for (int k = 0; k < 3; k++) size_xyz[k] = v->sizes[(size_t)model * 3 + k];
    return VRT_OK;
}

const vrt_vox_xyzi *vrt_vox_model_voxels(const vrt_vox *v, uint32_t model, uint64_t *count) {
    if (count) *count = 0;
    if (!v || model >= (uint32_t)v->num_models) return nullptr;
    if (count) *count = v->voxels[model].size();
    return v->voxels[model].data();
}

const vrt_vox_rgba *vrt_vox_palette(const vrt_vox *v) { return v ? v->palette : nullptr; }

int vrt_vox_materials(const vrt_vox *v, vrt_material *out, uint32_t count) { // src/main.zig:93-106
    if (!v || !out || count > 256u) return VRT_E_INVALID_ARG;
    for (uint32_t i = 0; i < count; i++) {
        const vrt_vox_rgba c = v->palette[i];
        const bool dielectric = ((float)c.a / 255.0f) < 0.8f;
        out[i].type = dielectric ? 2u : 0u;
        out[i].albedo_r = (float)c.r / 255.0f;
        out[i].albedo_g = (float)c.g / 255.0f;
        out[i].albedo_b = (float)c.b / 255.0f;
        out[i].type_data = dielectric ? 1.52f : 0.0f;
    }
    return VRT_OK;
}

int vrt_vox_insert(vrt_grid *gh, const vrt_vox *v, uint32_t model, uint32_t off_x, uint32_t off_y, uint32_t off_z, uint32_t material_offset) {
    if (!gh || !v || model >= (uint32_t)v->num_models) return VRT_E_INVALID_ARG;
    vrt::BrickGrid *g = reinterpret_cast<vrt::BrickGrid *>(gh);
    for (const vrt_vox_xyzi &e : v->voxels[model]) { // src/main.zig:109-117: (x, z, y) — .vox is z-up
        const int rc = g->insertUnlocked((uint64_t)e.x + off_x, (uint64_t)e.z + off_y, (uint64_t)e.y + off_z,
                                         (uint8_t)(e.color_index + material_offset));
        if (rc != VRT_OK) return rc;
    }
    return VRT_OK;
}

} // extern "C"
