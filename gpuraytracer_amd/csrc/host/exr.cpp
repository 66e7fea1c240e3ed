// exr.cpp - minimal OpenEXR scanline reader/writer, see exr.hpp.
#include "exr.hpp"

#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>

namespace host {
namespace {

void put_u32(std::vector<uint8_t> &b, uint32_t v) { for (int i = 0; i < 4; i++) b.push_back((uint8_t)(v >> (8 * i))); }
void put_u64(std::vector<uint8_t> &b, uint64_t v) { for (int i = 0; i < 8; i++) b.push_back((uint8_t)(v >> (8 * i))); }
void put_str(std::vector<uint8_t> &b, const char *s) { while (*s) b.push_back((uint8_t)*s++); b.push_back(0); }
void put_f32(std::vector<uint8_t> &b, float f) { uint32_t u; memcpy(&u, &f, 4); put_u32(b, u); }
void put_attr(std::vector<uint8_t> &b, const char *name, const char *type, const std::vector<uint8_t> &value) {
    put_str(b, name);
    put_str(b, type);
    put_u32(b, (uint32_t)value.size());
    b.insert(b.end(), value.begin(), value.end());
}

float half_to_float(uint16_t h) {
    const uint32_t s = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
    uint32_t bits;
    if (e == 0) {
        if (m == 0) bits = s;
        else {                       // subnormal half: renormalise
            int shift = 0;
            uint32_t mm = m;
            while (!(mm & 0x400u)) { mm <<= 1; shift++; }
            bits = s | ((uint32_t)(113 - shift) << 23) | ((mm & 0x3FFu) << 13);
        }
    } else if (e == 31) bits = s | 0x7F800000u | (m << 13);
    else bits = s | ((e + 112u) << 23) | (m << 13);
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

struct Reader {
    const uint8_t *p, *end;
    bool ok = true;
    uint32_t u32() { if (end - p < 4) { ok = false; return 0; } uint32_t v; memcpy(&v, p, 4); p += 4; return v; }
    uint64_t u64() { if (end - p < 8) { ok = false; return 0; } uint64_t v; memcpy(&v, p, 8); p += 8; return v; }
    std::string str() {
        std::string s;
        while (p < end && *p) s.push_back((char)*p++);
        if (p >= end) ok = false; else p++;
        return s;
    }
};

struct Channel { std::string name; int type; };

}  // namespace

bool writeEXR(const std::string &path, int width, int height, int channels, const float *pixels, std::string *err) {
    if (width <= 0 || height <= 0 || (channels != 3 && channels != 4) || !pixels) {
        if (err) *err = "writeEXR: bad arguments";
        return false;
    }
    std::vector<uint8_t> b;
    put_u32(b, 20000630u);   // magic
    put_u32(b, 2u);          // version 2, single-part scanline
    {
        std::vector<uint8_t> v;
        const char *names4[4] = {"A", "B", "G", "R"};
        for (int c = (channels == 4 ? 0 : 1); c < 4; c++) {
            put_str(v, names4[c]);
            put_u32(v, 2u);                     // FLOAT
            v.push_back(0); v.push_back(0); v.push_back(0); v.push_back(0);   // pLinear + reserved
            put_u32(v, 1u); put_u32(v, 1u);     // sampling
        }
        v.push_back(0);
        put_attr(b, "channels", "chlist", v);
    }
    put_attr(b, "compression", "compression", std::vector<uint8_t>{0});
    {
        std::vector<uint8_t> v;
        put_u32(v, 0); put_u32(v, 0); put_u32(v, (uint32_t)(width - 1)); put_u32(v, (uint32_t)(height - 1));
        put_attr(b, "dataWindow", "box2i", v);
        put_attr(b, "displayWindow", "box2i", v);
    }
    put_attr(b, "lineOrder", "lineOrder", std::vector<uint8_t>{0});
    { std::vector<uint8_t> v; put_f32(v, 1.0f); put_attr(b, "pixelAspectRatio", "float", v); }
    { std::vector<uint8_t> v; put_f32(v, 0.0f); put_f32(v, 0.0f); put_attr(b, "screenWindowCenter", "v2f", v); }
    { std::vector<uint8_t> v; put_f32(v, 1.0f); put_attr(b, "screenWindowWidth", "float", v); }
    b.push_back(0);          // end of header

    const uint64_t line_bytes = (uint64_t)width * channels * 4;
    const uint64_t table_at = b.size();
    uint64_t offset = table_at + 8ull * height;
    for (int y = 0; y < height; y++) { put_u64(b, offset); offset += 8 + line_bytes; }

    std::ofstream f(path, std::ios::binary);
    if (!f) { if (err) *err = "writeEXR: cannot open " + path; return false; }
    f.write((const char *)b.data(), (std::streamsize)b.size());
    std::vector<uint8_t> line;
    // channel order in the file is alphabetical: (A,) B, G, R; source pixels are RGB(A)
    const int src_of[4] = {3, 2, 1, 0};
    for (int y = 0; y < height; y++) {
        line.clear();
        put_u32(line, (uint32_t)y);
        put_u32(line, (uint32_t)line_bytes);
        for (int c = (channels == 4 ? 0 : 1); c < 4; c++)
            for (int x = 0; x < width; x++) put_f32(line, pixels[((size_t)y * width + x) * channels + src_of[c]]);
        f.write((const char *)line.data(), (std::streamsize)line.size());
    }
    if (!f) { if (err) *err = "writeEXR: write failed " + path; return false; }
    return true;
}

bool readEXR(const std::string &path, int &width, int &height, std::vector<float> &rgba, std::string *err) {
    auto fail = [&](const std::string &m) { if (err) *err = "readEXR: " + m + " (" + path + ")"; return false; };
    std::ifstream f(path, std::ios::binary);
    if (!f) return fail("cannot open");
    std::vector<uint8_t> data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    Reader r{data.data(), data.data() + data.size()};
    if (r.u32() != 20000630u) return fail("not an OpenEXR file");
    const uint32_t version = r.u32();
    if ((version & 0xFFu) != 2 || (version & 0x1C00u)) return fail("only single-part files without deep data are supported");
    const bool tiled = (version & 0x200u) != 0;
    std::vector<Channel> chans;
    int compression = -1, xmin = 0, ymin = 0, xmax = -1, ymax = -1;
    uint32_t tile_w = 0, tile_h = 0, tile_mode = 0;
    while (r.ok) {
        const std::string name = r.str();
        if (name.empty()) break;
        const std::string type = r.str();
        const uint32_t size = r.u32();
        if (!r.ok || (size_t)(r.end - r.p) < size) return fail("truncated header");
        Reader a{r.p, r.p + size};
        if (name == "channels") {
            while (a.ok && a.p < a.end && *a.p) {
                Channel c;
                c.name = a.str();
                c.type = (int)a.u32();
                a.u32();                        // pLinear + reserved
                const uint32_t xs = a.u32(), ys = a.u32();
                if (xs != 1 || ys != 1) return fail("subsampled channels are not supported");
                chans.push_back(c);
            }
        } else if (name == "compression") {
            compression = size ? (int)*a.p : -1;
        } else if (name == "dataWindow") {
            xmin = (int)a.u32(); ymin = (int)a.u32(); xmax = (int)a.u32(); ymax = (int)a.u32();
        } else if (name == "tiles") {
            tile_w = a.u32(); tile_h = a.u32();
            tile_mode = size > 8 ? (uint32_t)a.p[0] : 0u;
        }
        r.p += size;
    }
    if (!r.ok || chans.empty() || xmax < xmin || ymax < ymin) return fail("bad header");
    if (compression < 0 || compression > 3) return fail("compression must be NONE, RLE, ZIPS or ZIP");
    if (tiled && (tile_w == 0 || tile_h == 0)) return fail("tiled file without a tile description");
    if (tiled && (tile_mode & 0xFu) != 0) return fail("only tiled files with one level are supported");
    width = xmax - xmin + 1;
    height = ymax - ymin + 1;
    size_t pixel_bytes = 0;
    for (const Channel &c : chans) {
        if (c.type != 1 && c.type != 2) return fail("channel type must be HALF or FLOAT");
        pixel_bytes += c.type == 1 ? 2 : 4;
    }
    rgba.assign((size_t)width * height * 4, 0.0f);
    for (size_t i = 0; i < (size_t)width * height; i++) rgba[i * 4 + 3] = 1.0f;

    // A block is a run of scanlines over the full width, or one tile: rows [y0, y0 + rows) x columns [x0, x0 + cols),
    // stored row by row, channel by channel within a row.
    std::vector<uint8_t> raw, tmp;
    auto decode_block = [&](const uint8_t *src, uint32_t size, int x0, int y0, int cols, int rows) -> const char * {
        const size_t want = pixel_bytes * (size_t)cols * rows;
        raw.resize(want);
        if (compression == 0 || size == want) {
            if (size != want) return "bad block size";
            memcpy(raw.data(), src, want);
        } else {
            tmp.resize(want);
            if (compression == 1) {
                // run-length code: a count byte n, then n < 0: -n bytes follow as they are, n >= 0: the next byte n + 1 times
                size_t in = 0, out = 0;
                while (in < size) {
                    const int n = (int8_t)src[in++];
                    if (n < 0) {
                        const size_t len = (size_t)-n;
                        if (in + len > size || out + len > want) return "run-length error";
                        memcpy(tmp.data() + out, src + in, len);
                        in += len; out += len;
                    } else {
                        const size_t len = (size_t)n + 1;
                        if (in >= size || out + len > want) return "run-length error";
                        memset(tmp.data() + out, src[in++], len);
                        out += len;
                    }
                }
                if (out != want) return "run-length error";
            } else {
                uLongf got = (uLongf)want;
                if (uncompress(tmp.data(), &got, src, size) != Z_OK || got != want) return "zlib error";
            }
            for (size_t i = 1; i < want; i++) tmp[i] = (uint8_t)(tmp[i - 1] + tmp[i] - 128);     // undo the predictor
            const size_t half = (want + 1) / 2;
            for (size_t i = 0; i < want; i++) raw[i] = (i & 1) ? tmp[half + i / 2] : tmp[i / 2];  // undo the interleave
        }
        const uint8_t *p = raw.data();
        for (int l = 0; l < rows; l++) {
            for (const Channel &c : chans) {
                int dst = -1;
                if (c.name == "R") dst = 0; else if (c.name == "G") dst = 1; else if (c.name == "B") dst = 2;
                else if (c.name == "A") dst = 3;
                const size_t esz = c.type == 1 ? 2 : 4;
                if (dst >= 0)
                    for (int x = 0; x < cols; x++) {
                        float v;
                        if (c.type == 1) { uint16_t h; memcpy(&h, p + x * 2, 2); v = half_to_float(h); }
                        else memcpy(&v, p + (size_t)x * 4, 4);
                        rgba[((size_t)(y0 + l) * width + (x0 + x)) * 4 + dst] = v;
                    }
                p += esz * cols;
            }
        }
        return nullptr;
    };

    if (tiled) {
        const int nx = (width + (int)tile_w - 1) / (int)tile_w, ny = (height + (int)tile_h - 1) / (int)tile_h;
        std::vector<uint64_t> offsets((size_t)nx * ny);
        for (uint64_t &o : offsets) o = r.u64();
        if (!r.ok) return fail("truncated offset table");
        for (uint64_t off : offsets) {
            if (off + 20 > data.size()) return fail("bad tile offset");
            Reader br{data.data() + off, data.data() + data.size()};
            const int tx = (int)br.u32(), ty = (int)br.u32();
            const uint32_t lx = br.u32(), ly = br.u32(), size = br.u32();
            if (!br.ok || (size_t)(br.end - br.p) < size || tx < 0 || ty < 0 || tx >= nx || ty >= ny || lx != 0 || ly != 0)
                return fail("bad tile");
            const int x0 = tx * (int)tile_w, y0 = ty * (int)tile_h;
            const char *what = decode_block(br.p, size, x0, y0, std::min((int)tile_w, width - x0), std::min((int)tile_h, height - y0));
            if (what) return fail(what);
        }
        return true;
    }
    const int lines_per_block = compression == 3 ? 16 : 1;
    const int blocks = (height + lines_per_block - 1) / lines_per_block;
    std::vector<uint64_t> offsets(blocks);
    for (int i = 0; i < blocks; i++) offsets[i] = r.u64();
    if (!r.ok) return fail("truncated offset table");
    for (int bidx = 0; bidx < blocks; bidx++) {
        if (offsets[bidx] + 8 > data.size()) return fail("bad block offset");
        Reader br{data.data() + offsets[bidx], data.data() + data.size()};
        const int y0 = (int)br.u32() - ymin;
        const uint32_t size = br.u32();
        if (!br.ok || (size_t)(br.end - br.p) < size || y0 < 0 || y0 >= height) return fail("bad block");
        const char *what = decode_block(br.p, size, 0, y0, width, std::min(lines_per_block, height - y0));
        if (what) return fail(what);
    }
    return true;
}

}  // namespace host
