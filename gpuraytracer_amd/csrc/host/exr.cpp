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

// ---- PIZ blocks -------------------------------------------------------------------------------------------------
// OpenEXR's PIZ coder works on the 16-bit words of a block (a FLOAT sample is two words, low word first), gathered
// channel by channel: the words are renumbered densely through a bitmap of the values in use, each channel plane is
// transformed by a two-dimensional Haar-like wavelet, and the result is written with a canonical Huffman code whose
// table is a list of 6-bit code lengths; one extra symbol (the largest one) means "repeat the last word k times",
// k in the next 8 bits. Decoding runs the three steps backwards. This follows the published format of the OpenEXR
// library (the reference reads its textures through tinyexr's LoadEXR, Scene.cpp:163, which accepts such files).

constexpr uint32_t PIZ_SYMBOLS = 65537;        // 65536 word values and the repeat symbol
constexpr int PIZ_MAX_CODE = 58;               // longest code; lengths 59..63 in the table are runs of unused symbols
constexpr int PIZ_PEEK = 14;                   // codes up to this length resolve through one table look-up

struct HuffmanCode {
    uint64_t first[PIZ_MAX_CODE + 1] = {};     // value of the first code of each length
    uint32_t count[PIZ_MAX_CODE + 1] = {};     // symbols of each length
    uint32_t start[PIZ_MAX_CODE + 2] = {};     // where a length's symbols begin in `symbols`
    std::vector<uint32_t> symbols;             // by length, then by value: the order codes are handed out in
    std::vector<uint32_t> peek;                // 2^PIZ_PEEK entries: symbol << 6 | length, 0 = longer code or none
};

// Reads the table of code lengths for symbols lo..hi (bit stream, most significant bit first) and derives the codes:
// the longest codes start at 0, each shorter length continues at half the value the longer one ended at, and symbols
// of one length take consecutive codes in ascending order. Returns the bytes the table took, 0 on error.
size_t piz_read_code(const uint8_t *p, size_t n, uint32_t lo, uint32_t hi, HuffmanCode &hc) {
    std::vector<uint8_t> len(PIZ_SYMBOLS, 0);
    size_t at = 0;
    uint64_t acc = 0;
    int have = 0;
    auto bits = [&](int k, uint32_t &v) {
        while (have < k) {
            if (at >= n) return false;
            acc = (acc << 8) | p[at++];
            have += 8;
        }
        have -= k;
        v = (uint32_t)(acc >> have) & ((1u << k) - 1u);
        return true;
    };
    for (uint32_t s = lo; s <= hi;) {
        uint32_t l;
        if (!bits(6, l)) return 0;
        if (l <= (uint32_t)PIZ_MAX_CODE) { len[s++] = (uint8_t)l; continue; }
        uint32_t run = l - 57;                                   // 59..62: 2..5 unused symbols
        if (l == 63) { if (!bits(8, run)) return 0; run += 6; }  // 63: 6..261 of them, count in the next byte
        if ((uint64_t)s + run > (uint64_t)hi + 1) return 0;
        s += run;
    }
    for (uint32_t s = 0; s < PIZ_SYMBOLS; s++) hc.count[len[s]]++;
    hc.count[0] = 0;
    uint64_t next = 0;
    for (int l = PIZ_MAX_CODE; l >= 1; l--) {
        hc.first[l] = next;
        next = (next + hc.count[l]) >> 1;
    }
    for (int l = 1; l <= PIZ_MAX_CODE; l++) {
        hc.start[l + 1] = hc.start[l] + hc.count[l];
        if (hc.count[l] && ((hc.first[l] + hc.count[l] - 1) >> l)) return 0;   // more codes than the length has
    }
    hc.symbols.resize(hc.start[PIZ_MAX_CODE + 1]);
    {
        uint32_t fill[PIZ_MAX_CODE + 2];
        memcpy(fill, hc.start, sizeof(fill));
        for (uint32_t s = 0; s < PIZ_SYMBOLS; s++)
            if (len[s]) hc.symbols[fill[len[s]]++] = s;
    }
    hc.peek.assign((size_t)1 << PIZ_PEEK, 0u);
    for (int l = 1; l <= PIZ_PEEK; l++)
        for (uint32_t k = 0; k < hc.count[l]; k++) {
            const uint32_t entry = (hc.symbols[hc.start[l] + k] << 6) | (uint32_t)l;
            const size_t from = (size_t)(hc.first[l] + k) << (PIZ_PEEK - l);
            std::fill(hc.peek.begin() + from, hc.peek.begin() + from + ((size_t)1 << (PIZ_PEEK - l)), entry);
        }
    return at;
}

// The Huffman layer: a 20-byte header (smallest and largest symbol in the table, the table's byte length, the number
// of code bits, a reserved word), the table, the codes.
const char *piz_huffman_decode(const uint8_t *src, size_t n, uint16_t *out, size_t n_out) {
    if (n == 0) return n_out ? "PIZ: no data" : nullptr;
    if (n < 20) return "PIZ: truncated Huffman header";
    uint32_t lo, hi, nbits32;
    memcpy(&lo, src, 4); memcpy(&hi, src + 4, 4); memcpy(&nbits32, src + 12, 4);
    if (lo >= PIZ_SYMBOLS || hi >= PIZ_SYMBOLS) return "PIZ: bad symbol range";
    HuffmanCode hc;
    const size_t table = piz_read_code(src + 20, n - 20, lo, hi, hc);
    if (!table && lo <= hi) return "PIZ: bad code table";
    const uint8_t *d = src + 20 + table;
    const uint64_t avail = (uint64_t)(n - 20 - table) * 8, nbits = nbits32;
    if (nbits > avail) return "PIZ: fewer code bits than announced";
    const size_t nbytes = (size_t)((nbits + 7) / 8);
    auto window = [&](uint64_t pos, int k) {        // k <= 16 bits from bit `pos` on, zeros past the end
        const size_t b = (size_t)(pos >> 3);
        uint32_t w = 0;
        for (size_t i = 0; i < 3; i++) w = (w << 8) | (b + i < nbytes ? d[b + i] : 0u);
        return (w >> (24 - k - (int)(pos & 7))) & ((1u << k) - 1u);
    };
    const uint32_t repeat = hi;
    uint64_t pos = 0;
    size_t o = 0;
    while (pos < nbits) {
        uint32_t sym = 0;
        int l;
        const uint32_t head = window(pos, PIZ_PEEK), e = hc.peek[head];
        if (e) { sym = e >> 6; l = (int)(e & 63u); }
        else {
            uint64_t code = head;
            bool found = false;
            for (l = PIZ_PEEK + 1; l <= PIZ_MAX_CODE && pos + (uint64_t)l <= nbits; l++) {
                code = (code << 1) | window(pos + (uint64_t)l - 1, 1);
                if (code >= hc.first[l] && code - hc.first[l] < hc.count[l]) {
                    sym = hc.symbols[hc.start[l] + (size_t)(code - hc.first[l])];
                    found = true;
                    break;
                }
            }
            if (!found) return "PIZ: bad code";
        }
        if (pos + (uint64_t)l > nbits) return "PIZ: code runs past the end";
        pos += (uint64_t)l;
        if (sym == repeat) {
            if (pos + 8 > nbits) return "PIZ: repeat count runs past the end";
            const uint32_t k = window(pos, 8);
            pos += 8;
            if (o == 0 || o + k > n_out) return "PIZ: bad repeat";
            for (uint32_t i = 0; i < k; i++, o++) out[o] = out[o - 1];
        } else {
            if (o >= n_out) return "PIZ: more words than the block holds";
            out[o++] = (uint16_t)sym;
        }
    }
    return o == n_out ? nullptr : "PIZ: fewer words than the block holds";
}

// One wavelet step backwards: (average, difference) -> the two values. With fewer than 2^14 values in use the words
// are treated as signed and nothing can overflow; otherwise the arithmetic is modulo 2^16 around an offset of 2^15.
inline void piz_unstep(bool small, uint16_t avg, uint16_t diff, uint16_t &a, uint16_t &b) {
    if (small) {
        const int d = (int16_t)diff, x = (int16_t)avg + (d & 1) + (d >> 1);
        a = (uint16_t)x;
        b = (uint16_t)(x - d);
    } else {
        const int bb = ((int)avg - ((int)diff >> 1)) & 0xFFFF;
        a = (uint16_t)(((int)diff + bb - 0x8000) & 0xFFFF);
        b = (uint16_t)bb;
    }
}

// Undoes the wavelet transform of one plane of nx x ny words (ox, oy: distance between neighbours in x and in y),
// from the coarsest level down: each level combines words `half` apart inside cells `cell` = 2 * half wide, 2 x 2 at a
// time, and by pairs along the last column / row when the plane's size has that bit set.
void piz_wavelet_decode(uint16_t *w, int nx, int ox, int ny, int oy, uint32_t max_value) {
    const bool small = max_value < (1u << 14);
    const int n = std::min(nx, ny);
    int cell = 1;
    while (cell * 2 <= n) cell *= 2;
    for (int half = cell / 2; half >= 1; cell = half, half /= 2) {
        const long dx = (long)ox * half, dy = (long)oy * half;
        int y = 0;
        for (; y + cell <= ny; y += cell) {
            int x = 0;
            for (; x + cell <= nx; x += cell) {
                uint16_t *p00 = w + (long)y * oy + (long)x * ox, *p01 = p00 + dx, *p10 = p00 + dy, *p11 = p10 + dx;
                uint16_t t00, t01, t10, t11;
                piz_unstep(small, *p00, *p10, t00, t10);
                piz_unstep(small, *p01, *p11, t01, t11);
                piz_unstep(small, t00, t01, *p00, *p01);
                piz_unstep(small, t10, t11, *p10, *p11);
            }
            if (nx & half) {
                uint16_t *p00 = w + (long)y * oy + (long)x * ox, *p10 = p00 + dy, a, b;
                piz_unstep(small, *p00, *p10, a, b);
                *p00 = a; *p10 = b;
            }
        }
        if (ny & half)
            for (int x = 0; x + cell <= nx; x += cell) {
                uint16_t *p00 = w + (long)y * oy + (long)x * ox, *p01 = p00 + dx, a, b;
                piz_unstep(small, *p00, *p01, a, b);
                *p00 = a; *p01 = b;
            }
    }
}

// A whole PIZ block of cols x rows pixels into `out` (the uncompressed layout: row by row, channel by channel).
const char *piz_decode(const uint8_t *src, size_t size, const std::vector<Channel> &chans, int cols, int rows,
                       uint8_t *out, size_t want) {
    if (size < 4) return "PIZ: truncated block";
    uint16_t lo, hi;
    memcpy(&lo, src, 2); memcpy(&hi, src + 2, 2);
    size_t at = 4;
    std::vector<uint8_t> used(8192, 0);                 // bit v set: word value v occurs; value 0 is always in use
    if (hi >= used.size()) return "PIZ: bad bitmap range";
    if (lo <= hi) {
        const size_t len = (size_t)hi - lo + 1;
        if (size - at < len) return "PIZ: truncated bitmap";
        memcpy(used.data() + lo, src + at, len);
        at += len;
    }
    std::vector<uint16_t> value_of(65536, 0);           // dense number -> word value
    uint32_t n_values = 0;
    for (uint32_t v = 0; v < 65536; v++)
        if (v == 0 || (used[v >> 3] & (1u << (v & 7)))) value_of[n_values++] = (uint16_t)v;
    if (size - at < 4) return "PIZ: truncated block";
    int32_t len;
    memcpy(&len, src + at, 4);
    at += 4;
    if (len < 0 || (size_t)len > size - at) return "PIZ: bad length";
    const size_t words = want / 2;
    std::vector<uint16_t> w(words);
    if (const char *what = piz_huffman_decode(src + at, (size_t)len, w.data(), words)) return what;
    size_t plane = 0;
    for (const Channel &c : chans) {
        const int per = c.type == 1 ? 1 : 2;
        for (int j = 0; j < per; j++) piz_wavelet_decode(w.data() + plane + j, cols, per, rows, cols * per, n_values - 1);
        plane += (size_t)cols * rows * per;
    }
    for (uint16_t &v : w) v = value_of[v];
    std::vector<size_t> row_of(chans.size());
    plane = 0;
    for (size_t c = 0; c < chans.size(); c++) {
        row_of[c] = plane;
        plane += (size_t)cols * rows * (chans[c].type == 1 ? 1 : 2);
    }
    for (int y = 0; y < rows; y++)
        for (size_t c = 0; c < chans.size(); c++) {
            const size_t n = (size_t)cols * (chans[c].type == 1 ? 1 : 2);
            memcpy(out, w.data() + row_of[c], n * 2);
            out += n * 2;
            row_of[c] += n;
        }
    return nullptr;
}

}  // namespace

bool writeEXR(const std::string &path, int width, int height, int channels, const float *pixels, std::string *err,
              int compression) {
    if (width <= 0 || height <= 0 || (channels != 3 && channels != 4) || !pixels || (compression != 0 && compression != 3)) {
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
    put_attr(b, "compression", "compression", std::vector<uint8_t>{(uint8_t)compression});
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

    // Blocks of 1 line (no compression) or 16 lines (ZIP); the table of their offsets comes first and is filled in
    // once the sizes are known.
    const int lines_per_block = compression == 3 ? 16 : 1;
    const int blocks = (height + lines_per_block - 1) / lines_per_block;
    const uint64_t table_at = b.size();
    std::vector<uint64_t> offsets(blocks);
    b.resize(b.size() + 8ull * blocks, 0);

    std::ofstream f(path, std::ios::binary);
    if (!f) { if (err) *err = "writeEXR: cannot open " + path; return false; }
    f.write((const char *)b.data(), (std::streamsize)b.size());
    uint64_t offset = b.size();
    std::vector<uint8_t> raw, mixed, packed, head;
    // channel order in the file is alphabetical: (A,) B, G, R; source pixels are RGB(A)
    const int src_of[4] = {3, 2, 1, 0};
    for (int k = 0; k < blocks; k++) {
        const int y0 = k * lines_per_block, y1 = std::min(height, y0 + lines_per_block);
        raw.clear();
        for (int y = y0; y < y1; y++)
            for (int c = (channels == 4 ? 0 : 1); c < 4; c++)
                for (int x = 0; x < width; x++) put_f32(raw, pixels[((size_t)y * width + x) * channels + src_of[c]]);
        const std::vector<uint8_t> *data = &raw;
        if (compression == 3) {
            // even bytes, then odd bytes; each byte as its difference from the one before (+128); deflate
            const size_t n = raw.size(), half = (n + 1) / 2;
            mixed.resize(n);
            for (size_t i = 0; i < n; i++) mixed[(i & 1) ? half + i / 2 : i / 2] = raw[i];
            for (size_t i = n - 1; i > 0; i--) mixed[i] = (uint8_t)(mixed[i] - mixed[i - 1] + 128);
            uLongf got = compressBound((uLong)n);
            packed.resize(got);
            if (compress(packed.data(), &got, mixed.data(), (uLong)n) != Z_OK) { if (err) *err = "writeEXR: zlib failed"; return false; }
            packed.resize(got);
            if (packed.size() < raw.size()) data = &packed;        // a block that does not shrink is stored as it is
        }
        head.clear();
        put_u32(head, (uint32_t)y0);
        put_u32(head, (uint32_t)data->size());
        f.write((const char *)head.data(), (std::streamsize)head.size());
        f.write((const char *)data->data(), (std::streamsize)data->size());
        offsets[k] = offset;
        offset += 8 + data->size();
    }
    head.clear();
    for (uint64_t o : offsets) put_u64(head, o);
    f.seekp((std::streamoff)table_at);
    f.write((const char *)head.data(), (std::streamsize)head.size());
    f.flush();
    if (!f) { if (err) *err = "writeEXR: write failed " + path; return false; }
    return true;
}

static bool readEXRBody(const std::string &path, int &width, int &height, std::vector<float> &rgba, std::string *err);

// No exception leaves this function: a damaged file (or one too large for the memory at hand) is a `false` with a message,
// also for the callers that are not behind the C entry points (Scene::loadMaterial, Scene::loadEnvironmentTexture).
bool readEXR(const std::string &path, int &width, int &height, std::vector<float> &rgba, std::string *err) {
    try {
        return readEXRBody(path, width, height, rgba, err);
    } catch (const std::exception &e) {
        if (err) *err = "readEXR: " + std::string(e.what()) + " (" + path + ")";
    } catch (...) {
        if (err) *err = "readEXR: failed (" + path + ")";
    }
    rgba.clear();
    return false;
}

static bool readEXRBody(const std::string &path, int &width, int &height, std::vector<float> &rgba, std::string *err) {
    auto fail = [&](const std::string &m) { if (err) *err = "readEXR: " + m + " (" + path + ")"; return false; };
    std::ifstream f(path, std::ios::binary);
    if (!f) return fail("cannot open");
    std::vector<uint8_t> data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    Reader r{data.data(), data.data() + data.size()};
    if (r.u32() != 20000630u) return fail("not an OpenEXR file");
    const uint32_t version = r.u32();
    // flags: 0x200 tiled, 0x400 attribute names of up to 255 characters, 0x800 deep data, 0x1000 several parts
    if ((version & 0xFFu) != 2 || (version & 0x1800u)) return fail("only single-part files without deep data are supported");
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
                if (!a.ok) return fail("truncated channel list");
                if (xs != 1 || ys != 1) return fail("subsampled channels are not supported");
                chans.push_back(c);
            }
            if (!a.ok) return fail("truncated channel list");
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
    if (compression < 0 || compression > 4) return fail("compression must be NONE, RLE, ZIPS, ZIP or PIZ");
    if (tiled && (tile_w == 0 || tile_h == 0)) return fail("tiled file without a tile description");
    if (tiled && (tile_w > 65536u || tile_h > 65536u)) return fail("tile size above 65536");
    if (tiled && (tile_mode & 0xFu) > 2) return fail("bad level mode in the tile description");
    const int64_t wide = (int64_t)xmax - xmin + 1, high = (int64_t)ymax - ymin + 1;
    if (wide > 65536 || high > 65536 || wide * high > ((int64_t)1 << 28)) return fail("data window above 2^28 pixels");
    width = (int)wide;
    height = (int)high;
    size_t pixel_bytes = 0;
    for (const Channel &c : chans) {
        if (c.type != 1 && c.type != 2) return fail("channel type must be HALF or FLOAT");
        pixel_bytes += c.type == 1 ? 2 : 4;
    }
    // As tinyexr's LoadEXR (Scene.cpp:81): a file with one channel is a grey image (its value goes to all four
    // components), any other file needs R, G and B, and A is 1 where the file has none.
    const bool grey = chans.size() == 1;
    if (!grey)
        for (const char *need : {"R", "G", "B"})
            if (std::none_of(chans.begin(), chans.end(), [&](const Channel &c) { return c.name == need; }))
                return fail(std::string("no channel ") + need);
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
        } else if (compression == 4) {
            if (const char *what = piz_decode(src, size, chans, cols, rows, raw.data(), want)) return what;
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
                if (dst >= 0 || grey)
                    for (int x = 0; x < cols; x++) {
                        float v;
                        if (c.type == 1) { uint16_t h; memcpy(&h, p + x * 2, 2); v = half_to_float(h); }
                        else memcpy(&v, p + (size_t)x * 4, 4);
                        float *px = &rgba[((size_t)(y0 + l) * width + (x0 + x)) * 4];
                        if (grey) px[0] = px[1] = px[2] = px[3] = v;
                        else px[dst] = v;
                    }
                p += esz * cols;
            }
        }
        return nullptr;
    };

    if (tiled) {
        const int nx = (width + (int)tile_w - 1) / (int)tile_w, ny = (height + (int)tile_h - 1) / (int)tile_h;
        // Files with mip-map or rip-map levels list the tiles of the full-resolution level (0, 0) first in the offset
        // table; only those are read.
        if ((size_t)(r.end - r.p) / 8 < (size_t)nx * ny) return fail("truncated offset table");
        std::vector<uint64_t> offsets((size_t)nx * ny);
        for (uint64_t &o : offsets) o = r.u64();
        if (!r.ok) return fail("truncated offset table");
        for (uint64_t off : offsets) {
            if (off > data.size() || data.size() - off < 20) return fail("bad tile offset");
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
    const int lines_per_block = compression == 4 ? 32 : compression == 3 ? 16 : 1;
    const int blocks = (height + lines_per_block - 1) / lines_per_block;
    if ((size_t)(r.end - r.p) / 8 < (size_t)blocks) return fail("truncated offset table");
    std::vector<uint64_t> offsets(blocks);
    for (int i = 0; i < blocks; i++) offsets[i] = r.u64();
    if (!r.ok) return fail("truncated offset table");
    for (int bidx = 0; bidx < blocks; bidx++) {
        if (offsets[bidx] > data.size() || data.size() - offsets[bidx] < 8) return fail("bad block offset");
        Reader br{data.data() + offsets[bidx], data.data() + data.size()};
        const int64_t line = (int64_t)(int32_t)br.u32() - ymin;
        const uint32_t size = br.u32();
        if (!br.ok || (size_t)(br.end - br.p) < size || line < 0 || line >= height) return fail("bad block");
        const int y0 = (int)line;
        const char *what = decode_block(br.p, size, 0, y0, width, std::min(lines_per_block, height - y0));
        if (what) return fail(what);
    }
    return true;
}

}  // namespace host
