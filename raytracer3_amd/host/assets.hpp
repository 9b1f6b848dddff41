// assets.hpp -- native asset ingestion for the path tracer's host side (C++17 + zlib, no other dependency).
//
// Mirrors the reference's loaders (DerEchteKarsten/RayTracer3 `src/assets/mod.rs`):
//   GltfMeshLoader (:179-200, :207-252)  -> rt3::assets::load_glb      all meshes x primitives, node transforms baked,
//                                                                       u8/u16/u32 indices, KHR_materials_emissive_strength,
//                                                                       embedded / external PNG / baseline-JPEG base-colour textures
//   Mesh / Material / Vertex (:52-59, :118-133) -> rt3::assets::Mesh   flattened like world/mod.rs:103-125 uploads it
//   MeshSaver / bincode CONFIG (:135-137, :299-314) -> read_processed_mesh
//   skybox EXR (main.rs:94, commented)   -> read_exr                    scanline, NONE / RLE / ZIPS / ZIP / PIZ, HALF / FLOAT / UINT
// and pushes the result through the C ABI (upload()).  Same results as raytracer3_amd/assets.py; tests/test_host_assets.py
// compares the two loaders array by array.  Arithmetic-coded / 12-bit / CMYK JPEG and EXR PXR24/B44/DWA are not decoded (reported as errors).
#pragma once
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rt3.h"

namespace rt3::assets {

inline std::vector<uint8_t> read_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error(path + ": cannot open");
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

// ------------------------------------------------------------------------------------------------ JSON (RFC 8259 subset: no \u surrogates)
struct Json {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;

    const Json* find(const std::string& k) const {
        if (kind != Obj) return nullptr;
        for (auto& kv : obj)
            if (kv.first == k) return &kv.second;
        return nullptr;
    }
    bool has(const std::string& k) const { return find(k) != nullptr; }
    const Json& at(const std::string& k) const {
        const Json* j = find(k);
        if (!j) throw std::runtime_error("glTF: missing key '" + k + "'");
        return *j;
    }
    const Json& at(size_t i) const {
        if (kind != Arr || i >= arr.size()) throw std::runtime_error("glTF: array index out of range");
        return arr[i];
    }
    double number(double dflt) const { return kind == Num ? num : dflt; }
    int64_t integer(int64_t dflt) const { return kind == Num ? (int64_t)num : dflt; }
    size_t size() const { return kind == Arr ? arr.size() : (kind == Obj ? obj.size() : 0); }
};

class JsonParser {
  public:
    JsonParser(const char* p, size_t n) : p_(p), e_(p + n) {}
    Json parse() {
        Json j = value();
        ws();
        return j;
    }

  private:
    const char *p_, *e_;
    void ws() {
        while (p_ < e_ && (*p_ == ' ' || *p_ == '\n' || *p_ == '\r' || *p_ == '\t')) ++p_;
    }
    [[noreturn]] void fail(const char* what) { throw std::runtime_error(std::string("JSON: ") + what); }
    Json value() {
        ws();
        if (p_ >= e_) fail("unexpected end");
        Json j;
        char c = *p_;
        if (c == '{') {
            ++p_;
            j.kind = Json::Obj;
            ws();
            if (p_ < e_ && *p_ == '}') { ++p_; return j; }
            for (;;) {
                ws();
                std::string k = string();
                ws();
                if (p_ >= e_ || *p_ != ':') fail("':' expected");
                ++p_;
                j.obj.emplace_back(std::move(k), value());
                ws();
                if (p_ < e_ && *p_ == ',') { ++p_; continue; }
                if (p_ < e_ && *p_ == '}') { ++p_; return j; }
                fail("',' or '}' expected");
            }
        }
        if (c == '[') {
            ++p_;
            j.kind = Json::Arr;
            ws();
            if (p_ < e_ && *p_ == ']') { ++p_; return j; }
            for (;;) {
                j.arr.push_back(value());
                ws();
                if (p_ < e_ && *p_ == ',') { ++p_; continue; }
                if (p_ < e_ && *p_ == ']') { ++p_; return j; }
                fail("',' or ']' expected");
            }
        }
        if (c == '"') {
            j.kind = Json::Str;
            j.str = string();
            return j;
        }
        if (c == 't' && e_ - p_ >= 4 && !std::memcmp(p_, "true", 4)) { p_ += 4; j.kind = Json::Bool; j.b = true; return j; }
        if (c == 'f' && e_ - p_ >= 5 && !std::memcmp(p_, "false", 5)) { p_ += 5; j.kind = Json::Bool; return j; }
        if (c == 'n' && e_ - p_ >= 4 && !std::memcmp(p_, "null", 4)) { p_ += 4; return j; }
        // number: strtod parses the JSON grammar's numbers (and rounds correctly, like Python's float())
        std::string tmp;
        const char* q = p_;
        while (q < e_ && (std::strchr("+-0123456789.eE", *q) != nullptr)) ++q;
        if (q == p_) fail("unexpected character");
        tmp.assign(p_, q);
        char* endp = nullptr;
        j.kind = Json::Num;
        j.num = std::strtod(tmp.c_str(), &endp);
        if (endp == tmp.c_str()) fail("bad number");
        p_ = q;
        return j;
    }
    std::string string() {
        if (p_ >= e_ || *p_ != '"') fail("string expected");
        ++p_;
        std::string s;
        while (p_ < e_ && *p_ != '"') {
            char c = *p_++;
            if (c != '\\') { s.push_back(c); continue; }
            if (p_ >= e_) fail("bad escape");
            char x = *p_++;
            switch (x) {
                case 'n': s.push_back('\n'); break;
                case 't': s.push_back('\t'); break;
                case 'r': s.push_back('\r'); break;
                case 'b': s.push_back('\b'); break;
                case 'f': s.push_back('\f'); break;
                case 'u': {
                    if (e_ - p_ < 4) fail("bad \\u escape");
                    unsigned cp = (unsigned)std::strtoul(std::string(p_, p_ + 4).c_str(), nullptr, 16);
                    p_ += 4;
                    if (cp < 0x80) s.push_back((char)cp);
                    else if (cp < 0x800) { s.push_back((char)(0xC0 | (cp >> 6))); s.push_back((char)(0x80 | (cp & 0x3F))); }
                    else { s.push_back((char)(0xE0 | (cp >> 12))); s.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s.push_back((char)(0x80 | (cp & 0x3F))); }
                    break;
                }
                default: s.push_back(x);  // \" \\ \/
            }
        }
        if (p_ >= e_) fail("unterminated string");
        ++p_;
        return s;
    }
};

// ------------------------------------------------------------------------------------------------ zlib helper
inline std::vector<uint8_t> inflate_all(const uint8_t* src, size_t n, size_t expect /* 0 = unknown */) {
    std::vector<uint8_t> out(expect ? expect : n * 4 + 64);
    z_stream zs{};
    if (inflateInit(&zs) != Z_OK) throw std::runtime_error("zlib: inflateInit failed");
    zs.next_in = const_cast<Bytef*>(src);
    zs.avail_in = (uInt)n;
    size_t have = 0;
    for (;;) {
        if (have == out.size()) out.resize(out.size() * 2);
        zs.next_out = out.data() + have;
        zs.avail_out = (uInt)(out.size() - have);
        int r = inflate(&zs, Z_NO_FLUSH);
        have = out.size() - zs.avail_out;
        if (r == Z_STREAM_END) break;
        if (r != Z_OK || (zs.avail_in == 0 && zs.avail_out != 0)) {
            inflateEnd(&zs);
            throw std::runtime_error("zlib: corrupt deflate stream");
        }
    }
    inflateEnd(&zs);
    out.resize(have);
    return out;
}

// ------------------------------------------------------------------------------------------------ PNG -> RGBA8
struct Image {
    uint32_t w = 0, h = 0;
    std::vector<uint8_t> rgba;  // h x w x 4, sRGB-encoded colour as stored
};
inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

inline Image decode_png(const uint8_t* d, size_t n) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (n < 8 || std::memcmp(d, sig, 8)) throw std::runtime_error("image: not a PNG (only PNG and baseline JPEG are decoded natively)");
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    for (size_t p = 8; p + 12 <= n;) {
        uint32_t len = be32(d + p);
        const uint8_t* typ = d + p + 4;
        const uint8_t* body = d + p + 8;
        if (p + 12 + (size_t)len > n) throw std::runtime_error("PNG: truncated chunk");
        if (!std::memcmp(typ, "IHDR", 4)) {
            if (len < 13) throw std::runtime_error("PNG: short IHDR");
            w = be32(body);
            h = be32(body + 4);
            depth = body[8];
            ctype = body[9];
            interlace = body[12];
        } else if (!std::memcmp(typ, "PLTE", 4)) plte.assign(body, body + len);
        else if (!std::memcmp(typ, "tRNS", 4)) trns.assign(body, body + len);
        else if (!std::memcmp(typ, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!std::memcmp(typ, "IEND", 4)) break;
        p += 12 + (size_t)len;
    }
    if (!w || !h) throw std::runtime_error("PNG: no IHDR");
    if (w > 32768u || h > 32768u) throw std::runtime_error("PNG: image larger than 32768 x 32768");
    if (interlace > 1) throw std::runtime_error("PNG: unknown interlace method");
    const int chans = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!chans || !(depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) throw std::runtime_error("PNG: bad colour type / depth");
    const size_t bpp_bits = (size_t)chans * depth, bpp = bpp_bits >= 8 ? bpp_bits / 8 : 1;
    auto row_bytes = [&](uint32_t pw) { return ((size_t)pw * bpp_bits + 7) / 8; };
    // the image is one pass, or the seven Adam7 passes (PNG spec 8.2): pass pixel (px, py) is image pixel (x0 + px * dx, y0 + py * dy)
    struct Pass { uint32_t x0, y0, dx, dy, pw, ph; };
    std::vector<Pass> passes;
    if (!interlace) passes.push_back({0, 0, 1, 1, w, h});
    else {
        static const uint32_t a7[7][4] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
        for (auto& a : a7) {
            const uint32_t pw = w > a[0] ? (w - a[0] + a[2] - 1) / a[2] : 0, ph = h > a[1] ? (h - a[1] + a[3] - 1) / a[3] : 0;
            if (pw && ph) passes.push_back({a[0], a[1], a[2], a[3], pw, ph});
        }
    }
    size_t want = 0;
    for (auto& ps : passes) want += (row_bytes(ps.pw) + 1) * ps.ph;
    if (want > idat.size() * 1032 + 1024) throw std::runtime_error("PNG: IDAT is too short for the image size");  // deflate expands at most ~1032x
    std::vector<uint8_t> raw = inflate_all(idat.data(), idat.size(), want);
    if (raw.size() != want) throw std::runtime_error("PNG: wrong amount of image data");
    Image out;
    out.w = w;
    out.h = h;
    out.rgba.resize((size_t)w * h * 4);
    auto sample = [&](const uint8_t* r, size_t idx) -> uint32_t {  // idx-th sample of a row, as stored (not scaled)
        if (depth == 8) return r[idx];
        if (depth == 16) return ((uint32_t)r[2 * idx] << 8) | r[2 * idx + 1];
        const size_t bit = idx * depth;
        return (r[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u);
    };
    auto to8 = [&](uint32_t v) -> uint8_t { return depth == 16 ? (uint8_t)(v >> 8) : depth == 8 ? (uint8_t)v : (uint8_t)(v * 255u / ((1u << depth) - 1u)); };
    size_t off = 0;
    for (auto& ps : passes) {
        const size_t row = row_bytes(ps.pw);
        std::vector<uint8_t> img(row * ps.ph);
        for (uint32_t y = 0; y < ps.ph; y++) {  // undo the per-row filters (PNG spec 9.2)
            const uint8_t* in = raw.data() + off + (row + 1) * y;
            uint8_t* cur = img.data() + row * y;
            const uint8_t* up = y ? cur - row : nullptr;
            const int ft = in[0];
            for (size_t x = 0; x < row; x++) {
                const int a = x >= bpp ? cur[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= bpp) ? up[x - bpp] : 0;
                int pred = 0;
                switch (ft) {
                    case 0: pred = 0; break;
                    case 1: pred = a; break;
                    case 2: pred = b; break;
                    case 3: pred = (a + b) >> 1; break;
                    case 4: {
                        const int pp = a + b - c, pa = std::abs(pp - a), pb = std::abs(pp - b), pc = std::abs(pp - c);
                        pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                        break;
                    }
                    default: throw std::runtime_error("PNG: bad filter type");
                }
                cur[x] = (uint8_t)(in[1 + x] + pred);
            }
        }
        off += (row + 1) * ps.ph;
        for (uint32_t py = 0; py < ps.ph; py++) {
            const uint8_t* r = img.data() + row * py;
            for (uint32_t x = 0; x < ps.pw; x++) {
                uint8_t* o = out.rgba.data() + 4 * ((size_t)(ps.y0 + py * ps.dy) * w + ps.x0 + x * ps.dx);
                if (ctype == 3) {
                    const uint32_t i = sample(r, x);
                    if (3 * (size_t)i + 2 >= plte.size()) throw std::runtime_error("PNG: palette index out of range");
                    o[0] = plte[3 * i]; o[1] = plte[3 * i + 1]; o[2] = plte[3 * i + 2];
                    o[3] = i < trns.size() ? trns[i] : 255;
                } else if (ctype == 0 || ctype == 4) {
                    const uint32_t g = sample(r, (size_t)x * chans);
                    o[0] = o[1] = o[2] = to8(g);
                    o[3] = ctype == 4 ? to8(sample(r, (size_t)x * chans + 1)) : 255;
                    if (ctype == 0 && trns.size() >= 2 && g == (((uint32_t)trns[0] << 8) | trns[1])) o[3] = 0;
                } else {
                    const uint32_t R = sample(r, (size_t)x * chans), G = sample(r, (size_t)x * chans + 1), B = sample(r, (size_t)x * chans + 2);
                    o[0] = to8(R); o[1] = to8(G); o[2] = to8(B);
                    o[3] = ctype == 6 ? to8(sample(r, (size_t)x * chans + 3)) : 255;
                    if (ctype == 2 && trns.size() >= 6 && R == (((uint32_t)trns[0] << 8) | trns[1]) && G == (((uint32_t)trns[2] << 8) | trns[3]) &&
                        B == (((uint32_t)trns[4] << 8) | trns[5]))
                        o[3] = 0;
                }
            }
        }
    }
    return out;
}

// ------------------------------------------------------------------------------------------------ JPEG -> RGBA8
// ITU T.81 Huffman-coded DCT JPEG, 8-bit, 1 or 3 components (JFIF YCbCr), sampling factors 1 or 2, restart intervals:
// baseline / extended sequential (SOF0 / SOF1, interleaved or one scan per component) and progressive (SOF2: spectral selection and
// successive approximation, T.81 annex G).  Coefficients are collected over all scans, then dequantised and inverse-transformed once.
// Arithmetic-coded / lossless / 12-bit / CMYK files are reported as unsupported.  Chroma is upsampled by replication (libjpeg's
// default "fancy" triangle filter differs from this by a few levels along sharp chroma edges).
class JpegDecoder {
  public:
    static Image decode(const uint8_t* d, size_t n) {
        JpegDecoder j(d, n);
        return j.run();
    }

  private:
    const uint8_t* d_;
    size_t n_, p_ = 0;
    struct Huff {
        uint8_t bits[17] = {0}, vals[256] = {0};
        int mincode[17], maxcode[18], valptr[17];
        bool present = false;
    } dc_[4], ac_[4];
    uint16_t qt_[4][64] = {{0}};  // zig-zag order, as stored in the file
    bool qt_present_[4] = {false, false, false, false};
    struct Comp {
        int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0;
        int bw = 0, bh = 0;      // blocks per row / column of the MCU-padded plane
        int cw = 0, ch = 0;      // blocks that cover the component itself (geometry of a single-component scan)
        std::vector<int16_t> coef;  // bw * bh * 64, natural (row-major) order within a block
        uint16_t q[64] = {0};    // natural order; captured at the component's first scan
        bool q_set = false;
        std::vector<uint8_t> plane;
        int pw = 0, ph = 0;
    } comp_[3];
    int ncomp_ = 0, width_ = 0, height_ = 0, restart_ = 0, hmax_ = 1, vmax_ = 1, mcux_ = 0, mcuy_ = 0;
    bool progressive_ = false;
    uint32_t bitbuf_ = 0;
    int bitcnt_ = 0;
    bool hit_marker_ = false;
    int eobrun_ = 0;

    JpegDecoder(const uint8_t* d, size_t n) : d_(d), n_(n) {}
    [[noreturn]] static void fail(const char* w) { throw std::runtime_error(std::string("JPEG: ") + w); }
    int u8() { if (p_ >= n_) fail("truncated"); return d_[p_++]; }
    int u16() { int a = u8(); return (a << 8) | u8(); }
    static const uint8_t* zigzag() {
        static const uint8_t zz[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                       35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
        return zz;
    }

    static void build(Huff& h) {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; l++) {
            h.valptr[l] = k;
            h.mincode[l] = code;
            code += h.bits[l];
            if (code > (1 << l)) fail("over-subscribed Huffman table");
            k += h.bits[l];
            h.maxcode[l] = h.bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        h.maxcode[17] = 0x7FFFFFFF;
        h.present = true;
    }
    int getbit() {
        if (bitcnt_ == 0) {
            int b = 0;
            if (!hit_marker_) {
                b = p_ < n_ ? d_[p_++] : 0;
                if (b == 0xFF) {
                    int b2 = p_ < n_ ? d_[p_] : 0xD9;
                    if (b2 == 0) p_++;           // stuffed zero
                    else { hit_marker_ = true; p_--; b = 0; }  // a marker: feed zeros, leave the pointer on it
                }
            }
            bitbuf_ = (uint32_t)b;
            bitcnt_ = 8;
        }
        bitcnt_--;
        return (int)((bitbuf_ >> bitcnt_) & 1u);
    }
    int receive(int s) {
        if (s > 16) fail("bad magnitude category");
        uint32_t v = 0;
        for (int i = 0; i < s; i++) v = (v << 1) | (uint32_t)getbit();
        return (int)v;
    }
    static int extend(int v, int s) { return s && v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }
    int decode_sym(const Huff& h) {
        int code = 0;
        for (int l = 1; l <= 16; l++) {
            code = (code << 1) | getbit();
            if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
        }
        fail("bad Huffman code");
    }
    static void idct8x8(const float* in, uint8_t* out, int stride) {
        static float c[8][8];
        static bool init = false;
        if (!init) {
            for (int x = 0; x < 8; x++)
                for (int u = 0; u < 8; u++) c[x][u] = (u == 0 ? 0.35355339059327379f : 0.5f) * std::cos((2 * x + 1) * u * 3.14159265358979323846f / 16.0f);
            init = true;
        }
        float tmp[64];
        for (int y = 0; y < 8; y++)  // rows
            for (int x = 0; x < 8; x++) {
                float s = 0.0f;
                for (int u = 0; u < 8; u++) s += c[x][u] * in[8 * y + u];
                tmp[8 * y + x] = s;
            }
        for (int x = 0; x < 8; x++)  // columns
            for (int y = 0; y < 8; y++) {
                float s = 0.0f;
                for (int v = 0; v < 8; v++) s += c[y][v] * tmp[8 * v + x];
                const int q = (int)std::floor(s + 128.5f);
                out[y * stride + x] = (uint8_t)(q < 0 ? 0 : (q > 255 ? 255 : q));
            }
    }

    // ---- one block of one scan.  ss..se = spectral band, ah / al = successive-approximation bit positions (0,63,0,0 when sequential)
    void block_sequential(Comp& c, int16_t* coef) {
        const uint8_t* zz = zigzag();
        const int t = decode_sym(dc_[c.td]);
        c.pred += extend(receive(t), t);
        coef[0] = (int16_t)c.pred;
        for (int k = 1; k < 64;) {
            const int rs = decode_sym(ac_[c.ta]), r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r == 15) { k += 16; continue; }
                break;  // EOB
            }
            k += r;
            if (k > 63) fail("bad AC run");
            coef[zz[k]] = (int16_t)extend(receive(s), s);
            k++;
        }
    }
    void block_dc_first(Comp& c, int16_t* coef, int al) {  // G.1.2.1
        const int t = decode_sym(dc_[c.td]);
        c.pred += extend(receive(t), t);
        coef[0] = (int16_t)(c.pred * (1 << al));
    }
    void block_dc_refine(int16_t* coef, int al) {
        if (getbit()) coef[0] = (int16_t)(coef[0] | (1 << al));
    }
    void block_ac_first(Comp& c, int16_t* coef, int ss, int se, int al) {  // G.1.2.2
        const uint8_t* zz = zigzag();
        if (eobrun_ > 0) {
            eobrun_--;
            return;
        }
        for (int k = ss; k <= se;) {
            const int rs = decode_sym(ac_[c.ta]), r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r < 15) {  // EOBn: this band ends here for 2^r (+ extra bits) blocks, this one included
                    eobrun_ = (1 << r) - 1;
                    if (r) eobrun_ += receive(r);
                    break;
                }
                k += 16;
            } else {
                k += r;
                if (k > se) fail("bad AC run");
                coef[zz[k]] = (int16_t)(extend(receive(s), s) * (1 << al));
                k++;
            }
        }
    }
    void block_ac_refine(Comp& c, int16_t* coef, int ss, int se, int al) {  // G.1.2.3
        const uint8_t* zz = zigzag();
        const int p1 = 1 << al, m1 = -(1 << al);
        auto correct = [&](int16_t& v) {
            if (getbit() && (v & p1) == 0) v = (int16_t)(v + (v >= 0 ? p1 : m1));
        };
        int k = ss;
        if (eobrun_ == 0) {
            for (; k <= se; k++) {
                const int rs = decode_sym(ac_[c.ta]);
                int r = rs >> 4, s = rs & 15;
                if (s) {
                    if (s != 1) fail("bad refinement code");
                    s = getbit() ? p1 : m1;
                } else if (r != 15) {
                    eobrun_ = 1 << r;
                    if (r) eobrun_ += receive(r);
                    break;
                }
                // skip r zero-history coefficients; every nonzero-history one on the way takes a correction bit
                for (; k <= se; k++) {
                    int16_t& v = coef[zz[k]];
                    if (v != 0) correct(v);
                    else if (--r < 0) break;
                }
                if (s) {
                    if (k > se) fail("bad refinement run");
                    coef[zz[k]] = (int16_t)s;
                }
            }
        }
        if (eobrun_ > 0) {
            for (; k <= se; k++) {
                int16_t& v = coef[zz[k]];
                if (v != 0) correct(v);
            }
            eobrun_--;
        }
    }

    void setup_frame() {
        hmax_ = vmax_ = 1;
        for (int i = 0; i < ncomp_; i++) { hmax_ = std::max(hmax_, comp_[i].h); vmax_ = std::max(vmax_, comp_[i].v); }
        if (ncomp_ == 1) { comp_[0].h = comp_[0].v = 1; hmax_ = vmax_ = 1; }
        mcux_ = (width_ + 8 * hmax_ - 1) / (8 * hmax_);
        mcuy_ = (height_ + 8 * vmax_ - 1) / (8 * vmax_);
        for (int i = 0; i < ncomp_; i++) {
            Comp& c = comp_[i];
            c.bw = mcux_ * c.h;
            c.bh = mcuy_ * c.v;
            const int cwid = (width_ * c.h + hmax_ - 1) / hmax_, chei = (height_ * c.v + vmax_ - 1) / vmax_;
            c.cw = (cwid + 7) / 8;
            c.ch = (chei + 7) / 8;
            c.coef.assign((size_t)c.bw * c.bh * 64, 0);
        }
    }
    void restart_marker(const std::vector<int>& sc) {  // RSTn: byte-align, skip the marker, reset predictors and the EOB run
        bitcnt_ = 0;
        hit_marker_ = false;
        while (p_ + 1 < n_ && !(d_[p_] == 0xFF && d_[p_ + 1] >= 0xD0 && d_[p_ + 1] <= 0xD7)) p_++;
        p_ += 2;
        for (int ci : sc) comp_[ci].pred = 0;
        eobrun_ = 0;
    }
    void scan(const std::vector<int>& sc, int ss, int se, int ah, int al) {
        for (int ci : sc) {
            Comp& c = comp_[ci];
            c.pred = 0;
            if (!c.q_set) {
                if (!qt_present_[c.tq]) fail("missing quantisation table");
                for (int k = 0; k < 64; k++) c.q[zigzag()[k]] = qt_[c.tq][k];
                c.q_set = true;
            }
            const bool need_dc = ss == 0 && (!progressive_ || ah == 0), need_ac = se > 0;
            if ((need_dc && !dc_[c.td].present) || (need_ac && !ac_[c.ta].present)) fail("missing Huffman table");
        }
        bitcnt_ = 0;
        hit_marker_ = false;
        eobrun_ = 0;
        auto one = [&](Comp& c, int bx, int by) {
            int16_t* coef = c.coef.data() + ((size_t)by * c.bw + bx) * 64;
            if (!progressive_) block_sequential(c, coef);
            else if (ss == 0) ah == 0 ? block_dc_first(c, coef, al) : block_dc_refine(coef, al);
            else ah == 0 ? block_ac_first(c, coef, ss, se, al) : block_ac_refine(c, coef, ss, se, al);
        };
        int left = restart_;
        if (sc.size() == 1) {  // single-component scan: the component's own blocks in raster order, one block per "MCU"
            Comp& c = comp_[sc[0]];
            for (int by = 0; by < c.ch; by++)
                for (int bx = 0; bx < c.cw; bx++) {
                    if (restart_ && left == 0) { restart_marker(sc); left = restart_; }
                    one(c, bx, by);
                    if (restart_) left--;
                }
        } else {
            for (int my = 0; my < mcuy_; my++)
                for (int mx = 0; mx < mcux_; mx++) {
                    if (restart_ && left == 0) { restart_marker(sc); left = restart_; }
                    for (int ci : sc) {
                        Comp& c = comp_[ci];
                        for (int by = 0; by < c.v; by++)
                            for (int bx = 0; bx < c.h; bx++) one(c, mx * c.h + bx, my * c.v + by);
                    }
                    if (restart_) left--;
                }
        }
        // leave p_ on the next marker
        if (!hit_marker_)
            while (p_ + 1 < n_ && !(d_[p_] == 0xFF && d_[p_ + 1] != 0 && !(d_[p_ + 1] >= 0xD0 && d_[p_ + 1] <= 0xD7))) p_++;
    }
    Image run() {
        if (n_ < 4 || d_[0] != 0xFF || d_[1] != 0xD8) fail("no SOI marker");
        p_ = 2;
        bool have_frame = false, have_scan = false;
        for (;;) {
            if (p_ >= n_) {
                if (have_scan) break;  // tolerate a missing EOI
                fail("truncated");
            }
            int m = u8();
            if (m != 0xFF) continue;
            while (p_ < n_ && (m = u8()) == 0xFF) {}
            if (m == 0xD9) {
                if (!have_scan) fail("no scan before EOI");
                break;
            }
            if (m == 0 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
            const int len = u16();
            const size_t end = p_ + (size_t)len - 2;
            if (len < 2 || end > n_) fail("truncated segment");
            if (m == 0xDB) {
                while (p_ < end) {
                    const int pq = u8(), t = pq & 15;
                    if (t > 3) fail("bad quantisation table id");
                    for (int k = 0; k < 64; k++) qt_[t][k] = (uint16_t)((pq >> 4) ? u16() : u8());
                    qt_present_[t] = true;
                }
            } else if (m == 0xC4) {
                while (p_ < end) {
                    const int tc = u8(), t = tc & 15;
                    if (t > 3) fail("bad Huffman table id");
                    Huff& h = (tc >> 4) ? ac_[t] : dc_[t];
                    int total = 0;
                    for (int l = 1; l <= 16; l++) { h.bits[l] = (uint8_t)u8(); total += h.bits[l]; }
                    if (total > 256) fail("bad Huffman table");
                    for (int k = 0; k < total; k++) h.vals[k] = (uint8_t)u8();
                    build(h);
                }
            } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
                if (have_frame) fail("more than one frame header");
                progressive_ = m == 0xC2;
                if (u8() != 8) fail("only 8-bit samples are supported");
                height_ = u16();
                width_ = u16();
                ncomp_ = u8();
                if ((ncomp_ != 1 && ncomp_ != 3) || !width_ || !height_) fail("only 1- and 3-component images are supported");
                if ((size_t)width_ * (size_t)height_ > ((size_t)1 << 28)) fail("image larger than 2^28 pixels");
                if (p_ + 3 * (size_t)ncomp_ > end) fail("short frame header");
                for (int i = 0; i < ncomp_; i++) {
                    comp_[i].id = u8();
                    const int hv = u8();
                    comp_[i].h = hv >> 4;
                    comp_[i].v = hv & 15;
                    comp_[i].tq = u8();
                    if (comp_[i].h < 1 || comp_[i].h > 2 || comp_[i].v < 1 || comp_[i].v > 2 || comp_[i].tq > 3) fail("unsupported sampling factors");
                }
                setup_frame();
                have_frame = true;
            } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
                fail("lossless / hierarchical / arithmetic-coded JPEG is not supported");
            } else if (m == 0xDD) {
                restart_ = u16();
            } else if (m == 0xDA) {
                if (!have_frame) fail("scan before frame header");
                const int ns = u8();
                if (ns < 1 || ns > ncomp_) fail("bad component count in scan header");
                std::vector<int> sc;
                for (int i = 0; i < ns; i++) {
                    const int id = u8(), tt = u8();
                    int ci = -1;
                    for (int k = 0; k < ncomp_; k++)
                        if (comp_[k].id == id) ci = k;
                    if (ci < 0) fail("scan references an unknown component");
                    comp_[ci].td = tt >> 4;
                    comp_[ci].ta = tt & 15;
                    if (comp_[ci].td > 3 || comp_[ci].ta > 3) fail("bad Huffman table id");
                    sc.push_back(ci);
                }
                int ss = u8(), se = u8();
                const int a = u8(), ah = a >> 4, al = a & 15;
                if (!progressive_) { ss = 0; se = 63; }
                else if (ss > se || se > 63 || (ss == 0 && se != 0) || (ss > 0 && ns != 1) || al > 13) fail("bad progressive scan parameters");
                p_ = end;
                scan(sc, ss, se, progressive_ ? ah : 0, progressive_ ? al : 0);
                have_scan = true;
                continue;
            }
            p_ = end;
        }
        return finish();
    }
    Image finish() {
        for (int i = 0; i < ncomp_; i++) {
            Comp& c = comp_[i];
            if (!c.q_set) fail("a component has no scan");
            c.pw = c.bw * 8;
            c.ph = c.bh * 8;
            c.plane.assign((size_t)c.pw * c.ph, 0);
            float f[64];
            for (int by = 0; by < c.bh; by++)
                for (int bx = 0; bx < c.bw; bx++) {
                    const int16_t* coef = c.coef.data() + ((size_t)by * c.bw + bx) * 64;
                    for (int k = 0; k < 64; k++) f[k] = (float)((int)coef[k] * (int)c.q[k]);
                    idct8x8(f, c.plane.data() + (size_t)(8 * by) * c.pw + 8 * bx, c.pw);
                }
        }
        const int hmax = hmax_, vmax = vmax_;
        Image out;
        out.w = (uint32_t)width_;
        out.h = (uint32_t)height_;
        out.rgba.resize((size_t)width_ * height_ * 4);
        for (int y = 0; y < height_; y++)
            for (int x = 0; x < width_; x++) {
                uint8_t* o = out.rgba.data() + 4 * ((size_t)y * width_ + x);
                auto sample = [&](const Comp& c) -> float { return (float)c.plane[(size_t)(y * c.v / vmax) * c.pw + (x * c.h / hmax)]; };
                if (ncomp_ == 1) {
                    o[0] = o[1] = o[2] = (uint8_t)sample(comp_[0]);
                } else {
                    const float Y = sample(comp_[0]), cb = sample(comp_[1]) - 128.0f, cr = sample(comp_[2]) - 128.0f;
                    auto clamp8 = [](float v) -> uint8_t { const int q = (int)std::floor(v + 0.5f); return (uint8_t)(q < 0 ? 0 : (q > 255 ? 255 : q)); };
                    o[0] = clamp8(Y + 1.402f * cr);
                    o[1] = clamp8(Y - 0.344136f * cb - 0.714136f * cr);
                    o[2] = clamp8(Y + 1.772f * cb);
                }
                o[3] = 255;
            }
        return out;
    }
};
// PNG or JPEG, by signature
inline Image decode_image(const uint8_t* d, size_t n) {
    if (n >= 2 && d[0] == 0xFF && d[1] == 0xD8) return JpegDecoder::decode(d, n);
    return decode_png(d, n);
}

// ------------------------------------------------------------------------------------------------ Mesh (assets/mod.rs:118-133, world/mod.rs:103-125)
struct Material {  // assets/mod.rs:52-59 (+ emission, datatypes.slang:17)
    float color[3] = {0.8f, 0.8f, 0.8f};
    float metalic_factor = 0.0f, roughness_factor = 1.0f;
    float emission[3] = {0.0f, 0.0f, 0.0f};
    int32_t texture_offset = -1;
};
struct Mesh {
    std::vector<float> vertices;                // n x 8: position, normal, uv (Vertex, assets/mod.rs:127-133)
    std::vector<uint32_t> indices;              // relative to each geometry's vertex_offset
    std::vector<rt3_geometry_info> geometries;  // one per glTF primitive
    std::vector<uint32_t> prim_counts;
    std::vector<std::string> names;
    std::vector<Image> textures;
    size_t n_vertices() const { return vertices.size() / 8; }
    size_t n_triangles() const {
        size_t t = 0;
        for (uint32_t c : prim_counts) t += c;
        return t;
    }
};

struct Mat4 {  // row-major double
    double m[4][4];
    static Mat4 identity() {
        Mat4 r{};
        for (int i = 0; i < 4; i++) r.m[i][i] = 1.0;
        return r;
    }
    bool is_identity() const {
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++)
                if (m[i][j] != (i == j ? 1.0 : 0.0)) return false;
        return true;
    }
};
inline Mat4 mul(const Mat4& a, const Mat4& b) {
    Mat4 r{};
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0.0;
            for (int k = 0; k < 4; k++) s += a.m[i][k] * b.m[k][j];
            r.m[i][j] = s;
        }
    return r;
}
// TRS or matrix of a glTF node (column-major "matrix"; T * R * S otherwise)
inline Mat4 node_matrix(const Json& node) {
    Mat4 r = Mat4::identity();
    if (const Json* mm = node.find("matrix")) {
        for (int c = 0; c < 4; c++)
            for (int rr = 0; rr < 4; rr++) r.m[rr][c] = mm->at((size_t)(4 * c + rr)).number(0.0);
        return r;
    }
    if (const Json* s = node.find("scale")) {
        Mat4 S = Mat4::identity();
        for (int i = 0; i < 3; i++) S.m[i][i] = s->at((size_t)i).number(1.0);
        r = mul(S, r);
    }
    if (const Json* q = node.find("rotation")) {
        const double x = q->at(0).number(0), y = q->at(1).number(0), z = q->at(2).number(0), w = q->at(3).number(1);
        Mat4 R = Mat4::identity();
        R.m[0][0] = 1 - 2 * (y * y + z * z); R.m[0][1] = 2 * (x * y - z * w); R.m[0][2] = 2 * (x * z + y * w);
        R.m[1][0] = 2 * (x * y + z * w); R.m[1][1] = 1 - 2 * (x * x + z * z); R.m[1][2] = 2 * (y * z - x * w);
        R.m[2][0] = 2 * (x * z - y * w); R.m[2][1] = 2 * (y * z + x * w); R.m[2][2] = 1 - 2 * (x * x + y * y);
        r = mul(R, r);
    }
    if (const Json* t = node.find("translation")) {
        Mat4 T = Mat4::identity();
        for (int i = 0; i < 3; i++) T.m[i][3] = t->at((size_t)i).number(0.0);
        r = mul(T, r);
    }
    return r;
}
// inverse transpose of the upper 3x3 (normal matrix)
inline void normal_matrix(const Mat4& m, double n[3][3]) {
    const double a = m.m[0][0], b = m.m[0][1], c = m.m[0][2], d = m.m[1][0], e = m.m[1][1], f = m.m[1][2], g = m.m[2][0], h = m.m[2][1], i = m.m[2][2];
    const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    if (det == 0.0) throw std::runtime_error("glTF: singular node transform");
    const double inv[3][3] = {{(e * i - f * h) / det, (c * h - b * i) / det, (b * f - c * e) / det},
                              {(f * g - d * i) / det, (a * i - c * g) / det, (c * d - a * f) / det},
                              {(d * h - e * g) / det, (b * g - a * h) / det, (a * e - b * d) / det}};
    for (int r = 0; r < 3; r++)
        for (int cc = 0; cc < 3; cc++) n[r][cc] = inv[cc][r];
}

// ------------------------------------------------------------------------------------------------ glTF binary
class GltfMeshLoader {  // assets/mod.rs:179-200 (`impl AssetLoader`), extensions() = ["glb"]
  public:
    static Mesh load(const std::string& path) {
        std::vector<uint8_t> data = read_file(path);
        if (data.size() < 20 || std::memcmp(data.data(), "glTF", 4) || le32(data.data() + 4) != 2) throw std::runtime_error(path + ": not a glTF 2 binary");
        const size_t total = std::min<size_t>(le32(data.data() + 8), data.size());
        Json doc;
        bool have_doc = false;
        const uint8_t* blob = nullptr;
        size_t blob_n = 0;
        for (size_t p = 12; p + 8 <= total;) {
            const uint32_t ln = le32(data.data() + p);
            const uint8_t* typ = data.data() + p + 4;
            if (p + 8 + (size_t)ln > data.size()) throw std::runtime_error(path + ": truncated chunk");
            if (!std::memcmp(typ, "JSON", 4)) {
                doc = JsonParser((const char*)data.data() + p + 8, ln).parse();
                have_doc = true;
            } else if (!std::memcmp(typ, "BIN\0", 4)) {
                blob = data.data() + p + 8;
                blob_n = ln;
            }
            p += 8 + (size_t)ln;
        }
        if (!have_doc) throw std::runtime_error(path + ": no JSON chunk");
        GltfMeshLoader L{doc, blob, blob_n, path, {}};
        const Json& scenes = doc.at("scenes");
        const Json& scene = scenes.at((size_t)(doc.has("scene") ? doc.at("scene").integer(0) : 0));
        for (const Json& root : scene.at("nodes").arr) L.visit((size_t)root.integer(0), Mat4::identity());
        if (const Json* tex = doc.find("textures"))
            for (const Json& t : tex->arr) L.mesh_.textures.push_back(L.image((size_t)t.at("source").integer(0)));
        return std::move(L.mesh_);
    }

  private:
    const Json& doc_;
    const uint8_t* blob_;
    size_t blob_n_;
    std::string path_;
    Mesh mesh_;

    static uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

    struct Accessor {
        std::vector<double> v;  // count x ncomp (integers are exact in a double)
        size_t count = 0, ncomp = 0;
        bool is_float = false;
    };
    Accessor accessor(size_t i) const {
        const Json& a = doc_.at("accessors").at(i);
        const Json& bv = doc_.at("bufferViews").at((size_t)a.at("bufferView").integer(0));
        const int ct = (int)a.at("componentType").integer(0);
        const size_t sz = ct == 5120 || ct == 5121 ? 1 : ct == 5122 || ct == 5123 ? 2 : ct == 5125 || ct == 5126 ? 4 : 0;
        if (!sz) throw std::runtime_error("glTF: bad componentType");
        const std::string& ty = a.at("type").str;
        const size_t nc = ty == "SCALAR" ? 1 : ty == "VEC2" ? 2 : ty == "VEC3" ? 3 : ty == "VEC4" ? 4 : ty == "MAT4" ? 16 : 0;
        if (!nc) throw std::runtime_error("glTF: bad accessor type");
        const size_t start = (size_t)(bv.has("byteOffset") ? bv.at("byteOffset").integer(0) : 0) + (size_t)(a.has("byteOffset") ? a.at("byteOffset").integer(0) : 0);
        size_t stride = bv.has("byteStride") ? (size_t)bv.at("byteStride").integer(0) : 0;
        if (!stride) stride = sz * nc;
        Accessor r;
        r.count = (size_t)a.at("count").integer(0);
        r.ncomp = nc;
        r.is_float = ct == 5126;
        if (r.count && start + (r.count - 1) * stride + sz * nc > blob_n_) throw std::runtime_error("glTF: accessor exceeds the binary chunk");
        const bool normalized = a.has("normalized") && a.at("normalized").b && ct != 5126;
        const double nmax = ct == 5120 ? 127.0 : ct == 5121 ? 255.0 : ct == 5122 ? 32767.0 : ct == 5123 ? 65535.0 : 4294967295.0;
        r.v.resize(r.count * nc);
        for (size_t k = 0; k < r.count; k++)
            for (size_t c = 0; c < nc; c++) {
                const uint8_t* p = blob_ + start + k * stride + c * sz;
                double v;
                switch (ct) {
                    case 5120: v = (int8_t)p[0]; break;
                    case 5121: v = p[0]; break;
                    case 5122: v = (int16_t)((uint16_t)p[0] | ((uint16_t)p[1] << 8)); break;
                    case 5123: v = (uint16_t)((uint16_t)p[0] | ((uint16_t)p[1] << 8)); break;
                    case 5125: v = le32(p); break;
                    default: {
                        uint32_t u = le32(p);
                        float f;
                        std::memcpy(&f, &u, 4);
                        v = f;
                    }
                }
                if (normalized) v = (double)((float)v / (float)nmax);
                r.v[k * nc + c] = v;
            }
        if (normalized) r.is_float = true;
        return r;
    }

    Image image(size_t i) const {
        const Json& img = doc_.at("images").at(i);
        if (img.has("bufferView")) {
            const Json& bv = doc_.at("bufferViews").at((size_t)img.at("bufferView").integer(0));
            const size_t off = bv.has("byteOffset") ? (size_t)bv.at("byteOffset").integer(0) : 0, len = (size_t)bv.at("byteLength").integer(0);
            if (off + len > blob_n_) throw std::runtime_error("glTF: image exceeds the binary chunk");
            return decode_image(blob_ + off, len);
        }
        const size_t slash = path_.find_last_of('/');
        const std::string dir = slash == std::string::npos ? std::string() : path_.substr(0, slash + 1);
        std::vector<uint8_t> raw = read_file(dir + img.at("uri").str);
        return decode_image(raw.data(), raw.size());
    }

    void visit(size_t ni, const Mat4& parent) {
        const Json& node = doc_.at("nodes").at(ni);
        const Mat4 m = mul(parent, node_matrix(node));
        if (node.has("mesh")) {
            const Json& gm = doc_.at("meshes").at((size_t)node.at("mesh").integer(0));
            const bool identity = m.is_identity();
            double nm[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
            if (!identity) normal_matrix(m, nm);
            size_t pi = 0;
            for (const Json& prim : gm.at("primitives").arr) {
                const size_t this_pi = pi++;
                if (prim.has("mode") && prim.at("mode").integer(4) != 4) continue;  // triangles only
                const Json& at = prim.at("attributes");
                Accessor pos = accessor((size_t)at.at("POSITION").integer(0));
                const size_t nv = pos.count;
                std::vector<float> P(nv * 3);
                for (size_t k = 0; k < nv; k++)
                    for (int r = 0; r < 3; r++) {
                        const double* p = &pos.v[3 * k];
                        P[3 * k + r] = identity ? (float)p[r] : (float)(p[0] * m.m[r][0] + p[1] * m.m[r][1] + p[2] * m.m[r][2] + m.m[r][3]);
                    }
                std::vector<uint32_t> idx;
                if (prim.has("indices")) {
                    Accessor ia = accessor((size_t)prim.at("indices").integer(0));
                    idx.resize(ia.count);
                    for (size_t k = 0; k < ia.count; k++) idx[k] = (uint32_t)ia.v[k];
                } else {
                    idx.resize(nv);
                    for (size_t k = 0; k < nv; k++) idx[k] = (uint32_t)k;
                }
                idx.resize(idx.size() / 3 * 3);
                for (uint32_t v : idx)
                    if (v >= nv) throw std::runtime_error("glTF: index beyond the vertex count");
                std::vector<float> N(nv * 3);
                bool have_normals = at.has("NORMAL");
                if (have_normals) {
                    Accessor na = accessor((size_t)at.at("NORMAL").integer(0));
                    if (na.count != nv) throw std::runtime_error("glTF: NORMAL count differs from POSITION");
                    const bool keep = identity && na.is_float;
                    for (size_t k = 0; k < nv; k++) {
                        const double* n = &na.v[3 * k];
                        if (keep) {  // stored unit normals pass through bit for bit
                            for (int r = 0; r < 3; r++) N[3 * k + r] = (float)n[r];
                        } else {
                            double t[3];
                            for (int r = 0; r < 3; r++) t[r] = n[0] * nm[r][0] + n[1] * nm[r][1] + n[2] * nm[r][2];
                            store_normalized(t, &N[3 * k]);
                        }
                    }
                } else {  // area-weighted vertex normals from the (transformed, fp32) positions
                    std::vector<double> acc(nv * 3, 0.0);
                    for (int k = 0; k < 3; k++)  // corner by corner, like the Python loader's three np.add.at passes
                        for (size_t t = 0; t + 2 < idx.size(); t += 3) {
                            const float *a = &P[3 * idx[t]], *b = &P[3 * idx[t + 1]], *c = &P[3 * idx[t + 2]];
                            const float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
                            const float fn[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
                            for (int r = 0; r < 3; r++) acc[3 * idx[t + k] + r] += fn[r];
                        }
                    for (size_t k = 0; k < nv; k++) store_normalized(&acc[3 * k], &N[3 * k]);
                }
                std::vector<float> UV(nv * 2, 0.0f);
                if (at.has("TEXCOORD_0")) {
                    Accessor ua = accessor((size_t)at.at("TEXCOORD_0").integer(0));
                    if (ua.count != nv) throw std::runtime_error("glTF: TEXCOORD_0 count differs from POSITION");
                    for (size_t k = 0; k < 2 * nv; k++) UV[k] = (float)ua.v[k];
                }
                Material mat;
                if (prim.has("material")) {
                    const Json& gmtl = doc_.at("materials").at((size_t)prim.at("material").integer(0));
                    static const Json empty_obj = [] { Json j; j.kind = Json::Obj; return j; }();
                    const Json& pbr = gmtl.has("pbrMetallicRoughness") ? gmtl.at("pbrMetallicRoughness") : empty_obj;
                    for (int k = 0; k < 3; k++) mat.color[k] = pbr.has("baseColorFactor") ? (float)pbr.at("baseColorFactor").at((size_t)k).number(1.0) : 1.0f;
                    mat.metalic_factor = pbr.has("metallicFactor") ? (float)pbr.at("metallicFactor").number(1.0) : 1.0f;
                    mat.roughness_factor = pbr.has("roughnessFactor") ? (float)pbr.at("roughnessFactor").number(1.0) : 1.0f;
                    double strength = 1.0;
                    if (const Json* ext = gmtl.find("extensions"))
                        if (const Json* es = ext->find("KHR_materials_emissive_strength"))
                            if (const Json* s = es->find("emissiveStrength")) strength = s->number(1.0);
                    for (int k = 0; k < 3; k++) mat.emission[k] = gmtl.has("emissiveFactor") ? (float)(gmtl.at("emissiveFactor").at((size_t)k).number(0.0) * strength) : 0.0f;
                    if (const Json* bt = pbr.find("baseColorTexture")) mat.texture_offset = (int32_t)bt->at("index").integer(-1);
                }
                rt3_geometry_info g{};
                for (int k = 0; k < 3; k++) {
                    g.base_color[k] = mat.color[k];
                    g.emission[k] = mat.emission[k];
                }
                g.base_color[3] = 1.0f;
                g.emission[3] = 0.0f;
                g.base_color_texture_index = mat.texture_offset;
                g.metallic_factor = mat.metalic_factor;
                g.roughness = mat.roughness_factor;
                g.index_offset = (uint32_t)mesh_.indices.size();
                g.vertex_offset = (uint32_t)mesh_.n_vertices();
                for (size_t k = 0; k < nv; k++) {
                    const float v[8] = {P[3 * k], P[3 * k + 1], P[3 * k + 2], N[3 * k], N[3 * k + 1], N[3 * k + 2], UV[2 * k], UV[2 * k + 1]};
                    mesh_.vertices.insert(mesh_.vertices.end(), v, v + 8);
                }
                mesh_.indices.insert(mesh_.indices.end(), idx.begin(), idx.end());
                mesh_.geometries.push_back(g);
                mesh_.prim_counts.push_back((uint32_t)(idx.size() / 3));
                mesh_.names.push_back((gm.has("name") ? gm.at("name").str : std::string("mesh")) + "." + std::to_string(this_pi));
            }
        }
        if (const Json* ch = node.find("children"))
            for (const Json& c : ch->arr) visit((size_t)c.integer(0), m);
    }
    static void store_normalized(const double* t, float* out) {
        double len = std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
        if (len < 1e-20) len = 1e-20;
        for (int r = 0; r < 3; r++) out[r] = (float)(t[r] / len);
    }
    GltfMeshLoader(const Json& d, const uint8_t* b, size_t n, std::string p, Mesh m) : doc_(d), blob_(b), blob_n_(n), path_(std::move(p)), mesh_(std::move(m)) {}
};
inline Mesh load_glb(const std::string& path) { return GltfMeshLoader::load(path); }

// ------------------------------------------------------------------------------------------------ OpenEXR (scanline)
struct SkyImage {
    uint32_t w = 0, h = 0;
    std::vector<float> rgb;  // h x w x 3
};
inline float half_to_float(uint16_t hbits) {
    const uint32_t s = (hbits >> 15) & 1u, e = (hbits >> 10) & 31u, m = hbits & 1023u;
    uint32_t u;
    if (e == 0) {
        if (m == 0) u = s << 31;
        else {  // subnormal: renormalise
            int sh = 0;
            uint32_t mm = m;
            while (!(mm & 1024u)) { mm <<= 1; ++sh; }
            u = (s << 31) | ((uint32_t)(127 - 15 - sh + 1) << 23) | ((mm & 1023u) << 13);
        }
    } else if (e == 31) u = (s << 31) | 0x7F800000u | (m << 13);
    else u = (s << 31) | ((e + 112u) << 23) | (m << 13);
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// ---- OpenEXR PIZ: 16-bit range compaction (bitmap + LUT) -> 2-D Haar-like wavelet (ImfWav) -> Huffman with run lengths (ImfHuf),
//      restated from the published algorithm.  Parity unpinned: no EXR ships with the reference and no other EXR implementation exists in
//      this image; checked against the independent Python decoder / encoder of raytracer3_amd/assets.py.
namespace piz {
constexpr int kEncSize = (1 << 16) + 1, kDecBits = 14, kShortZeroRun = 59, kLongZeroRun = 63, kShortestLongRun = 2 + kLongZeroRun - kShortZeroRun;
struct BitReader {
    const uint8_t *p, *e;
    unsigned __int128 c = 0;  // up to 57 buffered bits + one more byte while a 58-bit code is matched
    int lc = 0;
    uint32_t get(int n) {  // MSB first; bytes past the end read as zero
        while (lc < n) {
            c = (c << 8) | (p < e ? *p : 0u);
            ++p;
            lc += 8;
        }
        lc -= n;
        return (uint32_t)((uint64_t)(c >> lc) & ((1ull << n) - 1ull));
    }
};
inline void canonical(std::vector<uint64_t>& h) {  // in: code lengths; out: length | code << 6 (hufCanonicalCodeTable)
    uint64_t n[59] = {0};
    for (uint64_t l : h) n[l]++;
    uint64_t c = 0;
    for (int i = 58; i > 0; --i) {
        const uint64_t nc = (c + n[i]) >> 1;
        n[i] = c;
        c = nc;
    }
    for (auto& v : h) {
        const uint64_t l = v;
        if (l > 0) v = l | (n[l]++ << 6);
    }
}
inline std::vector<uint16_t> huf_uncompress(const uint8_t* comp, size_t n_comp, size_t n_raw) {
    std::vector<uint16_t> out(n_raw);
    if (n_raw == 0) return out;
    if (n_comp < 20) throw std::runtime_error("EXR PIZ: truncated Huffman block");
    auto u32 = [&](size_t o) { return (uint32_t)comp[o] | ((uint32_t)comp[o + 1] << 8) | ((uint32_t)comp[o + 2] << 16) | ((uint32_t)comp[o + 3] << 24); };
    const uint32_t im = u32(0), iM = u32(4), table_len = u32(8), n_bits = u32(12);
    if (im >= (uint32_t)kEncSize || iM >= (uint32_t)kEncSize || 20ull + table_len > n_comp) throw std::runtime_error("EXR PIZ: bad Huffman header");
    std::vector<uint64_t> h(kEncSize, 0);
    BitReader tb{comp + 20, comp + 20 + table_len};
    for (uint32_t s = im; s <= iM;) {  // hufUnpackEncTable
        const uint32_t l = tb.get(6);
        if (l == (uint32_t)kLongZeroRun) s += tb.get(8) + kShortestLongRun;
        else if (l >= (uint32_t)kShortZeroRun) s += l - kShortZeroRun + 2;
        else h[s++] = l;
    }
    canonical(h);
    struct Dec { uint8_t len = 0; uint32_t lit = 0; std::vector<uint32_t> longs; };
    std::vector<Dec> dec(1u << kDecBits);
    for (uint32_t s = im; s <= iM; s++) {  // hufBuildDecTable
        const uint64_t c = h[s] >> 6;
        const int l = (int)(h[s] & 63);
        if (!l) continue;
        if (c >> l) throw std::runtime_error("EXR PIZ: invalid Huffman table");
        if (l > kDecBits) dec[c >> (l - kDecBits)].longs.push_back(s);
        else
            for (uint64_t k = c << (kDecBits - l), e = k + (1ull << (kDecBits - l)); k < e; k++) {
                dec[k].len = (uint8_t)l;
                dec[k].lit = s;
            }
    }
    const uint8_t* data = comp + 20 + table_len;
    const size_t n_data = std::min<size_t>(n_comp - 20 - table_len, ((size_t)n_bits + 7) / 8);
    BitReader br{data, data + n_data};
    size_t produced = 0;
    while (produced < n_raw) {  // hufDecode
        while (br.lc < 58) {
            br.c = (br.c << 8) | (br.p < br.e ? *br.p : 0u);
            ++br.p;
            br.lc += 8;
        }
        if (br.p > br.e + 32) throw std::runtime_error("EXR PIZ: Huffman data ends early");
        const Dec& d = dec[(uint32_t)(br.c >> (br.lc - kDecBits)) & ((1u << kDecBits) - 1u)];
        uint32_t sym = 0;
        if (d.len) {
            sym = d.lit;
            br.lc -= d.len;
        } else {
            bool found = false;
            for (uint32_t cand : d.longs) {
                const int l = (int)(h[cand] & 63);
                if (br.lc >= l && ((uint64_t)(br.c >> (br.lc - l)) & ((1ull << l) - 1ull)) == (h[cand] >> 6)) {
                    sym = cand;
                    br.lc -= l;
                    found = true;
                    break;
                }
            }
            if (!found) throw std::runtime_error("EXR PIZ: invalid Huffman code");
        }
        if (sym == iM) {  // run-length symbol: repeat the previous value
            const uint32_t run = br.get(8);
            if (produced == 0 || produced + run > n_raw) throw std::runtime_error("EXR PIZ: bad run length");
            for (uint32_t k = 0; k < run; k++, produced++) out[produced] = out[produced - 1];
        } else {
            out[produced++] = (uint16_t)sym;
        }
    }
    return out;
}
inline void wdec14(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) {
    const int ls = (int16_t)l, hs = (int16_t)h, ai = ls + (hs & 1) + (hs >> 1);
    a = (uint16_t)(int16_t)ai;
    b = (uint16_t)(int16_t)(ai - hs);
}
inline void wdec16(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) {
    const int m = l, d = h, bb = (m - (d >> 1)) & 0xFFFF, aa = (d + bb - 0x8000) & 0xFFFF;
    b = (uint16_t)bb;
    a = (uint16_t)aa;
}
inline void wav2_decode(uint16_t* in, int nx, int ox, int ny, int oy, uint16_t mx) {  // ImfWav wav2Decode
    const bool w14 = mx < (1 << 14);
    const int n = nx > ny ? ny : nx;
    int p = 1, p2;
    while (p <= n) p <<= 1;
    p >>= 1;
    p2 = p;
    p >>= 1;
    auto dec = [&](uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) { w14 ? wdec14(l, h, a, b) : wdec16(l, h, a, b); };
    while (p >= 1) {
        uint16_t* py = in;
        uint16_t* ey = in + (ptrdiff_t)oy * (ny - p2);
        const ptrdiff_t oy1 = (ptrdiff_t)oy * p, oy2 = (ptrdiff_t)oy * p2, ox1 = (ptrdiff_t)ox * p, ox2 = (ptrdiff_t)ox * p2;
        uint16_t i00, i01, i10, i11;
        for (; py <= ey; py += oy2) {
            uint16_t* px = py;
            uint16_t* ex = py + (ptrdiff_t)ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                uint16_t *p01 = px + ox1, *p10 = px + oy1, *p11 = p10 + ox1;
                dec(*px, *p10, i00, i10);
                dec(*p01, *p11, i01, i11);
                dec(i00, i01, *px, *p01);
                dec(i10, i11, *p10, *p11);
            }
            if (nx & p) {
                uint16_t* p10 = px + oy1;
                dec(*px, *p10, i00, *p10);
                *px = i00;
            }
        }
        if (ny & p) {
            uint16_t* px = py;
            uint16_t* ex = py + (ptrdiff_t)ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                uint16_t* p01 = px + ox1;
                dec(*px, *p01, i00, *p01);
                *px = i00;
            }
        }
        p2 = p;
        p >>= 1;
    }
}
// block -> uncompressed scanline block; sizes[c] = 16-bit words per sample of channel c (1 HALF, 2 FLOAT / UINT)
inline std::vector<uint8_t> unpack(const uint8_t* comp, size_t n_comp, const std::vector<int>& sizes, uint32_t w, uint32_t n_lines) {
    size_t total = 0;
    for (int sz : sizes) total += (size_t)sz * w * n_lines;
    if (n_comp < 8) throw std::runtime_error("EXR PIZ: truncated block");
    const uint32_t lo = comp[0] | (comp[1] << 8), hi = comp[2] | (comp[3] << 8);
    std::vector<uint8_t> bitmap(8192, 0);
    size_t p = 4;
    if (lo <= hi) {
        if (hi >= 8192 || p + (hi - lo + 1) + 4 > n_comp) throw std::runtime_error("EXR PIZ: bad bitmap range");
        std::memcpy(&bitmap[lo], comp + p, hi - lo + 1);
        p += hi - lo + 1;
    }
    const int32_t length = (int32_t)((uint32_t)comp[p] | ((uint32_t)comp[p + 1] << 8) | ((uint32_t)comp[p + 2] << 16) | ((uint32_t)comp[p + 3] << 24));
    p += 4;
    if (length < 0 || p + (size_t)length > n_comp) throw std::runtime_error("EXR PIZ: bad Huffman length");
    std::vector<uint16_t> lut(65536, 0);
    uint32_t k = 0;
    for (uint32_t i = 0; i < 65536; i++)
        if (i == 0 || (bitmap[i >> 3] & (1u << (i & 7)))) lut[k++] = (uint16_t)i;
    const uint16_t mx = (uint16_t)(k - 1);
    std::vector<uint16_t> buf = huf_uncompress(comp + p, (size_t)length, total);
    std::vector<size_t> start(sizes.size());
    size_t o = 0;
    for (size_t c = 0; c < sizes.size(); c++) {
        start[c] = o;
        for (int j = 0; j < sizes[c]; j++) wav2_decode(buf.data() + o + j, (int)w, sizes[c], (int)n_lines, (int)w * sizes[c], mx);
        o += (size_t)sizes[c] * w * n_lines;
    }
    for (auto& v : buf) v = lut[v];
    std::vector<uint8_t> out(total * 2);
    size_t q = 0;
    for (uint32_t y = 0; y < n_lines; y++)
        for (size_t c = 0; c < sizes.size(); c++) {
            const uint16_t* row = buf.data() + start[c] + (size_t)y * w * sizes[c];
            for (size_t i = 0; i < (size_t)w * sizes[c]; i++) {
                out[q++] = (uint8_t)(row[i] & 0xFF);
                out[q++] = (uint8_t)(row[i] >> 8);
            }
        }
    return out;
}
}  // namespace piz
inline std::vector<uint8_t> exr_rle_expand(const uint8_t* in, size_t n, size_t want) {  // ImfRle rleUncompress
    std::vector<uint8_t> out;
    out.reserve(want);
    for (size_t p = 0; p < n;) {
        const int c = (int8_t)in[p++];
        if (c < 0) {
            if (p + (size_t)(-c) > n) throw std::runtime_error("EXR RLE: truncated block");
            out.insert(out.end(), in + p, in + p + (-c));
            p += (size_t)(-c);
        } else {
            if (p >= n) throw std::runtime_error("EXR RLE: truncated block");
            out.insert(out.end(), (size_t)c + 1, in[p++]);
        }
    }
    return out;
}
inline SkyImage read_exr(const std::string& path) {
    std::vector<uint8_t> d = read_file(path);
    auto le32 = [&](size_t p) -> uint32_t {
        if (p + 4 > d.size()) throw std::runtime_error(path + ": truncated EXR");
        return (uint32_t)d[p] | ((uint32_t)d[p + 1] << 8) | ((uint32_t)d[p + 2] << 16) | ((uint32_t)d[p + 3] << 24);
    };
    auto le64 = [&](size_t p) -> uint64_t { return (uint64_t)le32(p) | ((uint64_t)le32(p + 4) << 32); };
    if (le32(0) != 20000630u) throw std::runtime_error(path + ": not an OpenEXR file");
    if (le32(4) & 0x200u) throw std::runtime_error(path + ": tiled EXR is not supported");
    size_t p = 8;
    auto cstr = [&](size_t& q, size_t limit) -> std::string {  // NUL-terminated string inside [q, limit)
        size_t e = q;
        while (e < limit && d[e] != 0) e++;
        if (e >= limit) throw std::runtime_error(path + ": unterminated string in the EXR header");
        std::string r((const char*)&d[q], e - q);
        q = e + 1;
        return r;
    };
    std::map<std::string, std::pair<size_t, size_t>> attrs;  // name -> (offset, length)
    while (p < d.size() && d[p] != 0) {
        const std::string name = cstr(p, d.size());
        cstr(p, d.size());  // the attribute's type name
        const uint32_t ln = le32(p);
        p += 4;
        if (p + (size_t)ln > d.size()) throw std::runtime_error(path + ": EXR attribute exceeds the file");
        attrs[name] = {p, ln};
        p += ln;
    }
    p += 1;
    for (const char* need : {"compression", "channels", "dataWindow"})
        if (!attrs.count(need)) throw std::runtime_error(path + ": EXR header lacks '" + need + "'");
    const int comp = d[attrs["compression"].first];
    if (comp < 0 || comp > 4)
        throw std::runtime_error(path + ": EXR compression " + std::to_string(comp) + " is not supported (uncompressed, RLE, ZIPS, ZIP and PIZ are; PXR24 / B44 / DWA are not)");
    const uint32_t lines_per_block = comp == 3 ? 16u : (comp == 4 ? 32u : 1u);
    struct Chan { std::string name; int type; };
    std::vector<Chan> chans;
    if (attrs["compression"].second < 1 || attrs["dataWindow"].second < 16) throw std::runtime_error(path + ": short EXR attribute");
    for (size_t q = attrs["channels"].first, qe = q + attrs["channels"].second; q < qe && d[q] != 0;) {
        const std::string nm = cstr(q, qe);
        if (q + 16 > qe) throw std::runtime_error(path + ": truncated EXR channel list");
        chans.push_back({nm, (int)le32(q)});
        q += 16;
    }
    if (chans.empty() || chans.size() > 64) throw std::runtime_error(path + ": bad EXR channel list");
    const size_t dw = attrs["dataWindow"].first;
    const int32_t x0 = (int32_t)le32(dw), y0 = (int32_t)le32(dw + 4), x1 = (int32_t)le32(dw + 8), y1 = (int32_t)le32(dw + 12);
    if (x1 < x0 || y1 < y0 || (int64_t)x1 - x0 >= 65536 || (int64_t)y1 - y0 >= 65536) throw std::runtime_error(path + ": bad EXR data window");
    SkyImage img;
    img.w = (uint32_t)(x1 - x0 + 1);
    img.h = (uint32_t)(y1 - y0 + 1);
    img.rgb.assign((size_t)img.w * img.h * 3, 0.0f);
    size_t line_bytes = 0;
    for (auto& c : chans) line_bytes += (c.type == 1 ? 2u : 4u) * (size_t)img.w;
    const uint32_t n_blocks = (img.h + lines_per_block - 1) / lines_per_block;
    for (uint32_t b = 0; b < n_blocks; b++) {
        const size_t o = (size_t)le64(p + 8 * (size_t)b);
        const int32_t y = (int32_t)le32(o);
        const uint32_t size = le32(o + 4);
        if (y < y0 || y > y1) throw std::runtime_error(path + ": EXR block outside the data window");
        const uint32_t n_lines = std::min<uint32_t>(lines_per_block, (uint32_t)(y1 + 1 - y));
        if (o + 8 + (size_t)size > d.size()) throw std::runtime_error(path + ": EXR block exceeds the file");
        std::vector<uint8_t> block(d.begin() + (long)(o + 8), d.begin() + (long)(o + 8 + size));
        const size_t want = line_bytes * n_lines;
        if (comp == 4 && size < want) {  // a block that does not shrink is stored raw
            std::vector<int> sizes;
            for (auto& c : chans) sizes.push_back(c.type == 1 ? 1 : 2);
            block = piz::unpack(block.data(), block.size(), sizes, img.w, n_lines);
        } else if (comp && size < want) {  // deflate / RLE -> undo the delta predictor -> re-interleave the two byte halves
            std::vector<uint8_t> t = comp == 1 ? exr_rle_expand(block.data(), block.size(), want) : inflate_all(block.data(), block.size(), want);
            if (t.size() != want) throw std::runtime_error(path + ": EXR ZIP / RLE block has the wrong size");
            for (size_t k = 1; k < want; k++) t[k] = (uint8_t)(t[k - 1] + t[k] - 128);
            block.resize(want);
            const size_t half = (want + 1) / 2;
            for (size_t k = 0; k < want; k++) block[k] = (k & 1) ? t[half + k / 2] : t[k / 2];
        }
        if (block.size() < want) throw std::runtime_error(path + ": EXR block is too short");
        size_t q = 0;
        for (uint32_t ly = 0; ly < n_lines; ly++)
            for (auto& c : chans) {
                const int ci = c.name == "R" ? 0 : c.name == "G" ? 1 : c.name == "B" ? 2 : -1;
                for (uint32_t x = 0; x < img.w; x++) {
                    float v;
                    if (c.type == 1) {
                        v = half_to_float((uint16_t)(block[q] | (block[q + 1] << 8)));
                        q += 2;
                    } else {
                        const uint32_t u = (uint32_t)block[q] | ((uint32_t)block[q + 1] << 8) | ((uint32_t)block[q + 2] << 16) | ((uint32_t)block[q + 3] << 24);
                        q += 4;
                        if (c.type == 2) std::memcpy(&v, &u, 4);
                        else v = (float)u;
                    }
                    if (ci >= 0) img.rgb[3 * ((size_t)(y - y0 + (int32_t)ly) * img.w + x) + ci] = v;
                }
            }
    }
    return img;
}

// ------------------------------------------------------------------------------------------------ bincode processed-asset cache
// bincode 2 `standard().with_variable_int_encoding().with_big_endian()` (assets/mod.rs:135-137); Mesh field order :118-125
struct Meshlet { uint32_t vertex_offset, triangle_offset, vertex_count, triangle_count; };
struct ProcessedMesh {
    std::vector<Meshlet> meshlets;
    std::vector<Material> materials;
    std::vector<float> vertices;   // n x 8
    std::vector<uint8_t> indices;  // meshlet-local
    bool uploaded = false;
};
class Bincode {
  public:
    explicit Bincode(std::vector<uint8_t> d) : d_(std::move(d)) {}
    const uint8_t* take(size_t n) {
        if (p_ + n > d_.size()) throw std::runtime_error("bincode: unexpected end of data");
        const uint8_t* r = d_.data() + p_;
        p_ += n;
        return r;
    }
    uint64_t varint() {
        const uint8_t b = *take(1);
        if (b < 251) return b;
        const size_t n = b == 251 ? 2 : b == 252 ? 4 : b == 253 ? 8 : 16;
        const uint8_t* q = take(n);
        uint64_t v = 0;
        for (size_t k = n > 8 ? n - 8 : 0; k < n; k++) v = (v << 8) | q[k];
        return v;
    }
    float f16() {
        const uint8_t* q = take(2);
        return half_to_float((uint16_t)((q[0] << 8) | q[1]));
    }
    bool done() const { return p_ == d_.size(); }
    size_t remaining() const { return d_.size() - p_; }

  private:
    std::vector<uint8_t> d_;
    size_t p_ = 0;
};
// layout "current" = assets/mod.rs:118-125; "old" = the file the reference tree still ships (imported_assets/Default/box.glb):
// meshlets, meshlet-local indices, materials, vertices, uploaded.  The reference's Material decoder re-reads the metallic
// bytes as roughness (assets/mod.rs:88,110); that bug is not reproduced.
inline ProcessedMesh read_processed_mesh(const std::string& path, bool old_layout = false) {
    Bincode r(read_file(path));
    ProcessedMesh pm;
    auto meshlets = [&] {
        const uint64_t n = r.varint();
        for (uint64_t k = 0; k < n; k++) {
            Meshlet m;
            m.vertex_offset = (uint32_t)r.varint();
            m.triangle_offset = (uint32_t)r.varint();
            m.vertex_count = (uint32_t)r.varint();
            m.triangle_count = (uint32_t)r.varint();
            pm.meshlets.push_back(m);
        }
    };
    auto materials = [&] {
        const uint64_t n = r.varint();
        for (uint64_t k = 0; k < n; k++) {
            Material m;
            m.metalic_factor = r.f16();
            m.roughness_factor = r.f16();
            for (int c = 0; c < 3; c++) m.color[c] = r.f16();
            const uint64_t tex = r.varint();
            m.texture_offset = tex == 0xFFFF ? -1 : (int32_t)tex;
            pm.materials.push_back(m);
        }
    };
    auto vertices = [&] {
        const uint64_t n = r.varint();
        const uint8_t* q = r.take(32 * (size_t)n);
        pm.vertices.resize(8 * (size_t)n);
        for (size_t k = 0; k < 8 * (size_t)n; k++) {
            const uint32_t u = be32(q + 4 * k);
            std::memcpy(&pm.vertices[k], &u, 4);
        }
    };
    auto bytes_vec = [&] {
        const uint64_t n = r.varint();
        const uint8_t* q = r.take((size_t)n);
        pm.indices.assign(q, q + n);
    };
    if (!old_layout) { meshlets(); materials(); vertices(); bytes_vec(); }
    else { meshlets(); bytes_vec(); materials(); vertices(); }
    pm.uploaded = *r.take(1) != 0;
    if (!r.done()) throw std::runtime_error("bincode: " + std::to_string(r.remaining()) + " trailing bytes (wrong layout?)");
    return pm;
}

// ------------------------------------------------------------------------------------------------ upload through the C ABI
// world/mod.rs:83-101 (`loaded_assets` -> DynamicBuffer::push x3) + init_world's create_acceleration_structure
inline void check(rt3_ctx* ctx, int rc, const char* what) {
    if (rc != RT3_OK) throw std::runtime_error(std::string(what) + " failed: " + rt3_last_error(ctx));
}
inline uint32_t upload(rt3_ctx* ctx, const Mesh& m) {
    check(ctx, rt3_scene_set_vertices(ctx, m.vertices.data(), (uint32_t)m.n_vertices()), "rt3_scene_set_vertices");
    check(ctx, rt3_scene_set_indices(ctx, m.indices.data(), (uint32_t)m.indices.size()), "rt3_scene_set_indices");
    check(ctx, rt3_scene_set_geometry(ctx, m.geometries.data(), m.prim_counts.data(), (uint32_t)m.geometries.size()), "rt3_scene_set_geometry");
    for (size_t i = 0; i < m.textures.size(); i++)
        check(ctx, rt3_scene_set_texture(ctx, (uint32_t)i, m.textures[i].rgba.data(), m.textures[i].w, m.textures[i].h), "rt3_scene_set_texture");
    uint32_t handle = 0;
    check(ctx, rt3_accel_build(ctx, &handle), "rt3_accel_build");
    return handle;
}
// rt3_scene_set_sky wants finite, non-negative radiance; HDR files do carry +inf sun texels, NaNs and slightly negative values, so
// the upload helper clamps them (NaN -> 0, negative -> 0, inf -> HALF max) and says how many it touched
inline size_t sanitize_sky(SkyImage& s) {
    size_t touched = 0;
    for (float& v : s.rgb)
        if (!(v >= 0.0f && v <= 65504.0f)) {
            v = v > 65504.0f ? 65504.0f : 0.0f;  // NaN compares false everywhere -> 0
            touched++;
        }
    return touched;
}
inline void upload_sky(rt3_ctx* ctx, SkyImage s) {
    if (const size_t n = sanitize_sky(s)) fprintf(stderr, "assets: clamped %zu sky values outside [0, 65504]\n", n);
    check(ctx, rt3_scene_set_sky(ctx, s.rgb.data(), s.w, s.h), "rt3_scene_set_sky");
}

}  // namespace rt3::assets
