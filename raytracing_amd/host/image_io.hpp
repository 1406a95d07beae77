// raytracing_amd/host/image_io.hpp -- image read/write for the host driver.
// The reference uses the vendored stb_image / stb_image_write (texture.hpp:87, raytracing.cpp:124-125); those are
// third-party sources this repository does not carry, so the two calls the renderer needs are provided here: decoders for every
// format stb_image loads -- PNG (on zlib; every colour type and bit depth, Adam7 included), BMP, GIF (first frame), PSD, PIC, JPEG
// (jpeg_decode.hpp: baseline and progressive), binary PGM / PPM, Radiance HDR and TGA -- and a PNG encoder.  Decoded texels are the ones stb_image hands the reference -- rows top
// to bottom, channels interleaved, as many channels as stb reports for the file (the `0` of stbi_load(path, &w, &h, &ch, 0)),
// its conversions and quirks included; tests/golden/images.npz holds the reference loader's output for 86 files and the host
// tests compare byte for byte.
#pragma once

#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "jpeg_decode.hpp"

inline bool rt_read_file(const char* path, std::vector<unsigned char>& out)
{
    FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? (size_t) n : 0);
    bool ok = n >= 0 && std::fread(out.data(), 1, out.size(), f) == out.size();
    std::fclose(f);
    return ok;
}

inline uint32_t rt_be32(const unsigned char* p) { return ((uint32_t) p[0] << 24) | ((uint32_t) p[1] << 16) | ((uint32_t) p[2] << 8) | p[3]; }

// PNG as stb_image hands it to an 8-bit caller: every colour type and bit depth of the standard, Adam7 interlace included; samples
// of 1 / 2 / 4 bits scaled to 0..255 (x255, x85, x17), 16-bit samples cut to their high byte; a palette becomes RGB, or RGBA when
// the file has a tRNS chunk; a grey or RGB file with a tRNS colour key gains an alpha channel of 0 / 255 (the key is compared at
// the file's own depth).  tests/golden/images.npz pins it against the reference's loader.
inline bool rt_decode_png(const std::vector<unsigned char>& file, std::vector<unsigned char>& pixels, int& w, int& h, int& ch)
{
    static const unsigned char sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    if (file.size() < 33 || std::memcmp(file.data(), sig, 8) != 0) return false;
    size_t pos = 8;
    int depth = 0, color = 0, interlace = 0;
    bool have_key = false;
    std::vector<unsigned char> idat, palette, trns;
    while (pos + 12 <= file.size()) {
        uint32_t len = rt_be32(&file[pos]);
        const unsigned char* type = &file[pos + 4];
        const unsigned char* body = &file[pos + 8];
        if (pos + 12 + (size_t) len > file.size()) return false;
        if (!std::memcmp(type, "IHDR", 4) && len >= 13) { w = (int) rt_be32(body); h = (int) rt_be32(body + 4); depth = body[8]; color = body[9]; interlace = body[12]; }
        else if (!std::memcmp(type, "PLTE", 4)) palette.assign(body, body + len);
        else if (!std::memcmp(type, "tRNS", 4)) { trns.assign(body, body + len); have_key = true; }
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!std::memcmp(type, "IEND", 4)) break;
        pos += 12 + (size_t) len;
    }
    int src_ch = color == 0 ? 1 : color == 2 ? 3 : color == 3 ? 1 : color == 4 ? 2 : color == 6 ? 4 : 0;
    bool depth_ok = color == 0 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)
                  : color == 3 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8) : (depth == 8 || depth == 16);
    if (!src_ch || !depth_ok || interlace > 1 || w <= 0 || h <= 0 || (uint64_t) w * h > (1ull << 28)) return false;
    if (color == 3 && palette.empty()) return false;
    if (color == 4 || color == 6) have_key = false;          // files with an alpha channel carry no key

    // the passes of the picture: one, or Adam7's seven sub-pictures
    struct pass { int x0, y0, dx, dy; };
    static const pass whole[1] = { { 0, 0, 1, 1 } };
    static const pass adam7[7] = { { 0, 0, 8, 8 }, { 4, 0, 8, 8 }, { 0, 4, 4, 8 }, { 2, 0, 4, 4 }, { 0, 2, 2, 4 }, { 1, 0, 2, 2 }, { 0, 1, 1, 2 } };
    const pass* passes = interlace ? adam7 : whole;
    int n_pass = interlace ? 7 : 1;
    int bits_px = src_ch * depth, filter_bytes = bits_px >= 8 ? bits_px / 8 : 1;
    size_t total = 0;
    for (int p = 0; p < n_pass; p++) {
        int pw = (w - passes[p].x0 + passes[p].dx - 1) / passes[p].dx, ph = (h - passes[p].y0 + passes[p].dy - 1) / passes[p].dy;
        if (pw > 0 && ph > 0) total += (((size_t) pw * bits_px + 7) / 8 + 1) * ph;
    }
    if (total > (size_t) idat.size() * 1100 + 4096) return false;      // deflate expands at most ~1030 : 1; a header that promises more is not believed
    std::vector<unsigned char> raw(total);
    uLongf raw_len = (uLongf) raw.size();
    int zr = uncompress(raw.data(), &raw_len, idat.data(), (uLong) idat.size());
    if ((zr != Z_OK && zr != Z_BUF_ERROR) || raw_len < raw.size()) return false;     // stb tolerates bytes after the last row

    std::vector<uint16_t> samples((size_t) w * h * src_ch);                          // as stored, before any scaling
    size_t at = 0;
    std::vector<unsigned char> prev, cur;
    for (int p = 0; p < n_pass; p++) {
        int pw = (w - passes[p].x0 + passes[p].dx - 1) / passes[p].dx, ph = (h - passes[p].y0 + passes[p].dy - 1) / passes[p].dy;
        if (pw <= 0 || ph <= 0) continue;
        size_t stride = ((size_t) pw * bits_px + 7) / 8;
        prev.assign(stride, 0);
        cur.assign(stride, 0);
        for (int y = 0; y < ph; y++) {
            const unsigned char* in = &raw[at];
            at += stride + 1;
            int filter = in[0];
            if (filter > 4) return false;
            for (size_t x = 0; x < stride; x++) {
                int a = x >= (size_t) filter_bytes ? cur[x - filter_bytes] : 0, b = prev[x], c = x >= (size_t) filter_bytes ? prev[x - filter_bytes] : 0;
                int v = in[1 + x];
                switch (filter) {
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: { int q = a + b - c, pa = std::abs(q - a), pb = std::abs(q - b), pc = std::abs(q - c); v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
                default: break;
                }
                cur[x] = (unsigned char) v;
            }
            uint16_t* out_row = &samples[((size_t) (passes[p].y0 + y * passes[p].dy) * w) * src_ch];
            for (int x = 0; x < pw; x++) {
                uint16_t* out = out_row + (size_t) (passes[p].x0 + x * passes[p].dx) * src_ch;
                for (int k = 0; k < src_ch; k++) {
                    size_t i = (size_t) x * src_ch + k;
                    if (depth == 8) out[k] = cur[i];
                    else if (depth == 16) out[k] = (uint16_t) ((cur[2 * i] << 8) | cur[2 * i + 1]);
                    else { size_t bit = i * depth; out[k] = (uint16_t) ((cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1)); }
                }
            }
            prev.swap(cur);
        }
    }

    size_t n_px = (size_t) w * h;
    if (color == 3) {
        ch = have_key ? 4 : 3;
        pixels.resize(n_px * ch);
        for (size_t i = 0; i < n_px; i++) {
            size_t k = samples[i];
            for (int c = 0; c < 3; c++) pixels[i * ch + c] = k * 3 + c < palette.size() ? palette[k * 3 + c] : 0;
            if (have_key) pixels[i * 4 + 3] = k < trns.size() ? trns[k] : 255;
        }
        return true;
    }
    int scale = depth == 1 ? 255 : depth == 2 ? 85 : depth == 4 ? 17 : 1;
    if (have_key && trns.size() < (size_t) src_ch * 2) have_key = false;
    ch = src_ch + (have_key ? 1 : 0);
    pixels.resize(n_px * ch);
    for (size_t i = 0; i < n_px; i++) {
        bool is_key = have_key;
        for (int k = 0; k < src_ch; k++) {
            uint16_t v = samples[i * src_ch + k];
            unsigned key = (unsigned) ((trns.size() >= (size_t) k * 2 + 2) ? (trns[2 * k] << 8) | trns[2 * k + 1] : 0);
            if (depth == 16) { is_key = is_key && v == key; pixels[i * ch + k] = (unsigned char) (v >> 8); }
            else {
                unsigned char v8 = (unsigned char) (v * scale);
                is_key = is_key && v8 == (unsigned char) ((key & 255) * scale);
                pixels[i * ch + k] = v8;
            }
        }
        if (have_key) pixels[i * ch + src_ch] = is_key ? 0 : 255;
    }
    return true;
}

// Binary PGM / PPM (P5 / P6) as stb reads them: the samples as stored (a maximum below 255 does not rescale them); of a two-byte
// sample (maximum above 255) the SECOND byte survives -- stb loads the big-endian pairs as host (little-endian) words before it
// takes their high byte, and the fixture pins that.
inline bool rt_decode_ppm(const std::vector<unsigned char>& file, std::vector<unsigned char>& pixels, int& w, int& h, int& ch)
{
    if (file.size() < 8 || file[0] != 'P' || (file[1] != '5' && file[1] != '6')) return false;
    size_t pos = 2;
    int vals[3], got = 0;
    while (got < 3 && pos < file.size()) {
        while (pos < file.size() && (file[pos] == ' ' || file[pos] == '\n' || file[pos] == '\r' || file[pos] == '\t' || file[pos] == '\v' || file[pos] == '\f')) pos++;
        if (pos < file.size() && file[pos] == '#') { while (pos < file.size() && file[pos] != '\n' && file[pos] != '\r') pos++; continue; }
        int v = 0; bool any = false;
        while (pos < file.size() && file[pos] >= '0' && file[pos] <= '9') { if (v < (1 << 24)) v = v * 10 + (file[pos] - '0'); pos++; any = true; }
        if (!any) return false;
        vals[got++] = v;
    }
    pos++;
    if (got < 3) return false;
    w = vals[0]; h = vals[1]; ch = file[1] == '6' ? 3 : 1;
    int bytes = vals[2] > 255 ? 2 : 1;
    if (vals[2] < 1 || vals[2] > 65535 || w <= 0 || h <= 0 || (uint64_t) w * h > (1ull << 28)) return false;
    size_t n = (size_t) w * h * ch;
    if (pos + n * bytes > file.size()) return false;
    pixels.resize(n);
    for (size_t i = 0; i < n; i++) pixels[i] = file[pos + i * bytes + (bytes - 1)];
    return true;
}

// little-endian fields of BMP and TGA headers; reads past the end give 0, as stb's do
struct rt_le_reader {
    const std::vector<unsigned char>& f;
    size_t pos = 0;
    explicit rt_le_reader(const std::vector<unsigned char>& file) : f(file) {}
    int u8() { return pos < f.size() ? f[pos++] : (pos++, 0); }
    int u16() { int lo = u8(); return lo | (u8() << 8); }
    uint32_t u32() { uint32_t lo = (uint32_t) u16(); return lo | ((uint32_t) u16() << 16); }
    void skip(long long n) { if (n > 0) pos += (size_t) n; }
};

// Windows / OS/2 bitmaps as stb reads them: 1 / 4 / 8-bit palettes, 16 / 24 / 32-bit pixels with the default or BI_BITFIELDS masks
// (a channel of n < 8 bits is widened by bit replication), bottom-up or top-down rows; no RLE.  Three channels, or four when the
// file has an alpha mask (an alpha plane that is zero everywhere reads as opaque).
inline bool rt_decode_bmp(const std::vector<unsigned char>& file, std::vector<unsigned char>& pixels, int& w, int& h, int& ch)
{
    if (file.size() < 26 || file[0] != 'B' || file[1] != 'M') return false;
    rt_le_reader r(file);
    r.skip(10);
    long long offset = (int32_t) r.u32();
    int hsz = (int) r.u32(), extra = 14;
    if (offset < 0 || (hsz != 12 && hsz != 40 && hsz != 56 && hsz != 108 && hsz != 124)) return false;
    long long iw, ih;
    if (hsz == 12) { iw = r.u16(); ih = r.u16(); } else { iw = (int32_t) r.u32(); ih = (int32_t) r.u32(); }
    if (r.u16() != 1) return false;
    int bpp = r.u16(), compress = 0;
    uint32_t mr = 0, mg = 0, mb = 0, ma = 0;
    bool alpha_seen = true;                      // only the default 32-bit layout starts from "no alpha seen yet"
    auto default_masks = [&]() {
        if (bpp == 16) { mr = 31u << 10; mg = 31u << 5; mb = 31u; }
        else if (bpp == 32) { mr = 0xffu << 16; mg = 0xffu << 8; mb = 0xffu; ma = 0xffu << 24; alpha_seen = false; }
        else mr = mg = mb = ma = 0;
    };
    if (hsz != 12) {
        compress = (int) r.u32();
        if (compress == 1 || compress == 2 || compress >= 4 || compress < 0) return false;
        if (compress == 3 && bpp != 16 && bpp != 32) return false;
        r.skip(20);
        if (hsz == 40 || hsz == 56) {
            if (hsz == 56) r.skip(16);
            if (bpp == 16 || bpp == 32) {
                if (compress == 0) default_masks();
                else { mr = r.u32(); mg = r.u32(); mb = r.u32(); extra += 12; if (mr == mg && mg == mb) return false; }
            }
        }
        else {
            mr = r.u32(); mg = r.u32(); mb = r.u32(); ma = r.u32();
            if (compress != 3) default_masks();
            r.skip(hsz == 124 ? 68 : 52);
        }
    }
    bool bottom_up = ih > 0;
    if (ih < 0) ih = -ih;
    if (iw <= 0 || ih <= 0 || iw > (1 << 24) || ih > (1 << 24) || (uint64_t) iw * ih > (1ull << 28)) return false;
    w = (int) iw; h = (int) ih;
    if ((uint64_t) w * h * (uint64_t) bpp / 8 > (uint64_t) file.size()) return false;   // the pixels have to be in the file
    int psize = 0;
    if (hsz == 12) { if (bpp < 24) psize = (int) ((offset - extra - 24) / 3); }
    else if (bpp < 16) psize = (int) ((offset - extra - hsz) >> 2);
    if (psize == 0 && offset != (long long) r.pos) return false;
    ch = (bpp == 24 && ma == 0xff000000u) ? 3 : (ma ? 4 : 3);
    pixels.assign((size_t) w * h * ch, 0);
    size_t z = 0;
    if (bpp < 16) {
        if (psize <= 0 || psize > 256 || (bpp != 1 && bpp != 4 && bpp != 8)) return false;
        unsigned char pal[256][3] = { { 0 } };
        for (int i = 0; i < psize; i++) { pal[i][2] = (unsigned char) r.u8(); pal[i][1] = (unsigned char) r.u8(); pal[i][0] = (unsigned char) r.u8(); if (hsz != 12) r.u8(); }
        r.skip(offset - extra - hsz - (long long) psize * (hsz == 12 ? 3 : 4));
        int row_bytes = bpp == 1 ? (w + 7) >> 3 : bpp == 4 ? (w + 1) >> 1 : w, pad = (-row_bytes) & 3;
        for (int j = 0; j < h; j++) {
            size_t row_at = r.pos;
            for (int i = 0; i < w; i++) {
                size_t bit = (size_t) i * bpp;
                size_t at = row_at + (bit >> 3);
                int v = at < file.size() ? file[at] : 0;
                int index = (v >> (8 - bpp - (int) (bit & 7))) & ((1 << bpp) - 1);
                for (int c = 0; c < 3; c++) pixels[z++] = pal[index][c];
                if (ch == 4) pixels[z++] = 255;
            }
            r.pos = row_at + row_bytes + pad;
        }
    }
    else {
        if (bpp != 16 && bpp != 24 && bpp != 32) return false;
        r.skip(offset - extra - hsz);
        int pad = bpp == 24 ? (-3 * w) & 3 : bpp == 16 ? (-2 * w) & 3 : 0;
        bool plain = bpp == 24 || (bpp == 32 && mb == 0xff && mg == 0xff00 && mr == 0x00ff0000 && ma == 0xff000000u);
        auto top_bit = [](uint32_t m) { int n = -1; while (m) { n++; m >>= 1; } return n; };
        auto count = [](uint32_t m) { int n = 0; while (m) { n += m & 1; m >>= 1; } return n; };
        auto widen = [](uint32_t v, int shift, int bits) {       // the field's top bit to bit 7, then the n-bit value replicated to 8
            static const unsigned mul[9] = { 0, 0xff, 0x55, 0x49, 0x11, 0x21, 0x41, 0x81, 0x01 }, down[9] = { 0, 0, 0, 1, 0, 2, 4, 6, 0 };
            v = shift < 0 ? v << -shift : v >> shift;
            v >>= (8 - bits);
            return (int) ((v * mul[bits]) >> down[bits]);
        };
        int rs = 0, gs = 0, bs = 0, as = 0, rc = 0, gc = 0, bc = 0, ac = 0;
        if (!plain) {
            if (!mr || !mg || !mb) return false;
            rs = top_bit(mr) - 7; gs = top_bit(mg) - 7; bs = top_bit(mb) - 7; as = top_bit(ma) - 7;
            rc = count(mr); gc = count(mg); bc = count(mb); ac = count(ma);
            if (rc > 8 || gc > 8 || bc > 8 || ac > 8) return false;
        }
        unsigned all_alpha = alpha_seen ? 255 : 0;
        for (int j = 0; j < h; j++) {
            for (int i = 0; i < w; i++) {
                unsigned alpha;
                if (plain) {
                    pixels[z + 2] = (unsigned char) r.u8(); pixels[z + 1] = (unsigned char) r.u8(); pixels[z] = (unsigned char) r.u8();
                    z += 3;
                    alpha = bpp == 32 ? (unsigned) r.u8() : 255;
                }
                else {
                    uint32_t v = bpp == 16 ? (uint32_t) r.u16() : r.u32();
                    pixels[z++] = (unsigned char) widen(v & mr, rs, rc);
                    pixels[z++] = (unsigned char) widen(v & mg, gs, gc);
                    pixels[z++] = (unsigned char) widen(v & mb, bs, bc);
                    alpha = ma ? (unsigned) widen(v & ma, as, ac) : 255;
                }
                all_alpha |= alpha;
                if (ch == 4) pixels[z++] = (unsigned char) alpha;
            }
            r.skip(pad);
        }
        if (ch == 4 && all_alpha == 0) for (size_t i = 3; i < pixels.size(); i += 4) pixels[i] = 255;
    }
    if (bottom_up) {
        size_t stride = (size_t) w * ch;
        for (int j = 0; j < h / 2; j++) std::swap_ranges(&pixels[stride * j], &pixels[stride * j] + stride, &pixels[stride * (h - 1 - j)]);
    }
    return true;
}

// Truevision TGA as stb reads it: true-colour 15 / 16 / 24 / 32 bits, grey 8 / 16 bits, colour-mapped with 8 or 16-bit indices, each
// raw or run-length packed, either row order; 15 / 16-bit colours become RGB through x*255/31 with the top bit ignored.  The format
// has no signature, so this reader comes last and checks the header fields stb checks.
inline bool rt_decode_tga(const std::vector<unsigned char>& file, std::vector<unsigned char>& pixels, int& w, int& h, int& ch)
{
    if (file.size() < 18) return false;
    rt_le_reader r(file);
    int id_len = r.u8(), mapped = r.u8(), kind = r.u8();
    int map_first = r.u16(), map_len = r.u16(), map_bits = r.u8();
    r.skip(4);
    w = r.u16(); h = r.u16();
    int bits = r.u8(), descriptor = r.u8();
    if (mapped > 1 || w < 1 || h < 1) return false;
    if ((uint64_t) w * h > (uint64_t) file.size() * 128) return false;       // a run packs at most 128 pixels into two bytes
    if (mapped) { if ((kind != 1 && kind != 9) || (bits != 8 && bits != 16)) return false; }
    else if (kind != 2 && kind != 3 && kind != 10 && kind != 11) return false;
    bool packed = kind >= 8, grey = (kind & 7) == 3, top_down = (descriptor >> 5) & 1;
    int depth = mapped ? map_bits : bits;
    bool rgb16 = depth == 15 || (depth == 16 && !(grey && !mapped));
    ch = depth == 8 ? 1 : (depth == 16 && grey && !mapped) ? 2 : rgb16 ? 3 : (depth == 24 || depth == 32) ? depth / 8 : 0;
    if (!ch) return false;
    r.skip(id_len);
    auto read_px = [&](unsigned char* out, bool as16) {
        if (as16) { int px = r.u16(); out[0] = (unsigned char) ((((px >> 10) & 31) * 255) / 31); out[1] = (unsigned char) ((((px >> 5) & 31) * 255) / 31); out[2] = (unsigned char) (((px & 31) * 255) / 31); }
        else for (int c = 0; c < ch; c++) out[c] = (unsigned char) r.u8();
    };
    std::vector<unsigned char> palette;
    if (mapped) {
        if (map_len == 0) return false;
        r.skip(map_first);
        palette.resize((size_t) map_len * ch);
        if (!rgb16 && r.pos + palette.size() > file.size()) return false;
        for (int i = 0; i < map_len; i++) read_px(&palette[(size_t) i * ch], rgb16);
    }
    size_t n_px = (size_t) w * h;
    pixels.assign(n_px * ch, 0);
    unsigned char px[4] = { 0, 0, 0, 0 };
    int run = 0;
    bool repeating = false;
    for (size_t i = 0; i < n_px; i++) {
        bool fetch = true;
        if (packed) {
            if (run == 0) { int cmd = r.u8(); run = 1 + (cmd & 127); repeating = cmd >> 7; }
            else if (repeating) fetch = false;
        }
        if (fetch) {
            if (mapped) {
                int index = bits == 8 ? r.u8() : r.u16();
                if (index >= map_len) index = 0;
                for (int c = 0; c < ch; c++) px[c] = palette[(size_t) index * ch + c];
            }
            else read_px(px, rgb16);
        }
        for (int c = 0; c < ch; c++) pixels[i * ch + c] = px[c];
        run--;
    }
    if (!top_down) {
        size_t stride = (size_t) w * ch;
        for (int j = 0; j * 2 < h; j++) if (j != h - 1 - j) std::swap_ranges(&pixels[stride * j], &pixels[stride * j] + stride, &pixels[stride * (h - 1 - j)]);
    }
    if (ch >= 3 && !rgb16) for (size_t i = 0; i < n_px; i++) std::swap(pixels[i * ch], pixels[i * ch + 2]);     // stored blue first
    return true;
}

// GIF as stb reads it for a still picture: the first frame only, always four channels.  Pixels of the transparent index and pixels
// the frame does not cover stay (0,0,0,0) -- unless the header names a background index above 0, in which case the uncovered ones
// take that palette entry, opaque, and with red and blue exchanged (stb copies its blue-first palette entry as it lies; the
// fixture pins it).  LZW per the GIF89a specification, interlaced frames included.
inline bool rt_decode_gif(const std::vector<unsigned char>& file, std::vector<unsigned char>& pixels, int& w, int& h, int& ch)
{
    if (file.size() < 13 || std::memcmp(file.data(), "GIF8", 4) != 0 || (file[4] != '7' && file[4] != '9') || file[5] != 'a') return false;
    rt_le_reader r(file);
    r.skip(6);
    w = r.u16(); h = r.u16();
    int flags = r.u8(), background = r.u8();
    r.u8();
    if (w <= 0 || h <= 0 || (uint64_t) w * h > (1ull << 28) || (uint64_t) w * h > (uint64_t) file.size() * 4096) return false;   // (LZW packs at most a few thousand pixels into a code)
    struct entry { unsigned char r, g, b, a; };
    std::vector<entry> global(256, entry{ 0, 0, 0, 0 }), local(256, entry{ 0, 0, 0, 0 });
    auto read_table = [&](std::vector<entry>& t, int n, int transparent) {
        for (int i = 0; i < n; i++) { t[i].r = (unsigned char) r.u8(); t[i].g = (unsigned char) r.u8(); t[i].b = (unsigned char) r.u8(); t[i].a = i == transparent ? 0 : 255; }
    };
    if (flags & 0x80) read_table(global, 2 << (flags & 7), -1);
    int transparent = -1, control = 0;
    ch = 4;
    for (;;) {
        if (r.pos >= file.size()) return false;
        int tag = r.u8();
        if (tag == 0x21) {
            int kind = r.u8();
            if (kind == 0xf9) {
                int len = r.u8();
                if (len != 4) { r.skip(len); continue; }
                control = r.u8();
                r.u16();
                if (transparent >= 0) global[transparent].a = 255;
                if (control & 1) { transparent = r.u8(); global[transparent].a = 0; }
                else { r.u8(); transparent = -1; }
            }
            for (int len = r.u8(); len != 0; len = r.u8()) r.skip(len);
            continue;
        }
        if (tag != 0x2c) return false;               // a trailer before any image, or an unknown block
        int fx = r.u16(), fy = r.u16(), fw = r.u16(), fh = r.u16();
        if (fx + fw > w || fy + fh > h) return false;
        int lflags = r.u8();
        const std::vector<entry>* table;
        if (lflags & 0x80) { read_table(local, 2 << (lflags & 7), (control & 1) ? transparent : -1); table = &local; }
        else if (flags & 0x80) table = &global;
        else return false;
        pixels.assign((size_t) w * h * 4, 0);
        std::vector<unsigned char> touched((size_t) w * h, 0);
        // raster position: rows go 0, 8, 16, ... then 4, 12, ... then 2, 6, ... then 1, 3, ... when the frame is interlaced
        int x = fx, y = fy, step = (lflags & 0x40) ? 8 : 1, passes_left = (lflags & 0x40) ? 3 : 0;
        if (fw == 0) y = fy + fh;
        auto put = [&](int index) {
            if (y >= fy + fh) return;
            size_t at = (size_t) y * w + x;
            touched[at] = 1;
            const entry& e = (*table)[index];
            if (e.a > 128) { pixels[at * 4] = e.r; pixels[at * 4 + 1] = e.g; pixels[at * 4 + 2] = e.b; pixels[at * 4 + 3] = e.a; }
            if (++x >= fx + fw) {
                x = fx;
                y += step;
                while (y >= fy + fh && passes_left > 0) { step = 1 << passes_left; y = fy + (step >> 1); passes_left--; }
            }
        };
        int min_size = r.u8();
        if (min_size > 12) return false;
        const int clear = 1 << min_size;
        std::vector<int16_t> prefix(8192, -1);
        std::vector<unsigned char> head(8192, 0), tail(8192, 0), chain(8192);
        for (int i = 0; i < clear; i++) head[i] = tail[i] = (unsigned char) i;
        int width = min_size + 1, mask = (1 << width) - 1, next = clear + 2, previous = -1, in_block = 0;
        uint32_t acc = 0;
        int have = 0;
        bool cleared = false, done = false;
        while (!done) {
            if (have < width) {
                if (in_block == 0) { in_block = r.u8(); if (in_block == 0) break; }
                in_block--;
                acc |= (uint32_t) r.u8() << have;
                have += 8;
                continue;
            }
            int code = (int) (acc & (uint32_t) mask);
            acc >>= width; have -= width;
            if (code == clear) { width = min_size + 1; mask = (1 << width) - 1; next = clear + 2; previous = -1; cleared = true; }
            else if (code == clear + 1) done = true;
            else if (code <= next) {
                if (!cleared) return false;
                if (previous >= 0) {
                    if (next >= 8192) return false;
                    prefix[next] = (int16_t) previous;
                    head[next] = head[previous];
                    tail[next] = code == next ? head[next] : head[code];
                    next++;
                }
                else if (code == next) return false;
                int n = 0;
                for (int c = code; c >= 0 && n < 8192; c = prefix[c]) chain[n++] = tail[c];
                while (n > 0) put(chain[--n]);
                if ((next & mask) == 0 && next <= 0x0fff) { width++; mask = (1 << width) - 1; }
                previous = code;
            }
            else return false;
        }
        if (background > 0) {
            entry e = global[background];
            for (size_t i = 0; i < touched.size(); i++)
                if (!touched[i]) { pixels[i * 4] = e.b; pixels[i * 4 + 1] = e.g; pixels[i * 4 + 2] = e.r; pixels[i * 4 + 3] = 255; }
        }
        return true;
    }
}

// big-endian fields (PSD, PIC); reads past the end give 0 and are remembered
struct rt_be_reader {
    const std::vector<unsigned char>& f;
    size_t pos = 0;
    explicit rt_be_reader(const std::vector<unsigned char>& file) : f(file) {}
    bool at_end() const { return pos >= f.size(); }
    int u8() { return pos < f.size() ? f[pos++] : (pos++, 0); }
    int u16() { int hi = u8(); return (hi << 8) | u8(); }
    uint32_t u32() { uint32_t hi = (uint32_t) u16(); return (hi << 16) | (uint32_t) u16(); }
    void skip(uint32_t n) { pos += n; }
};

// Photoshop PSD as stb reads it: the flattened RGB picture of an 8- or 16-bit file (16-bit samples cut to the high byte), raw or
// PackBits rows, always four channels (alpha 255 when the file has none); where alpha is neither 0 nor 255 the colour is "un-matted"
// from white with stb's float arithmetic, its wrap-around on out-of-range results included (the fixture pins it).
inline bool rt_decode_psd(const std::vector<unsigned char>& file, std::vector<unsigned char>& pixels, int& w, int& h, int& ch)
{
    if (file.size() < 26 || std::memcmp(file.data(), "8BPS", 4) != 0) return false;
    rt_be_reader r(file);
    r.skip(4);
    if (r.u16() != 1) return false;
    r.skip(6);
    int channels = r.u16();
    if (channels < 0 || channels > 16) return false;
    uint32_t hh = r.u32(), ww = r.u32();
    int depth = r.u16();
    if ((depth != 8 && depth != 16) || r.u16() != 3) return false;          // RGB only
    if (ww == 0 || hh == 0 || ww > (1u << 24) || hh > (1u << 24) || (uint64_t) ww * hh > (1ull << 28)) return false;
    if ((uint64_t) ww * hh > (uint64_t) file.size() * 128) return false;     // PackBits expands at most 64 : 1
    r.skip(r.u32()); r.skip(r.u32()); r.skip(r.u32());                       // mode data, image resources, layers
    int compression = r.u16();
    if (compression > 1 || r.pos > file.size()) return false;
    w = (int) ww; h = (int) hh; ch = 4;
    size_t n_px = (size_t) w * h;
    pixels.assign(n_px * 4, 0);
    if (compression) r.skip((uint32_t) h * channels * 2);                    // the rows' byte counts
    for (int c = 0; c < 4; c++) {
        unsigned char* p = &pixels[c];
        if (c >= channels) { for (size_t i = 0; i < n_px; i++) p[i * 4] = c == 3 ? 255 : 0; continue; }
        if (!compression) {
            for (size_t i = 0; i < n_px; i++) p[i * 4] = (unsigned char) (depth == 16 ? r.u16() >> 8 : r.u8());
            continue;
        }
        size_t done = 0;
        while (done < n_px) {
            int len = r.u8();
            if (r.pos > file.size()) return false;
            if (len == 128) continue;
            if (len < 128) { len++; if ((size_t) len > n_px - done) return false; for (; len; len--) p[done++ * 4] = (unsigned char) r.u8(); }
            else { len = 257 - len; if ((size_t) len > n_px - done) return false; unsigned char v = (unsigned char) r.u8(); for (; len; len--) p[done++ * 4] = v; }
        }
    }
    if (channels >= 4)
        for (size_t i = 0; i < n_px; i++) {
            unsigned char* px = &pixels[i * 4];
            if (px[3] != 0 && px[3] != 255) {
                float a = px[3] / 255.0f, ra = 1.0f / a, inv_a = 255.0f * (1 - ra);
                for (int c = 0; c < 3; c++) px[c] = (unsigned char) (int) (px[c] * ra + inv_a);
            }
        }
    return true;
}

// Softimage PIC as stb reads it: 8-bit channel packets, each raw, run-length or mixed; three channels, four when a packet carries alpha
// (a channel no packet carries stays 255).
inline bool rt_decode_pic(const std::vector<unsigned char>& file, std::vector<unsigned char>& pixels, int& w, int& h, int& ch)
{
    static const unsigned char magic[4] = { 0x53, 0x80, 0xf6, 0x34 };
    if (file.size() < 104 || std::memcmp(file.data(), magic, 4) != 0 || std::memcmp(&file[88], "PICT", 4) != 0) return false;
    rt_be_reader r(file);
    r.skip(92);
    w = r.u16(); h = r.u16();
    r.skip(8);                                            // ratio, fields, pad
    if (w <= 0 || h <= 0 || (uint64_t) w * h > (uint64_t) file.size() * 65536) return false;   // (a mixed run holds up to 65535 pixels)
    struct packet { int type, channel; } packets[10];
    int n_packets = 0, carried = 0, chained;
    do {
        if (n_packets == 10) return false;
        chained = r.u8();
        int size = r.u8();
        packets[n_packets].type = r.u8(); packets[n_packets].channel = r.u8();
        carried |= packets[n_packets].channel;
        n_packets++;
        if (r.at_end() || size != 8) return false;
    } while (chained);
    std::vector<unsigned char> rgba((size_t) w * h * 4, 0xff);
    auto read_value = [&](int channel, unsigned char* dest) {
        for (int i = 0, mask = 0x80; i < 4; i++, mask >>= 1)
            if (channel & mask) { if (r.at_end()) return false; dest[i] = (unsigned char) r.u8(); }
        return true;
    };
    auto copy_value = [](int channel, unsigned char* dest, const unsigned char* src) { for (int i = 0, mask = 0x80; i < 4; i++, mask >>= 1) if (channel & mask) dest[i] = src[i]; };
    for (int y = 0; y < h; y++)
        for (int k = 0; k < n_packets; k++) {
            unsigned char* dest = &rgba[(size_t) y * w * 4];
            const int channel = packets[k].channel;
            if (packets[k].type == 0) { for (int x = 0; x < w; x++, dest += 4) if (!read_value(channel, dest)) return false; }
            else if (packets[k].type == 1) {
                for (int left = w; left > 0;) {
                    int count = r.u8();
                    unsigned char value[4];
                    if (r.at_end()) return false;
                    if (count > left) count = left & 255;
                    if (!read_value(channel, value)) return false;
                    for (int i = 0; i < count; i++, dest += 4) copy_value(channel, dest, value);
                    left -= count;
                    if (count == 0) return false;      // a run of nothing would never end
                }
            }
            else if (packets[k].type == 2) {
                for (int left = w; left > 0;) {
                    int count = r.u8();
                    if (r.at_end()) return false;
                    if (count >= 128) {
                        unsigned char value[4];
                        count = count == 128 ? r.u16() : count - 127;
                        if (count > left || !read_value(channel, value)) return false;
                        for (int i = 0; i < count; i++, dest += 4) copy_value(channel, dest, value);
                    }
                    else {
                        count++;
                        if (count > left) return false;
                        for (int i = 0; i < count; i++, dest += 4) if (!read_value(channel, dest)) return false;
                    }
                    left -= count;
                    if (count == 0) return false;
                }
            }
            else return false;
        }
    ch = (carried & 0x10) ? 4 : 3;
    pixels.resize((size_t) w * h * ch);
    for (size_t i = 0; i < (size_t) w * h; i++) for (int c = 0; c < ch; c++) pixels[i * ch + c] = rgba[i * 4 + c];
    return true;
}

// Radiance RGBE (.hdr) as an 8-bit caller of stb gets it: new-style run-length rows or flat pixels, each RGBE quadruple to float
// (mantissa * 2^(e - 136)), then stb's tone curve for 8-bit callers: (float) pow(v, 1 / 2.2f) * 255 + 0.5f, clamped, truncated.
inline bool rt_decode_hdr(const std::vector<unsigned char>& file, std::vector<unsigned char>& pixels, int& w, int& h, int& ch)
{
    size_t pos = 0;
    auto line = [&]() { std::string s; while (pos < file.size() && file[pos] != '\n') s += (char) file[pos++]; pos++; return s; };
    auto get = [&]() { return pos < file.size() ? (int) file[pos++] : (pos++, 0); };
    std::string first = line();
    if (first != "#?RADIANCE" && first != "#?RGBE") return false;
    bool rgbe_format = false;
    for (;;) {
        if (pos >= file.size()) return false;
        std::string s = line();
        if (s.empty()) break;
        if (s == "FORMAT=32-bit_rle_rgbe") rgbe_format = true;
    }
    if (!rgbe_format) return false;
    std::string size = line();
    if (size.compare(0, 3, "-Y ") != 0) return false;
    char* end = nullptr;
    h = (int) std::strtol(size.c_str() + 3, &end, 10);
    while (*end == ' ') end++;
    if (std::strncmp(end, "+X ", 3) != 0) return false;
    w = (int) std::strtol(end + 3, nullptr, 10);
    if (w <= 0 || h <= 0 || w > (1 << 24) || h > (1 << 24) || (uint64_t) w * h > (1ull << 28) || (uint64_t) w * h > (uint64_t) file.size() * 128) return false;
    ch = 3;
    std::vector<float> rgb((size_t) w * h * 3, 0.f);
    auto convert = [](float* out, const unsigned char* in) {
        if (in[3] != 0) { float f = (float) std::ldexp(1.0f, in[3] - (128 + 8)); out[0] = in[0] * f; out[1] = in[1] * f; out[2] = in[2] * f; }
        else out[0] = out[1] = out[2] = 0.f;
    };
    size_t flat_from = 0;                 // first pixel still to be read as flat data
    bool flat = w < 8 || w >= 32768;
    if (!flat) {
        std::vector<unsigned char> row((size_t) w * 4);
        for (int j = 0; j < h && !flat; j++) {
            int c1 = get(), c2 = get(), len = get();
            if (c1 != 2 || c2 != 2 || (len & 0x80)) {
                // not a run-length row: stb takes these four bytes as the picture's FIRST pixel and reads everything after them flat
                unsigned char px[4] = { (unsigned char) c1, (unsigned char) c2, (unsigned char) len, (unsigned char) get() };
                convert(&rgb[0], px);
                flat = true; flat_from = 1;
                break;
            }
            len = (len << 8) | get();
            if (len != w) return false;
            for (int k = 0; k < 4; k++)
                for (int i = 0; i < w;) {
                    int count = get();
                    if (pos > file.size()) return false;
                    if (count > 128) { int value = get(); count -= 128; if (count > w - i) return false; for (int z = 0; z < count; z++) row[(size_t) i++ * 4 + k] = (unsigned char) value; }
                    else { if (count > w - i || count == 0) return false; for (int z = 0; z < count; z++) row[(size_t) i++ * 4 + k] = (unsigned char) get(); }
                }
            for (int i = 0; i < w; i++) convert(&rgb[((size_t) j * w + i) * 3], &row[(size_t) i * 4]);
        }
    }
    if (flat)
        for (size_t i = flat_from; i < (size_t) w * h; i++) {
            unsigned char px[4];
            for (int k = 0; k < 4; k++) px[k] = (unsigned char) get();
            convert(&rgb[i * 3], px);
        }
    pixels.resize(rgb.size());
    const float gamma_i = 1.0f / 2.2f;
    for (size_t i = 0; i < rgb.size(); i++) {
        float z = (float) std::pow(rgb[i] * 1.0f, gamma_i) * 255 + 0.5f;
        if (z < 0) z = 0;
        if (z > 255) z = 255;
        pixels[i] = (unsigned char) (int) z;
    }
    return true;
}

inline bool rt_load_image(const char* path, std::vector<unsigned char>& pixels, int& w, int& h, int& ch)
{
    std::vector<unsigned char> file;
    if (!rt_read_file(path, file)) return false;
    // stb's order of trying: PNG, BMP, GIF, PSD, PIC, JPEG, PNM, HDR, and TGA last, which has no signature
    return rt_decode_png(file, pixels, w, h, ch) || rt_decode_bmp(file, pixels, w, h, ch) || rt_decode_gif(file, pixels, w, h, ch) ||
           rt_decode_psd(file, pixels, w, h, ch) || rt_decode_pic(file, pixels, w, h, ch) || rt_decode_jpeg(file, pixels, w, h, ch) ||
           rt_decode_ppm(file, pixels, w, h, ch) || rt_decode_hdr(file, pixels, w, h, ch) || rt_decode_tga(file, pixels, w, h, ch);
}

// 8-bit RGB/RGBA/gray PNG.  flip_vertically mirrors stbi_flip_vertically_on_write(true) (raytracing.cpp:124): the
// renderer's row 0 is the bottom of the picture.
inline bool rt_write_png(const char* path, int w, int h, int ch, const unsigned char* data, bool flip_vertically)
{
    if (w <= 0 || h <= 0 || (ch != 1 && ch != 3 && ch != 4)) return false;
    size_t stride = (size_t) w * ch;
    std::vector<unsigned char> raw((stride + 1) * h);
    for (int y = 0; y < h; y++) {
        const unsigned char* row = data + stride * (size_t) (flip_vertically ? h - 1 - y : y);
        raw[(stride + 1) * y] = 0;
        std::memcpy(&raw[(stride + 1) * y + 1], row, stride);
    }
    uLongf zlen = compressBound((uLong) raw.size());
    std::vector<unsigned char> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong) raw.size(), 6) != Z_OK) return false;
    FILE* f = std::fopen(path, "wb");
    if (!f) return false;
    auto chunk = [&](const char* type, const unsigned char* body, uint32_t len) {
        unsigned char hdr[8] = { (unsigned char) (len >> 24), (unsigned char) (len >> 16), (unsigned char) (len >> 8), (unsigned char) len,
                                 (unsigned char) type[0], (unsigned char) type[1], (unsigned char) type[2], (unsigned char) type[3] };
        uLong crc = crc32(0L, hdr + 4, 4);
        if (len) crc = crc32(crc, body, len);
        unsigned char tail[4] = { (unsigned char) (crc >> 24), (unsigned char) (crc >> 16), (unsigned char) (crc >> 8), (unsigned char) crc };
        std::fwrite(hdr, 1, 8, f);
        if (len) std::fwrite(body, 1, len, f);
        std::fwrite(tail, 1, 4, f);
    };
    static const unsigned char sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    std::fwrite(sig, 1, 8, f);
    unsigned char ihdr[13] = { (unsigned char) (w >> 24), (unsigned char) (w >> 16), (unsigned char) (w >> 8), (unsigned char) w,
                               (unsigned char) (h >> 24), (unsigned char) (h >> 16), (unsigned char) (h >> 8), (unsigned char) h,
                               8, (unsigned char) (ch == 1 ? 0 : ch == 3 ? 2 : 6), 0, 0, 0 };
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", z.data(), (uint32_t) zlen);
    chunk("IEND", nullptr, 0);
    return std::fclose(f) == 0;
}
