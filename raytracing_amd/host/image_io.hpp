// raytracing_amd/host/image_io.hpp -- image read/write for the host driver.
// The reference uses the vendored stb_image / stb_image_write (texture.hpp:87, raytracing.cpp:124-125); those are
// third-party sources this repository does not carry, so the two calls the renderer needs are provided here: PNG decode on
// top of zlib (every colour type and bit depth, Adam7 included), JPEG decode (jpeg_decode.hpp: baseline and progressive), binary PPM
// decode, and PNG encode.  Decoded texels use stb's layout: rows top to bottom, channels interleaved, as many channels as the
// file holds (the `0` of stbi_load(path, &w, &h, &ch, 0)).
#pragma once

#include <zlib.h>

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

inline bool rt_decode_ppm(const std::vector<unsigned char>& file, std::vector<unsigned char>& pixels, int& w, int& h, int& ch)
{
    if (file.size() < 11 || file[0] != 'P' || file[1] != '6') return false;
    size_t pos = 2;
    int vals[3], got = 0;
    while (got < 3 && pos < file.size()) {
        while (pos < file.size() && (file[pos] == ' ' || file[pos] == '\n' || file[pos] == '\r' || file[pos] == '\t')) pos++;
        if (pos < file.size() && file[pos] == '#') { while (pos < file.size() && file[pos] != '\n') pos++; continue; }
        int v = 0; bool any = false;
        while (pos < file.size() && file[pos] >= '0' && file[pos] <= '9') { if (v < (1 << 24)) v = v * 10 + (file[pos] - '0'); pos++; any = true; }
        if (!any) return false;
        vals[got++] = v;
    }
    pos++;
    w = vals[0]; h = vals[1]; ch = 3;
    if (got < 3 || vals[2] != 255 || w <= 0 || h <= 0 || (uint64_t) w * h > (1ull << 28) || pos + (size_t) w * h * 3 > file.size()) return false;
    pixels.assign(file.begin() + pos, file.begin() + pos + (size_t) w * h * 3);
    return true;
}

inline bool rt_load_image(const char* path, std::vector<unsigned char>& pixels, int& w, int& h, int& ch)
{
    std::vector<unsigned char> file;
    if (!rt_read_file(path, file)) return false;
    return rt_decode_png(file, pixels, w, h, ch) || rt_decode_jpeg(file, pixels, w, h, ch) || rt_decode_ppm(file, pixels, w, h, ch);   // stb tries them in this order
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
