// raytracing_amd/host/image_io.hpp -- minimal PNG read/write for the host driver.
// The reference uses the vendored stb_image / stb_image_write (texture.hpp:87, raytracing.cpp:124-125); those are
// third-party sources this repository does not carry, so the two calls the renderer needs are provided here on
// top of zlib: 8-bit non-interlaced PNG decode (gray / gray+alpha / RGB / RGBA / palette), binary PPM decode, and
// PNG encode.  Decoded texels use stb's layout: rows top to bottom, channels interleaved.
#pragma once

#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

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

inline bool rt_decode_png(const std::vector<unsigned char>& file, std::vector<unsigned char>& pixels, int& w, int& h, int& ch)
{
    static const unsigned char sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    if (file.size() < 33 || std::memcmp(file.data(), sig, 8) != 0) return false;
    size_t pos = 8;
    int depth = 0, color = 0, interlace = 0;
    std::vector<unsigned char> idat, palette;
    while (pos + 12 <= file.size()) {
        uint32_t len = rt_be32(&file[pos]);
        const unsigned char* type = &file[pos + 4];
        const unsigned char* body = &file[pos + 8];
        if (pos + 12 + (size_t) len > file.size()) return false;
        if (!std::memcmp(type, "IHDR", 4)) { w = (int) rt_be32(body); h = (int) rt_be32(body + 4); depth = body[8]; color = body[9]; interlace = body[12]; }
        else if (!std::memcmp(type, "PLTE", 4)) palette.assign(body, body + len);
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!std::memcmp(type, "IEND", 4)) break;
        pos += 12 + (size_t) len;
    }
    if (depth != 8 || interlace != 0 || w <= 0 || h <= 0) return false;
    int src_ch = color == 0 ? 1 : color == 2 ? 3 : color == 3 ? 1 : color == 4 ? 2 : color == 6 ? 4 : 0;
    if (!src_ch) return false;
    size_t stride = (size_t) w * src_ch;
    std::vector<unsigned char> raw((stride + 1) * h);
    uLongf raw_len = (uLongf) raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong) idat.size()) != Z_OK || raw_len != raw.size()) return false;
    std::vector<unsigned char> img(stride * h);
    for (int y = 0; y < h; y++) {
        const unsigned char* in = &raw[(stride + 1) * y];
        unsigned char* cur = &img[stride * y];
        const unsigned char* up = y ? &img[stride * (y - 1)] : nullptr;
        int filter = in[0];
        for (size_t x = 0; x < stride; x++) {
            int a = x >= (size_t) src_ch ? cur[x - src_ch] : 0, b = up ? up[x] : 0, c = (up && x >= (size_t) src_ch) ? up[x - src_ch] : 0;
            int v = in[1 + x];
            switch (filter) {
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) >> 1; break;
            case 4: { int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
            default: break;
            }
            cur[x] = (unsigned char) v;
        }
    }
    if (color == 3) {   // palette -> RGB
        ch = 3;
        pixels.resize((size_t) w * h * 3);
        for (size_t i = 0; i < (size_t) w * h; i++) {
            size_t k = (size_t) img[i] * 3;
            for (int c = 0; c < 3; c++) pixels[i * 3 + c] = k + c < palette.size() ? palette[k + c] : 0;
        }
    }
    else { ch = src_ch; pixels.swap(img); }
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
        while (pos < file.size() && file[pos] >= '0' && file[pos] <= '9') { v = v * 10 + (file[pos] - '0'); pos++; any = true; }
        if (!any) return false;
        vals[got++] = v;
    }
    pos++;
    w = vals[0]; h = vals[1]; ch = 3;
    if (got < 3 || vals[2] != 255 || pos + (size_t) w * h * 3 > file.size()) return false;
    pixels.assign(file.begin() + pos, file.begin() + pos + (size_t) w * h * 3);
    return true;
}

inline bool rt_load_image(const char* path, std::vector<unsigned char>& pixels, int& w, int& h, int& ch)
{
    std::vector<unsigned char> file;
    if (!rt_read_file(path, file)) return false;
    return rt_decode_png(file, pixels, w, h, ch) || rt_decode_ppm(file, pixels, w, h, ch);
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
