// raytracing_amd/host/texture.hpp -- texture classes (reference: src/cpp/texture.hpp:13-133): constant, checker, image.
// Host objects only carry parameters; colour evaluation happens on the GPU (csrc/rt_device.h tex_color).
#pragma once

#include <cstdio>
#include <string>
#include <vector>

#include "algebra.hpp"
#include "flatten.hpp"
#include "image_io.hpp"

class texture {
public:
    uint64_t serial = rt_next_serial();
    virtual ~texture() {}
    virtual void flatten(flattener& f, int id) const = 0;   // fills out.tex_kind[id] / tex_data
};

class constant : public texture {
public:
    vec3d base_color;
    constant() {}
    constant(vec3d c) : base_color(c) {}
    constant(double r, double g, double b) : base_color(r, g, b) {}
    void flatten(flattener& f, int id) const override
    {
        f.out.tex_kind[id] = RT_TEX_CONSTANT;
        for (int k = 0; k < 3; k++) f.out.tex_data[(size_t) id * RT_TEX_STRIDE + k] = base_color[k];
    }
};

class checker : public texture {
public:
    texture* even;
    texture* odds;
    checker() : even(new constant(0, 0, 0)), odds(new constant(1, 1, 1)) {}
    checker(texture* e, texture* o) : even(e), odds(o) {}
    checker(const vec3d& even_color, const vec3d odds_color) : even(new constant(even_color)), odds(new constant(odds_color)) {}
    void flatten(flattener& f, int id) const override
    {
        int e = f.texture_id(even), o = f.texture_id(odds);
        f.out.tex_kind[id] = RT_TEX_CHECKER;
        f.out.tex_data[(size_t) id * RT_TEX_STRIDE] = e;
        f.out.tex_data[(size_t) id * RT_TEX_STRIDE + 1] = o;
    }
};

class image : public texture {
public:
    image() {}
    image(const char* file, bool color_based_ = false) : path(file), color_based(color_based_)
    {
        if (!rt_load_image(file, pixels, width, height, channels)) std::fprintf(stderr, "[ERROR] can not load image: %s\n", file);
    }
    // in-memory texels (stb layout: rows top to bottom, interleaved channels)
    image(const unsigned char* data, int w, int h, int ch, bool color_based_ = false)
        : pixels(data, data + (size_t) w * h * ch), width(w), height(h), channels(ch), color_based(color_based_) {}
    void flatten(flattener& f, int id) const override
    {
        double* d = &f.out.tex_data[(size_t) id * RT_TEX_STRIDE];
        f.out.tex_kind[id] = RT_TEX_IMAGE;
        d[0] = pixels.empty() ? 0 : width; d[1] = height; d[2] = channels; d[3] = color_based; d[4] = (double) f.out.tex_pixels.size();
        f.out.tex_pixels.insert(f.out.tex_pixels.end(), pixels.begin(), pixels.end());
    }
private:
    std::string path;
    std::vector<unsigned char> pixels;
    int width = 0, height = 0, channels = 0;
    bool color_based = false;
};

inline int flattener::texture_id(const texture* t)
{
    auto it = tex_ids.find(t);
    if (it != tex_ids.end()) return it->second;
    int id = (int) out.tex_kind.size();
    tex_ids[t] = id;
    out.tex_kind.push_back(0);
    out.tex_data.resize(out.tex_data.size() + RT_TEX_STRIDE, 0.0);
    t->flatten(*this, id);
    return id;
}
