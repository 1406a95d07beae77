// raytracing_amd/host/loader.hpp -- OBJ/MTL loader with the reference loader's interface and semantics
// (src/cpp/loader.hpp:17-172): one `triangle` per face, float attributes promoted to double, one `phong` per MTL
// entry (Kd/Ks/Ke/Tr/Ns/Ni, map_Kd -> image(path, color_based=true)), emissive faces also pushed to lights().
//
// The reference parses with the vendored tinyobjloader 2.0.0, which this repository does not carry; the small parser
// below reproduces what the loader consumes from it: `v/vt/vn/f/usemtl/mtllib`, negative (relative) indices,
// float32 storage, quads split along the shorter diagonal (ties -> [0,1,3],[1,2,3]), larger polygons ear-clipped by tinyobj's
// built-in method (ear_clip below; pinned by tests/golden/ngon.npz, a fixture of the real loader), material defaults of tinyobj
// (all colours 0, Ns 1, Ni 1).
// Also reproduced: the missing-normal fallback computes the face normal from the scratch vertices as they are at
// that moment -- corners 1 and 2 still hold the PREVIOUS face while corner 0 is processed (loader.hpp:104-110).
#pragma once

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <vector>

#include "group.hpp"
#include "material.hpp"
#include "polygon.hpp"
#include "texture.hpp"

class loader {
public:
    loader(const char* path) : scene_path(path)
    {
        std::ifstream in(path);
        if (!in) { std::printf("[OBJLoader] error: cannot open %s\n", path); return; }
        parse(in);
    }
    group meshes() { return grp; }
    group* lights() { return lgt; }
    int skipped_faces() const { return n_bad_faces; }      // faces dropped because an index pointed outside the vertex arrays

private:
    int n_bad_faces = 0;
    struct corner { int v, vt, vn; };
    group grp;
    group* lgt = new group();
    std::string scene_path;
    std::vector<material*> mats;
    std::map<std::string, int> mat_index;
    std::set<std::string> loaded_mtl;
    std::vector<float> pos, nrm, tex;

    static std::string dirname(const std::string& p)
    {
        std::string::size_type s = p.find_last_of("/\\");
        if (s == std::string::npos) return ".";
        if (s == 0) return "/";
        return p.substr(0, s);
    }
    static float to_float(const std::string& s) { return (float) std::strtod(s.c_str(), nullptr); }

    // `map_Kd [options] file name`, the way tinyobjloader reads it (ParseTextureNameAndOption, tiny_obj_loader.hpp:1247-1331): options
    // in front of the name are skipped with the number of arguments tinyobj consumes for each (three for -o / -s / -t, two for -mm, one
    // for the others, whatever the tokens are), and the name is the REST of the line, blanks included.
    static std::string texture_name(std::istringstream& ss)
    {
        std::string tok;
        while (ss >> tok) {
            int skip = -1;
            if (tok == "-o" || tok == "-s" || tok == "-t") skip = 3;
            else if (tok == "-mm") skip = 2;
            else if (tok == "-blendu" || tok == "-blendv" || tok == "-clamp" || tok == "-boost" || tok == "-bm" || tok == "-type" || tok == "-texres" ||
                     tok == "-imfchan" || tok == "-colorspace") skip = 1;
            if (skip < 0) {
                std::string rest;
                std::getline(ss, rest);
                tok += rest;
                while (!tok.empty() && (tok.back() == '\r' || tok.back() == ' ' || tok.back() == '\t')) tok.pop_back();
                return tok;
            }
            std::string arg;
            for (int k = 0; k < skip; k++) ss >> arg;
        }
        return std::string();
    }

    // one MTL file; false when it cannot be opened
    bool load_mtl(const std::string& file)
    {
        std::ifstream in(dirname(scene_path) + "/" + file);
        if (!in) return false;
        phong* cur = nullptr;
        std::string line;
        while (std::getline(in, line)) {
            std::istringstream ss(line);
            std::string key;
            if (!(ss >> key) || key[0] == '#') continue;
            if (key == "newmtl") {
                std::string name; ss >> name;
                cur = new phong((texture*) nullptr);
                cur->Kd = vec3d(0, 0, 0); cur->Ns = 1.0f; cur->Ni = 1.0f;      // tinyobj InitMaterial defaults
                mat_index[name] = (int) mats.size();
                mats.push_back(cur);
                continue;
            }
            if (!cur) continue;
            auto rgb = [&]() { std::string a, b, c; ss >> a >> b >> c; return vec3d(to_float(a), to_float(b), to_float(c)); };
            if (key == "Kd") cur->Kd = rgb();
            else if (key == "Ks") cur->Ks = rgb();
            else if (key == "Ke") cur->Ke = rgb();
            else if (key == "Kt" || key == "Tf") cur->Tr = rgb();
            else if (key == "Ns") { std::string a; ss >> a; cur->Ns = to_float(a); }
            else if (key == "Ni") { std::string a; ss >> a; cur->Ni = to_float(a); }
            else if (key == "map_Kd") {
                std::string name = texture_name(ss);
                if (name.empty()) continue;
                std::string image_path = dirname(scene_path) + "/" + name;
                std::printf("[OBJLoader] load diffuse image: %s\n", image_path.c_str());
                cur->diffuse_map = new image(image_path.c_str(), true);
            }
        }
        return true;
    }

    // `mtllib a.mtl b.mtl`: names separated by blanks (a backslash keeps the next character), tried in turn until one loads; a file
    // that has been loaded before is not read again (tiny_obj_loader.hpp:2843-2893)
    void load_mtllib(const std::string& rest)
    {
        std::vector<std::string> names(1);
        for (size_t i = 0; i < rest.size(); i++) {
            char c = rest[i];
            if (c == '\\' && i + 1 < rest.size()) names.back() += rest[++i];
            else if (c == ' ') { if (!names.back().empty()) names.emplace_back(); }
            else if (c != '\r' && c != '\t') names.back() += c;
        }
        if (names.back().empty()) names.pop_back();
        bool found = false;
        for (const std::string& n : names) {
            if (loaded_mtl.count(n)) { found = true; continue; }
            if (load_mtl(n)) { loaded_mtl.insert(n); found = true; break; }
        }
        if (!found) std::printf("[OBJLoader] warning: failed to load material file(s): %s\n", rest.c_str());
    }

    // Polygons with more than four corners, the way tinyobjloader 2.0.0's built-in triangulation cuts them (tiny_obj_loader.hpp:1715-1936;
    // the reference does not build with the mapbox variant): project onto the two axes picked from the first non-degenerate corner,
    // then clip ears in float32 -- a corner is skipped while it is reflex ("cross * area < 0", with tinyobj's own `area` expression)
    // or while another remaining corner lies inside its triangle (pnpoly), and the walk gives up after a full round without progress.
    // Restated here because which triangles come out, and in which order, decides primitive ids, the BVH and every image.
    static int pnpoly3(const float* vx, const float* vy, float tx, float ty)
    {
        int c = 0;
        for (int i = 0, j = 2; i < 3; j = i++)
            if (((vy[i] > ty) != (vy[j] > ty)) && (tx < (vx[j] - vx[i]) * (ty - vy[i]) / (vy[j] - vy[i]) + vx[i])) c = !c;
        return c;
    }
    void ear_clip(const std::vector<corner>& face, std::vector<corner>& tris) const
    {
        size_t n = face.size();
        size_t axes[2] = { 1, 2 };
        for (size_t k = 0; k < n; k++) {
            const float* p0 = &pos[(size_t) face[k % n].v * 3];
            const float* p1 = &pos[(size_t) face[(k + 1) % n].v * 3];
            const float* p2 = &pos[(size_t) face[(k + 2) % n].v * 3];
            const float e0x = p1[0] - p0[0], e0y = p1[1] - p0[1], e0z = p1[2] - p0[2];
            const float e1x = p2[0] - p1[0], e1y = p2[1] - p1[1], e1z = p2[2] - p1[2];
            const float cx = std::fabs(e0y * e1z - e0z * e1y), cy = std::fabs(e0z * e1x - e0x * e1z), cz = std::fabs(e0x * e1y - e0y * e1x);
            const float eps = 1.1920929e-7f;       // std::numeric_limits<float>::epsilon()
            if (cx > eps || cy > eps || cz > eps) {
                if (!(cx > cy && cx > cz)) { axes[0] = 0; if (cz > cx && cz > cy) axes[1] = 1; }
                break;
            }
        }
        std::vector<corner> rest = face;
        size_t guess = 0, budget = n, before = n;
        while (rest.size() > 3 && budget > 0) {
            n = rest.size();
            if (guess >= n) guess -= n;
            if (before != n) { before = n; budget = n; }
            else budget--;
            corner ind[3];
            float vx[3], vy[3];
            for (size_t k = 0; k < 3; k++) {
                ind[k] = rest[(guess + k) % n];
                vx[k] = pos[(size_t) ind[k].v * 3 + axes[0]];
                vy[k] = pos[(size_t) ind[k].v * 3 + axes[1]];
            }
            const float e0x = vx[1] - vx[0], e0y = vy[1] - vy[0], e1x = vx[2] - vx[1], e1y = vy[2] - vy[1];
            const float cross = e0x * e1y - e0y * e1x;
            const float area = (vx[0] * vy[1] - vy[0] * vx[1]) * 0.5f;
            if (cross * area < 0.0f) { guess += 1; continue; }
            bool overlap = false;
            for (size_t other = 3; other < n && !overlap; other++) {
                const corner& o = rest[(guess + other) % n];
                overlap = pnpoly3(vx, vy, pos[(size_t) o.v * 3 + axes[0]], pos[(size_t) o.v * 3 + axes[1]]) != 0;
            }
            if (overlap) { guess += 1; continue; }
            tris.push_back(ind[0]); tris.push_back(ind[1]); tris.push_back(ind[2]);
            rest.erase(rest.begin() + (long) ((guess + 1) % n));
        }
        if (rest.size() == 3) { tris.push_back(rest[0]); tris.push_back(rest[1]); tris.push_back(rest[2]); }
    }

    static int fix_index(int idx, size_t count) { return idx > 0 ? idx - 1 : (idx < 0 ? (int) count + idx : -1); }

    void parse(std::istream& in)
    {
        vertex scratch[3];
        int cur_mat = -1;
        std::string line;
        while (std::getline(in, line)) {
            std::istringstream ss(line);
            std::string key;
            if (!(ss >> key) || key[0] == '#') continue;
            if (key == "v") { std::string a, b, c; ss >> a >> b >> c; pos.push_back(to_float(a)); pos.push_back(to_float(b)); pos.push_back(to_float(c)); }
            else if (key == "vn") { std::string a, b, c; ss >> a >> b >> c; nrm.push_back(to_float(a)); nrm.push_back(to_float(b)); nrm.push_back(to_float(c)); }
            else if (key == "vt") { std::string a, b; ss >> a >> b; tex.push_back(to_float(a)); tex.push_back(to_float(b)); }
            else if (key == "mtllib") { std::string rest; std::getline(ss, rest); size_t b = rest.find_first_not_of(" \t"); load_mtllib(b == std::string::npos ? std::string() : rest.substr(b)); }
            else if (key == "usemtl") { std::string n; ss >> n; auto it = mat_index.find(n); cur_mat = it == mat_index.end() ? -1 : it->second; }
            else if (key == "f") {
                std::vector<corner> face;
                std::string tok;
                bool bad_index = false;
                while (ss >> tok) {
                    corner c = { 0, 0, 0 };
                    int field = 0;
                    size_t start = 0;
                    for (size_t i = 0; i <= tok.size(); i++) {
                        if (i == tok.size() || tok[i] == '/') {
                            int val = i > start ? std::atoi(tok.substr(start, i - start).c_str()) : 0;
                            if (field == 0) c.v = val; else if (field == 1) c.vt = val; else if (field == 2) c.vn = val;
                            field++; start = i + 1;
                        }
                    }
                    const bool has_vt = c.vt != 0, has_vn = c.vn != 0;
                    c.v = fix_index(c.v, pos.size() / 3); c.vt = fix_index(c.vt, tex.size() / 2); c.vn = fix_index(c.vn, nrm.size() / 3);
                    // tinyobj fails the whole parse on an index it cannot resolve ("Failed parse `f' line"); here the face is dropped, so that a
                    // malformed or truncated file can never index outside pos / tex / nrm
                    if (c.v < 0 || (size_t) c.v >= pos.size() / 3 || (has_vt && (c.vt < 0 || (size_t) c.vt >= tex.size() / 2)) ||
                        (has_vn && (c.vn < 0 || (size_t) c.vn >= nrm.size() / 3))) bad_index = true;
                    face.push_back(c);
                }
                if (bad_index) { n_bad_faces++; std::printf("[OBJLoader] face with an index outside the vertex arrays skipped: %s\n", line.c_str()); continue; }
                if (face.size() < 3) continue;
                std::vector<corner> tris;
                if (face.size() == 4) {
                    const float* q[4];
                    for (int k = 0; k < 4; k++) q[k] = &pos[(size_t) face[k].v * 3];
                    float e02x = q[2][0] - q[0][0], e02y = q[2][1] - q[0][1], e02z = q[2][2] - q[0][2];
                    float e13x = q[3][0] - q[1][0], e13y = q[3][1] - q[1][1], e13z = q[3][2] - q[1][2];
                    float s02 = e02x * e02x + e02y * e02y + e02z * e02z, s13 = e13x * e13x + e13y * e13y + e13z * e13z;
                    if (s02 < s13) tris = { face[0], face[1], face[2], face[0], face[2], face[3] };
                    else tris = { face[0], face[1], face[3], face[1], face[2], face[3] };
                }
                else if (face.size() > 4) ear_clip(face, tris);
                else tris = face;
                for (size_t t = 0; t + 2 < tris.size(); t += 3) {
                    for (int c = 0; c < 3; c++) {
                        const corner& idx = tris[t + c];
                        scratch[c].position = vec3d(pos[(size_t) idx.v * 3], pos[(size_t) idx.v * 3 + 1], pos[(size_t) idx.v * 3 + 2]);
                        if (idx.vn >= 0) scratch[c].normal = vec3d(nrm[(size_t) idx.vn * 3], nrm[(size_t) idx.vn * 3 + 1], nrm[(size_t) idx.vn * 3 + 2]);
                        else {
                            vec3d e1 = scratch[1].position - scratch[0].position, e2 = scratch[2].position - scratch[0].position;
                            scratch[c].normal = e1.cross(e2).normalize();
                        }
                        scratch[c].uv_coord = idx.vt >= 0 ? vec2d(tex[(size_t) idx.vt * 2], tex[(size_t) idx.vt * 2 + 1]) : vec2d(0, 0);
                    }
                    if (cur_mat < 0) { std::printf("[OBJLoader] face without material skipped (the reference indexes mats[-1] here)\n"); continue; }
                    triangle* tri = new triangle(scratch, mats[cur_mat]);
                    grp.add(tri);
                    if (mats[cur_mat]->has_emission()) lgt->add(tri);
                }
            }
        }
    }
};
