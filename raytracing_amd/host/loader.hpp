// raytracing_amd/host/loader.hpp -- OBJ/MTL loader with the reference loader's interface and semantics
// (src/cpp/loader.hpp:17-172): one `triangle` per face, float attributes promoted to double, one `phong` per MTL
// entry (Kd/Ks/Ke/Tr/Ns/Ni, map_Kd -> image(path, color_based=true)), emissive faces also pushed to lights().
//
// The reference parses with the vendored tinyobjloader 2.0.0, which this repository does not carry; the small parser
// below reproduces what the loader consumes from it: `v/vt/vn/f/usemtl/mtllib`, negative (relative) indices,
// float32 storage, quads split along the shorter diagonal (ties -> [0,1,3],[1,2,3]), larger polygons as a fan
// (tinyobj ear-clips those: documented deviation), material defaults of tinyobj (all colours 0, Ns 1, Ni 1).
// Also reproduced: the missing-normal fallback computes the face normal from the scratch vertices as they are at
// that moment -- corners 1 and 2 still hold the PREVIOUS face while corner 0 is processed (loader.hpp:104-110).
#pragma once

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
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
    std::vector<float> pos, nrm, tex;

    static std::string dirname(const std::string& p)
    {
        std::string::size_type s = p.find_last_of("/\\");
        if (s == std::string::npos) return ".";
        if (s == 0) return "/";
        return p.substr(0, s);
    }
    static float to_float(const std::string& s) { return (float) std::strtod(s.c_str(), nullptr); }

    void load_mtl(const std::string& file)
    {
        std::ifstream in(dirname(scene_path) + "/" + file);
        if (!in) { std::printf("[OBJLoader] warning: material file %s not found\n", file.c_str()); return; }
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
                std::string name; ss >> name;
                std::string image_path = dirname(scene_path) + "/" + name;
                std::printf("[OBJLoader] load diffuse image: %s\n", image_path.c_str());
                cur->diffuse_map = new image(image_path.c_str(), true);
            }
        }
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
            else if (key == "mtllib") { std::string f; ss >> f; load_mtl(f); }
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
                else {
                    for (size_t k = 1; k + 1 < face.size(); k++) { tris.push_back(face[0]); tris.push_back(face[k]); tris.push_back(face[k + 1]); }
                }
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
