// obj_loader.cpp -- see obj_loader.h.
#include "obj_loader.h"

#include <stdlib.h>
#include <string.h>

#include <fstream>
#include <unordered_map>

namespace rto {

namespace {

std::string trim(const std::string &s) {
    size_t a = 0, b = s.size();
    while (a < b && (s[a] == ' ' || s[a] == '\t' || s[a] == '\r' || s[a] == '\n')) ++a;
    while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t' || s[b - 1] == '\r' || s[b - 1] == '\n')) --b;
    return s.substr(a, b - a);
}

// first whitespace-delimited token and the rest of the line
void split_key(const std::string &line, std::string &key, std::string &rest) {
    size_t i = 0;
    while (i < line.size() && line[i] != ' ' && line[i] != '\t') ++i;
    key = line.substr(0, i);
    rest = trim(line.substr(i));
}

bool parse_mtl(const std::string &path, std::vector<MtlMaterial> &mats) {
    std::ifstream in(path);
    if (!in) return false;
    std::string raw;
    while (std::getline(in, raw)) {
        std::string line = trim(raw);
        if (line.empty() || line[0] == '#') continue;
        std::string key, rest;
        split_key(line, key, rest);
        if (key == "newmtl") {
            MtlMaterial m;
            m.name = rest;
            mats.push_back(m);
        } else if (key == "Kd" && !mats.empty()) {
            const char *p = rest.c_str();
            char *end = nullptr;
            for (int k = 0; k < 3; ++k) {
                mats.back().diffuse[k] = strtof(p, &end);
                if (end == p) break;
                p = end;
            }
        }
    }
    return true;
}

struct PendingModel {
    std::vector<int64_t> corners; // 3 global vertex indices per triangle
};

} // namespace

bool load_obj(const std::string &path, ObjFile &out, std::string &err) {
    out = ObjFile();
    std::ifstream in(path);
    if (!in) {
        err = "cannot open " + path;
        return false;
    }
    std::string dir;
    {
        size_t slash = path.find_last_of('/');
        dir = slash == std::string::npos ? std::string() : path.substr(0, slash + 1);
    }
    std::vector<float> verts;
    std::unordered_map<std::string, int> mat_map;
    std::string name = "unnamed_object";
    int mat_id = -1;
    PendingModel cur;

    auto flush = [&]() {
        Model m;
        m.name = name;
        m.material_id = mat_id;
        std::unordered_map<int64_t, uint32_t> remap;
        remap.reserve(cur.corners.size());
        m.indices.reserve(cur.corners.size());
        for (int64_t v : cur.corners) {
            auto it = remap.find(v);
            uint32_t j;
            if (it == remap.end()) {
                j = (uint32_t)(m.positions.size() / 3);
                remap.emplace(v, j);
                m.positions.push_back(verts[3 * (size_t)v]);
                m.positions.push_back(verts[3 * (size_t)v + 1]);
                m.positions.push_back(verts[3 * (size_t)v + 2]);
            } else {
                j = it->second;
            }
            m.indices.push_back(j);
        }
        out.models.push_back(std::move(m));
        cur.corners.clear();
    };

    std::string raw;
    size_t line_no = 0;
    while (std::getline(in, raw)) {
        ++line_no;
        std::string line = trim(raw);
        if (line.empty() || line[0] == '#') continue;
        std::string key, rest;
        split_key(line, key, rest);
        if (key == "v") {
            const char *p = rest.c_str();
            char *end = nullptr;
            float xyz[3] = {0.0f, 0.0f, 0.0f};
            for (int k = 0; k < 3; ++k) {
                xyz[k] = strtof(p, &end);
                if (end == p) {
                    err = path + ":" + std::to_string(line_no) + ": malformed vertex";
                    return false;
                }
                p = end;
            }
            verts.insert(verts.end(), xyz, xyz + 3);
        } else if (key == "f") {
            const int64_t nv = (int64_t)(verts.size() / 3);
            std::vector<int64_t> corners;
            const char *p = rest.c_str();
            while (*p) {
                while (*p == ' ' || *p == '\t') ++p;
                if (!*p) break;
                char *end = nullptr;
                long long idx = strtoll(p, &end, 10);
                if (end == p) {
                    err = path + ":" + std::to_string(line_no) + ": malformed face";
                    return false;
                }
                int64_t v = idx < 0 ? nv + idx : idx - 1;
                if (v < 0 || v >= nv) {
                    err = path + ":" + std::to_string(line_no) + ": face index out of range";
                    return false;
                }
                corners.push_back(v);
                p = end;
                while (*p && *p != ' ' && *p != '\t') ++p; // skip /vt/vn
            }
            if (corners.size() < 3) {
                err = path + ":" + std::to_string(line_no) + ": face with fewer than 3 vertices";
                return false;
            }
            int64_t a = corners[0], b = corners[1];
            for (size_t k = 2; k < corners.size(); ++k) {
                cur.corners.push_back(a);
                cur.corners.push_back(b);
                cur.corners.push_back(corners[k]);
                b = corners[k];
            }
        } else if (key == "o" || key == "g") {
            if (!cur.corners.empty()) flush();
            name = rest.empty() ? "unnamed_object" : rest;
        } else if (key == "mtllib") {
            std::vector<MtlMaterial> mats;
            if (parse_mtl(dir + rest, mats)) {
                out.materials_ok = true;
                for (auto &m : mats) {
                    mat_map[m.name] = (int)out.materials.size();
                    out.materials.push_back(m);
                }
            }
        } else if (key == "usemtl") {
            if (!rest.empty()) {
                auto it = mat_map.find(rest);
                const int new_mat = it == mat_map.end() ? -1 : it->second;
                if (new_mat != mat_id && !cur.corners.empty()) flush();
                mat_id = new_mat;
            }
        }
    }
    flush();
    return true;
}

} // namespace rto
