// obj_loader.h -- Wavefront OBJ/MTL reader with the semantics the reference gets from the `tobj` crate (3.2.5,
// third-party) as called in crates/rt/src/loader/obj.rs:31-34: GPU_LOAD_OPTIONS + single_index, i.e. faces
// fan-triangulated, one index stream, points and lines ignored, one model per o/g block and per material change.
#pragma once

#include <stdint.h>

#include <string>
#include <vector>

namespace rto {

struct Model {
    std::string name;
    std::vector<float> positions;  // 3 per vertex, in order of first use, only the vertices the model references
    std::vector<uint32_t> indices; // 3 per triangle
    int material_id = -1;          // index into materials, -1 = none
};

struct MtlMaterial {
    std::string name;
    float diffuse[3] = {0.0f, 0.0f, 0.0f};
};

struct ObjFile {
    std::vector<Model> models;
    std::vector<MtlMaterial> materials;
    bool materials_ok = false; // false when no MTL could be read: every model then uses the caller's default
};

// Returns false (and sets err) when the OBJ itself cannot be read or holds a malformed face.
bool load_obj(const std::string &path, ObjFile &out, std::string &err);

} // namespace rto
