// OBJ(.gz) reader standing in for tobj 4.0.0 `load_obj_buf` with default LoadOptions as the reference
// calls it (load.rs:207-215): no triangulation, multi-index export, positions parsed as f32.
// Only what load_mesh consumes is produced: the FIRST model's positions / normals / indices.
#include <zlib.h>

#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <unordered_map>

#include "scene_host.h"

namespace rt {

namespace {

std::string read_all(const std::string& path) {
    // gzopen reads plain files transparently; the reference gunzips iff the extension is "gz"
    // (load.rs:318) — same outcome for well-formed inputs.
    gzFile f = gzopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("could not open '" + path + "': " + std::strerror(errno));
    gzbuffer(f, 1 << 20);
    std::string data;
    char buf[1 << 16];
    int n;
    while ((n = gzread(f, buf, sizeof buf)) > 0) data.append(buf, (size_t)n);
    int errnum = 0;
    const char* msg = gzerror(f, &errnum);
    if (n < 0 || (errnum != Z_OK && errnum != Z_STREAM_END)) {
        std::string m = msg ? msg : "read error";
        gzclose(f);
        throw std::runtime_error("could not read '" + path + "': " + m);
    }
    gzclose(f);
    return data;
}

struct FaceVert {
    long v = -1, vn = -1;
};

// "v", "v/vt", "v//vn", "v/vt/vn"; 1-based, negative = relative to the counts so far.
bool parse_face_vert(const char* tok, size_t len, size_t n_pos, size_t n_nrm, FaceVert& out) {
    long idx[3] = {0, 0, 0};
    bool have[3] = {false, false, false};
    int field = 0;
    size_t i = 0;
    while (i <= len && field < 3) {
        size_t b = i;
        while (i < len && tok[i] != '/') ++i;
        if (i > b) {
            char tmp[32];
            size_t l = i - b < 31 ? i - b : 31;
            std::memcpy(tmp, tok + b, l);
            tmp[l] = 0;
            char* end = nullptr;
            idx[field] = std::strtol(tmp, &end, 10);
            if (*end) return false;
            have[field] = true;
        }
        ++field;
        ++i;
    }
    if (!have[0]) return false;
    out.v = idx[0] < 0 ? (long)n_pos + idx[0] : idx[0] - 1;
    out.vn = !have[2] ? -1 : (idx[2] < 0 ? (long)n_nrm + idx[2] : idx[2] - 1);
    return out.v >= 0 && (size_t)out.v < n_pos && (out.vn < 0 || (size_t)out.vn < n_nrm);
}

}  // namespace

void load_obj_first_model(const std::string& path, ObjMesh& out) {
    std::string data = read_all(path);
    std::vector<float> pos, nrm;
    std::vector<FaceVert> face_verts;  // 3 per face of the first model
    std::string name = "unnamed_object";
    bool done = false;

    size_t p = 0;
    size_t lineno = 0;
    while (p < data.size() && !done) {
        size_t e = data.find('\n', p);
        if (e == std::string::npos) e = data.size();
        ++lineno;
        size_t b = p;
        p = e + 1;
        while (b < e && (data[b] == ' ' || data[b] == '\t')) ++b;
        size_t le = e;
        while (le > b && (data[le - 1] == '\r' || data[le - 1] == ' ' || data[le - 1] == '\t')) --le;
        if (b >= le || data[b] == '#') continue;
        size_t k = b;
        while (k < le && data[k] != ' ' && data[k] != '\t') ++k;
        std::string key = data.substr(b, k - b);
        const char* rest = data.c_str() + k;  // NUL-terminated buffer: strtof stops at line end anyway
        if (key == "v" || key == "vn") {
            char* end = nullptr;
            float xyz[3];
            const char* q = rest;
            for (int a = 0; a < 3; ++a) {
                xyz[a] = std::strtof(q, &end);  // correctly rounded decimal -> f32, like Rust's str::parse::<f32>
                if (end == q || (size_t)(end - data.c_str()) > le)
                    throw std::runtime_error("could not load mesh from file: bad '" + key + "' at line " +
                                             std::to_string(lineno));
                q = end;
            }
            auto& dst = key == "v" ? pos : nrm;
            dst.insert(dst.end(), xyz, xyz + 3);
        } else if (key == "f") {
            FaceVert fv[8];
            int n = 0;
            size_t i = k;
            while (i < le) {
                while (i < le && (data[i] == ' ' || data[i] == '\t')) ++i;
                size_t tb = i;
                while (i < le && data[i] != ' ' && data[i] != '\t') ++i;
                if (i == tb) break;
                if (n == 8 || !parse_face_vert(data.c_str() + tb, i - tb, pos.size() / 3, nrm.size() / 3, fv[n]))
                    throw std::runtime_error("could not load mesh from file: bad face at line " + std::to_string(lineno));
                ++n;
            }
            if (n != 3)
                // tobj keeps polygons un-triangulated and the reference then rejects or mis-reads them
                // (load.rs:251-256); refuse them outright.
                throw std::runtime_error("number of indices not divisible by 3: face with " + std::to_string(n) +
                                         " vertices at line " + std::to_string(lineno));
            face_verts.insert(face_verts.end(), fv, fv + 3);
        } else if (key == "o" || key == "g") {
            // A new object/group closes the current model once it has faces; only the first model is used
            // (load.rs:225-229).
            if (!face_verts.empty()) {
                done = true;
            } else {
                size_t nb = k;
                while (nb < le && (data[nb] == ' ' || data[nb] == '\t')) ++nb;
                if (nb < le) name = data.substr(nb, le - nb);
            }
        }
        // vt, usemtl, mtllib, s: no effect on what load_mesh reads (materials are refused, load.rs:213).
    }
    if (face_verts.empty()) throw std::runtime_error("expected exactly one model, got: 0");

    // export_faces_multi_index: vertices are re-indexed in order of first use by the faces.
    out = ObjMesh{};
    out.name = name;
    std::unordered_map<long, uint32_t> vmap, nmap;
    vmap.reserve(pos.size() / 3);
    for (const FaceVert& fv : face_verts) {
        auto it = vmap.find(fv.v);
        if (it == vmap.end()) {
            uint32_t next = (uint32_t)vmap.size();
            vmap.emplace(fv.v, next);
            out.positions.insert(out.positions.end(), pos.begin() + fv.v * 3, pos.begin() + fv.v * 3 + 3);
            out.indices.push_back(next);
        } else {
            out.indices.push_back(it->second);
        }
        if (fv.vn >= 0 && nmap.find(fv.vn) == nmap.end()) {
            nmap.emplace(fv.vn, (uint32_t)nmap.size());
            out.normals.insert(out.normals.end(), nrm.begin() + fv.vn * 3, nrm.begin() + fv.vn * 3 + 3);
        }
    }
}

}  // namespace rt
