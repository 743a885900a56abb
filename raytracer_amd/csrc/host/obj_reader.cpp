// OBJ(.gz) reader standing in for tobj 4.0.0 `load_obj_buf` with default LoadOptions as the reference
// calls it (load.rs:207-215): no triangulation, multi-index export, positions parsed as f32.
// Only what load_mesh consumes is produced: the FIRST model's positions / normals / indices.
#include <zlib.h>

#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <functional>
#include <stdexcept>
#include <future>
#include <thread>

#include "scene_host.h"
#include "worker.h"

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

// One face corner as written: "v", "v/vt", "v//vn", "v/vt/vn"; 1-based, negative = relative to the counts so far
// (resolved by the caller, who knows them).
struct RawCorner {
    long v = 0, vn = 0;
    bool has_vn = false;
};
bool parse_corner(const char* tok, size_t len, RawCorner& out) {
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
    out.v = idx[0], out.vn = idx[2], out.has_vn = have[2];
    return true;
}

// What one worker found in its run of lines.  Counts "so far" inside the run are kept with every face so that
// relative (negative) indices can be resolved once the runs before it have been counted.
struct Run {
    std::vector<float> pos, nrm;
    struct Face {
        RawCorner c[3];
        size_t pos_before, nrm_before;  // v / vn lines of this run above the face
        size_t lineno;
    };
    std::vector<Face> faces;
    size_t error_line = 0;  // first bad line (1-based), 0 = none
    std::string error;
};

void parse_run(const std::string& data, const std::vector<size_t>& line_start, size_t first, size_t last, Run& out) {
    auto fail = [&](size_t lineno, std::string m) {
        if (!out.error_line) out.error_line = lineno, out.error = std::move(m);
    };
    for (size_t li = first; li < last && !out.error_line; ++li) {
        size_t b = line_start[li], e = line_start[li + 1] - 1;  // e: the newline (or the buffer's end)
        const size_t lineno = li + 1;
        while (b < e && (data[b] == ' ' || data[b] == '\t')) ++b;
        size_t le = e;
        while (le > b && (data[le - 1] == '\r' || data[le - 1] == ' ' || data[le - 1] == '\t')) --le;
        if (b >= le || data[b] == '#') continue;
        size_t k = b;
        while (k < le && data[k] != ' ' && data[k] != '\t') ++k;
        const size_t klen = k - b;
        const bool is_v = klen == 1 && data[b] == 'v', is_vn = klen == 2 && data[b] == 'v' && data[b + 1] == 'n';
        if (is_v || is_vn) {
            const char* q = data.c_str() + k;  // NUL-terminated buffer: strtof stops at the line's end anyway
            char* end = nullptr;
            float xyz[3];
            for (int a = 0; a < 3; ++a) {
                xyz[a] = std::strtof(q, &end);  // correctly rounded decimal -> f32, like Rust's str::parse::<f32>
                if (end == q || (size_t)(end - data.c_str()) > le) {
                    fail(lineno, std::string("could not load mesh from file: bad '") + (is_v ? "v" : "vn") + "' at line " + std::to_string(lineno));
                    break;
                }
                q = end;
            }
            auto& dst = is_v ? out.pos : out.nrm;
            dst.insert(dst.end(), xyz, xyz + 3);
        } else if (klen == 1 && data[b] == 'f') {
            Run::Face f;
            int n = 0;
            size_t i = k;
            bool bad = false;
            while (i < le) {
                while (i < le && (data[i] == ' ' || data[i] == '\t')) ++i;
                size_t tb = i;
                while (i < le && data[i] != ' ' && data[i] != '\t') ++i;
                if (i == tb) break;
                RawCorner c;
                if (n == 8 || !parse_corner(data.c_str() + tb, i - tb, c)) {
                    bad = true;
                    break;
                }
                if (n < 3) f.c[n] = c;
                ++n;
            }
            if (bad) {
                fail(lineno, "could not load mesh from file: bad face at line " + std::to_string(lineno));
            } else if (n != 3) {
                // tobj keeps polygons un-triangulated and the reference then rejects or mis-reads them
                // (load.rs:251-256); refuse them outright.
                fail(lineno, "number of indices not divisible by 3: face with " + std::to_string(n) + " vertices at line " +
                                 std::to_string(lineno));
            } else {
                f.pos_before = out.pos.size() / 3, f.nrm_before = out.nrm.size() / 3, f.lineno = lineno;
                out.faces.push_back(f);
            }
        }
        // vt, usemtl, mtllib, s: no effect on what load_mesh reads (materials are refused, load.rs:213).
    }
}

}  // namespace

void load_obj_first_model(const std::string& path, ObjMesh& out) {
    const std::string data = read_all(path);

    // Pass 1 (one thread, a memchr per line): where the lines start, the model's name, and where the first model
    // ends — at the first object/group statement that follows a face (only the first model is used,
    // load.rs:225-229).
    std::vector<size_t> line_start;
    line_start.reserve(data.size() / 24 + 16);
    std::string name = "unnamed_object";
    bool seen_face = false;
    size_t n_lines = 0, sentinel = 0;
    for (size_t p = 0; p < data.size();) {
        sentinel = p;  // if the loop stops here, the kept lines end before this one
        const char* nl = (const char*)std::memchr(data.data() + p, '\n', data.size() - p);
        const size_t e = nl ? (size_t)(nl - data.data()) : data.size();
        size_t b = p;
        while (b < e && (data[b] == ' ' || data[b] == '\t')) ++b;
        // the keyword is the first whitespace-separated token (tobj: split_whitespace), so a bare `g` or `o` line —
        // `g\r\n` included — is an object/group statement too (its model is then tobj's "unnamed_object")
        size_t le = e;
        while (le > b && (data[le - 1] == '\r' || data[le - 1] == ' ' || data[le - 1] == '\t')) --le;
        if (b < le && (b + 1 == le || data[b + 1] == ' ' || data[b + 1] == '\t')) {
            if (data[b] == 'f') seen_face = true;
            if (data[b] == 'o' || data[b] == 'g') {
                if (seen_face) break;
                size_t nb = b + 1;
                while (nb < le && (data[nb] == ' ' || data[nb] == '\t')) ++nb;
                name = nb < le ? data.substr(nb, le - nb) : std::string("unnamed_object");
            }
        }
        line_start.push_back(p);
        ++n_lines;
        p = e + 1;
        sentinel = p;
    }
    line_start.push_back(sentinel);  // one past the last kept line's newline

    // Pass 2: the lines, in runs, on as many threads as pay (strtof is most of a large mesh's load time).
    const size_t n_threads = n_lines < 20000 ? 1 : std::min<size_t>(16, std::max(1u, std::thread::hardware_concurrency()));
    std::vector<Run> runs(n_threads);
    {
        // (futures, not bare threads: a failed thread start or a throwing run unwinds through joins, not std::terminate)
        std::vector<std::future<void>> pool;
        for (size_t t = 0; t < n_threads; ++t) {
            const size_t first = n_lines * t / n_threads, last = n_lines * (t + 1) / n_threads;
            if (t + 1 == n_threads) parse_run(data, line_start, first, last, runs[t]);
            else pool.push_back(run_async([&data, &line_start, first, last, &runs, t]() { parse_run(data, line_start, first, last, runs[t]); }));
        }
        for (auto& job : pool) job.get();
    }
    const Run* first_error = nullptr;
    for (const Run& r : runs)
        if (r.error_line && (!first_error || r.error_line < first_error->error_line)) first_error = &r;

    // Splice the runs in order; relative indices count the v / vn lines above the face.
    std::vector<float> pos, nrm;
    std::vector<FaceVert> face_verts;  // 3 per face of the first model
    size_t total_pos = 0, total_nrm = 0, total_faces = 0;
    for (const Run& r : runs) total_pos += r.pos.size(), total_nrm += r.nrm.size(), total_faces += r.faces.size();
    pos.reserve(total_pos), nrm.reserve(total_nrm), face_verts.reserve(3 * total_faces);
    size_t bad_face_line = 0;
    for (const Run& r : runs) {
        const size_t pos_base = pos.size() / 3, nrm_base = nrm.size() / 3;
        for (const Run::Face& f : r.faces) {
            const size_t n_pos = pos_base + f.pos_before, n_nrm = nrm_base + f.nrm_before;
            for (int k = 0; k < 3; ++k) {
                FaceVert fv;
                fv.v = f.c[k].v < 0 ? (long)n_pos + f.c[k].v : f.c[k].v - 1;
                fv.vn = !f.c[k].has_vn ? -1 : (f.c[k].vn < 0 ? (long)n_nrm + f.c[k].vn : f.c[k].vn - 1);
                if (!(fv.v >= 0 && (size_t)fv.v < n_pos && (fv.vn < 0 || (size_t)fv.vn < n_nrm)) && !bad_face_line)
                    bad_face_line = f.lineno;
                face_verts.push_back(fv);
            }
        }
        pos.insert(pos.end(), r.pos.begin(), r.pos.end());
        nrm.insert(nrm.end(), r.nrm.begin(), r.nrm.end());
    }
    if (first_error && (!bad_face_line || first_error->error_line < bad_face_line)) throw std::runtime_error(first_error->error);
    if (bad_face_line) throw std::runtime_error("could not load mesh from file: bad face at line " + std::to_string(bad_face_line));
    if (face_verts.empty()) throw std::runtime_error("expected exactly one model, got: 0");

    // export_faces_multi_index: vertices are re-indexed in order of first use by the faces.
    out = ObjMesh{};
    out.name = name;
    std::vector<int32_t> vmap(pos.size() / 3, -1);
    std::vector<uint8_t> nseen(nrm.size() / 3, 0);
    out.positions.reserve(pos.size()), out.indices.reserve(face_verts.size());
    for (const FaceVert& fv : face_verts) {
        int32_t& slot = vmap[(size_t)fv.v];
        if (slot < 0) {
            slot = (int32_t)(out.positions.size() / 3);
            out.positions.insert(out.positions.end(), pos.begin() + fv.v * 3, pos.begin() + fv.v * 3 + 3);
        }
        out.indices.push_back((uint32_t)slot);
        if (fv.vn >= 0 && !nseen[(size_t)fv.vn]) {
            nseen[(size_t)fv.vn] = 1;
            out.normals.insert(out.normals.end(), nrm.begin() + fv.vn * 3, nrm.begin() + fv.vn * 3 + 3);
        }
    }
}

}  // namespace rt
