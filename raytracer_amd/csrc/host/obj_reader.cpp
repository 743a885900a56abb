// OBJ(.gz) reader standing in for tobj 4.0.0 `load_obj_buf` with default LoadOptions as the reference
// calls it (load.rs:207-215): no triangulation, multi-index export, positions parsed as f32.
// Only what load_mesh consumes is produced: the FIRST model's positions / normals / indices.
#include <zlib.h>

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <algorithm>
#include <functional>
#include <stdexcept>
#include <system_error>
#include <future>
#include <thread>

#include "scene_host.h"
#include "worker.h"

namespace rt {

namespace {

// The file's bytes as they are on disk.
std::string read_file(const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("could not open '" + path + "': " + std::strerror(errno));
    std::string raw;
    char buf[1 << 16];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) raw.append(buf, n);
    const bool bad = std::ferror(f) != 0;
    std::fclose(f);
    if (bad) throw std::runtime_error("could not read '" + path + "': read error");
    return raw;
}

// The text of the file, block by block: gunzipped when the bytes start with the gzip magic (the reference gunzips iff
// the extension is "gz", load.rs:318 — same outcome for well-formed inputs), as they are otherwise.  Inflating is most
// of a large mesh's load time and cannot be split over threads, but what follows can start before it ends:
// `on_block(begin, end, last)` is called with the range of `text` that each further run of whole lines occupies, as
// soon as it is there.  The memory the earlier blocks point into stays where it is unless the gzip trailer
// understated the size (several members); `before_move` is then called first.
struct BlockSource {
    std::string raw;
    std::string text;  // size(): room for the inflated text; `filled` bytes of it are there
    size_t filled = 0;

    template <class OnBlock, class BeforeMove>
    void run(const std::string& path, size_t block_bytes, OnBlock&& on_block, BeforeMove&& before_move) {
        raw = read_file(path);
        const bool gz = raw.size() >= 18 && (unsigned char)raw[0] == 0x1f && (unsigned char)raw[1] == 0x8b;
        size_t done = 0;  // text[0, done) has been handed out
        if (!gz) {
            text.swap(raw);
            filled = text.size();
            while (filled - done > 2 * block_bytes) {  // blocks of about block_bytes, cut at line ends
                const char* nl = (const char*)std::memchr(text.data() + done + block_bytes, '\n', filled - done - block_bytes);
                if (!nl) break;
                const size_t end = (size_t)(nl - text.data()) + 1;
                on_block(done, end, false);
                done = end;
            }
            on_block(done, filled, true);
            return;
        }
        // ISIZE, the last four bytes of a gzip member: its inflated size modulo 2^32
        uint32_t isize = 0;
        std::memcpy(&isize, raw.data() + raw.size() - 4, 4);
        text.resize(std::max<size_t>((size_t)isize, 2 * raw.size()) + 1);  // + the terminating NUL
        z_stream z;
        std::memset(&z, 0, sizeof z);
        if (inflateInit2(&z, 15 + 16) != Z_OK) throw std::runtime_error("could not read '" + path + "': zlib init failed");
        size_t fed = 0, member_begin = 0;
        int rc = Z_OK, members = 0;
        while (true) {
            if (z.avail_in == 0 && fed < raw.size()) {
                const size_t n = std::min<size_t>(raw.size() - fed, 1u << 30);
                z.next_in = (Bytef*)raw.data() + fed, z.avail_in = (uInt)n, fed += n;
            }
            if (filled + 1 >= text.size()) {
                before_move();
                text.resize(2 * text.size());
            }
            const size_t room = std::min<size_t>(text.size() - 1 - filled, block_bytes);
            z.next_out = (Bytef*)&text[filled], z.avail_out = (uInt)room;
            rc = inflate(&z, Z_NO_FLUSH);
            filled += room - z.avail_out;
            if (rc == Z_STREAM_END) {
                ++members, member_begin = filled;
                if ((z.avail_in > 0 || fed < raw.size()) && inflateReset(&z) == Z_OK) continue;  // another member?
                break;
            }
            if (rc != Z_OK) break;
            // whole lines only, and not before there is a block's worth of them
            size_t end = filled;
            while (end > done && text[end - 1] != '\n') --end;
            if (end - done >= block_bytes) on_block(done, end, false), done = end;
        }
        // what follows a complete member and is no gzip member is ignored, as gzread ignores it
        if (rc == Z_DATA_ERROR && members > 0 && filled == member_begin) rc = Z_STREAM_END;
        const std::string msg = z.msg ? z.msg : (rc == Z_BUF_ERROR ? "unexpected end of file" : "read error");
        inflateEnd(&z);
        if (rc != Z_STREAM_END) {
            before_move();
            throw std::runtime_error("could not read '" + path + "': " + msg);
        }
        text[filled] = 0;
        on_block(done, filled, true);
    }
};

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

// What one worker found in its block of lines.  Counts "so far" inside the block are kept with every face so that
// relative (negative) indices can be resolved once the blocks before it have been counted; line numbers count from
// the block's first line.
struct Run {
    std::vector<float, OsBlocks<float>> pos, nrm;
    struct Face {
        RawCorner c[3];
        size_t pos_before, nrm_before;  // v / vn lines of this block above the face
        size_t line;
    };
    std::vector<Face, OsBlocks<Face>> faces;
    // every object / group statement of the block: the first one that follows a face — of this block or an earlier
    // one — ends the first model (only the first model is used, load.rs:225-229)
    struct Mark {
        size_t line, pos_before, nrm_before, faces_before;
        std::string name;
    };
    std::vector<Mark> marks;
    size_t n_lines = 0;
    size_t error_line = 0;  // first bad line (1-based in the block), 0 = none; the parse stops there
    std::string error;      // its message, up to the line number
};

void parse_block(const char* text, size_t begin, size_t end, Run& out) {
    auto fail = [&](size_t line, std::string m) {
        if (!out.error_line) out.error_line = line, out.error = std::move(m);
    };
    size_t line = 0;
    for (size_t p = begin; p < end;) {
        const char* nl = (const char*)std::memchr(text + p, '\n', end - p);
        const size_t e = nl ? (size_t)(nl - text) : end;  // e: the newline (or the text's end)
        size_t b = p;
        p = e + 1;
        ++line;
        if (out.error_line) continue;  // (the lines are still counted)
        while (b < e && (text[b] == ' ' || text[b] == '\t')) ++b;
        size_t le = e;
        while (le > b && (text[le - 1] == '\r' || text[le - 1] == ' ' || text[le - 1] == '\t')) --le;
        if (b >= le || text[b] == '#') continue;
        // the keyword is the first whitespace-separated token (tobj: split_whitespace), so a bare `g` or `o` line —
        // `g\r\n` included — is an object/group statement too (its model is then tobj's "unnamed_object")
        size_t k = b;
        while (k < le && text[k] != ' ' && text[k] != '\t') ++k;
        const size_t klen = k - b;
        const bool is_v = klen == 1 && text[b] == 'v', is_vn = klen == 2 && text[b] == 'v' && text[b + 1] == 'n';
        if (is_v || is_vn) {
            const char* q = text + k;  // strtof stops at the line's end: a block ends with its newline or the NUL
            char* stop = nullptr;
            float xyz[3];
            for (int a = 0; a < 3; ++a) {
                xyz[a] = std::strtof(q, &stop);  // correctly rounded decimal -> f32, like Rust's str::parse::<f32>
                if (stop == q || (size_t)(stop - text) > le) {
                    fail(line, std::string("could not load mesh from file: bad '") + (is_v ? "v" : "vn") + "' at line ");
                    break;
                }
                q = stop;
            }
            auto& dst = is_v ? out.pos : out.nrm;
            dst.insert(dst.end(), xyz, xyz + 3);
        } else if (klen == 1 && text[b] == 'f') {
            Run::Face f;
            int n = 0;
            size_t i = k;
            bool bad = false;
            while (i < le) {
                while (i < le && (text[i] == ' ' || text[i] == '\t')) ++i;
                size_t tb = i;
                while (i < le && text[i] != ' ' && text[i] != '\t') ++i;
                if (i == tb) break;
                RawCorner c;
                if (n == 8 || !parse_corner(text + tb, i - tb, c)) {
                    bad = true;
                    break;
                }
                if (n < 3) f.c[n] = c;
                ++n;
            }
            if (bad) {
                fail(line, "could not load mesh from file: bad face at line ");
            } else if (n != 3) {
                // tobj keeps polygons un-triangulated and the reference then rejects or mis-reads them
                // (load.rs:251-256); refuse them outright.
                fail(line, "number of indices not divisible by 3: face with " + std::to_string(n) + " vertices at line ");
            } else {
                f.pos_before = out.pos.size() / 3, f.nrm_before = out.nrm.size() / 3, f.line = line;
                out.faces.push_back(f);
            }
        } else if (klen == 1 && (text[b] == 'o' || text[b] == 'g')) {
            size_t nb = k;
            while (nb < le && (text[nb] == ' ' || text[nb] == '\t')) ++nb;
            out.marks.push_back({line, out.pos.size() / 3, out.nrm.size() / 3, out.faces.size(),
                                 nb < le ? std::string(text + nb, le - nb) : std::string("unnamed_object")});
        }
        // vt, usemtl, mtllib, s: no effect on what load_mesh reads (materials are refused, load.rs:213).
    }
    out.n_lines = line;
}

}  // namespace

void load_obj_first_model(const std::string& path, ObjMesh& out) {
    // The blocks of lines are parsed as they come out of the inflater, each as a task of the pool (strtof is most of
    // what is left of a large mesh's load time), at most kInFlight of them at a time.
    constexpr size_t kBlockBytes = 256 << 10, kInFlight = 16;
    BlockSource source;
    std::vector<std::unique_ptr<Run>> runs;
    std::vector<TaskRef> jobs;
    size_t joined = 0;
    auto join_all = [&]() {
        for (; joined < jobs.size(); ++joined) pool_wait(jobs[joined]);
    };
    try {
        source.run(
            path, kBlockBytes,
            [&](size_t begin, size_t end, bool last) {
                runs.push_back(std::make_unique<Run>());
                Run* run = runs.back().get();
                const char* text = source.text.data();
                if (last || workers_disabled()) return parse_block(text, begin, end, *run);  // nothing left to overlap it with
                if (jobs.size() - joined >= kInFlight) pool_wait(jobs[joined++]);
                jobs.push_back(pool_submit([text, begin, end, run]() { parse_block(text, begin, end, *run); }));
            },
            join_all);
    } catch (...) {
        for (; joined < jobs.size(); ++joined) {  // no task outlives the text it reads
            try {
                pool_wait(jobs[joined]);
            } catch (...) {
            }
        }
        throw;
    }
    join_all();

    // Splice the blocks in order, up to the first model's end; relative indices count the v / vn lines above the face.
    std::vector<float> pos, nrm;
    std::vector<FaceVert> face_verts;  // 3 per face of the first model
    {
        size_t total_pos = 0, total_nrm = 0, total_faces = 0;
        for (const auto& r : runs) total_pos += r->pos.size(), total_nrm += r->nrm.size(), total_faces += r->faces.size();
        pos.reserve(total_pos), nrm.reserve(total_nrm), face_verts.reserve(3 * total_faces);
    }
    std::string name = "unnamed_object", error;
    size_t line_base = 0, error_line = 0, bad_face_line = 0;
    bool seen_face = false;
    for (const auto& rp : runs) {
        const Run& r = *rp;
        const Run::Mark* cut = nullptr;
        for (const Run::Mark& m : r.marks) {
            if (seen_face || m.faces_before > 0) {
                cut = &m;
                break;
            }
            name = m.name;
        }
        if (r.error_line && (!cut || r.error_line < cut->line)) {
            error_line = line_base + r.error_line, error = r.error + std::to_string(error_line);
            cut = nullptr;
        }
        const size_t n_faces = cut ? cut->faces_before : r.faces.size();
        const size_t n_pos = cut ? cut->pos_before : r.pos.size() / 3, n_nrm = cut ? cut->nrm_before : r.nrm.size() / 3;
        const size_t pos_base = pos.size() / 3, nrm_base = nrm.size() / 3;
        for (size_t i = 0; i < n_faces; ++i) {
            const Run::Face& f = r.faces[i];
            const size_t have_pos = pos_base + f.pos_before, have_nrm = nrm_base + f.nrm_before;
            for (int k = 0; k < 3; ++k) {
                FaceVert fv;
                fv.v = f.c[k].v < 0 ? (long)have_pos + f.c[k].v : f.c[k].v - 1;
                fv.vn = !f.c[k].has_vn ? -1 : (f.c[k].vn < 0 ? (long)have_nrm + f.c[k].vn : f.c[k].vn - 1);
                if (!(fv.v >= 0 && (size_t)fv.v < have_pos && (fv.vn < 0 || (size_t)fv.vn < have_nrm)) && !bad_face_line)
                    bad_face_line = line_base + f.line;
                face_verts.push_back(fv);
            }
        }
        pos.insert(pos.end(), r.pos.begin(), r.pos.begin() + 3 * n_pos);
        nrm.insert(nrm.end(), r.nrm.begin(), r.nrm.begin() + 3 * n_nrm);
        if (cut || error_line) break;
        seen_face = seen_face || !r.faces.empty();
        line_base += r.n_lines;
    }
    if (error_line && (!bad_face_line || error_line < bad_face_line)) throw std::runtime_error(error);
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
