// Several GPUs of one process: the render path's one exchange step — the sum of the per-GPU f32
// framebuffers onto one of them — as an RCCL reduce over xGMI.  The reference has no counterpart (its pixels are
// shared memory under rayon, main.rs:150).  librccl is loaded on first use (dlopen), so a process that never
// opens a group — bench.py and the tests reduce through torch.distributed, which brings its own RCCL — never
// loads a second copy of it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/rt.h"

int rt_set_error_message(const std::string& m);  // api.cpp

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

const Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    static std::string error;
    std::call_once(once, []() {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            error = std::string("could not load librccl: ") + dlerror();
            return;
        }
        auto sym = [&](const char* n) {
            void* p = dlsym(r.handle, n);
            if (!p) error = std::string("librccl lacks ") + n;
            return p;
        };
        r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.Reduce = (decltype(r.Reduce))sym("ncclReduce");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    });
    if (!error.empty()) throw std::runtime_error(error);
    return r;
}

void check(ncclResult_t e, const char* what) {
    if (e != ncclSuccess) throw std::runtime_error(std::string(what) + " failed: " + rccl().GetErrorString(e));
}

}  // namespace

struct rt_group {
    std::vector<int> devices;
    std::vector<ncclComm_t> comms;
};

extern "C" {

int rt_group_create(const int* devices, int n, rt_group** out) {
    if (!devices || !out || n < 1) return rt_set_error_message("rt_group_create: bad argument");
    try {
        auto g = new rt_group;
        g->devices.assign(devices, devices + n);
        g->comms.assign((size_t)n, nullptr);
        try {
            check(rccl().CommInitAll(g->comms.data(), n, devices), "ncclCommInitAll");
        } catch (...) {
            delete g;
            throw;
        }
        *out = g;
        return 0;
    } catch (const std::exception& e) {
        return rt_set_error_message(std::string("rt_group_create: ") + e.what());
    }
}

int rt_group_reduce_sum(rt_group* g, float* const* d_bufs, size_t count, int root, void* const* streams) {
    if (!g || !d_bufs || root < 0 || (size_t)root >= g->devices.size())
        return rt_set_error_message("rt_group_reduce_sum: bad argument");
    if (count == 0) return rt_set_error_message("rt_group_reduce_sum: count must be positive");
    for (size_t i = 0; i < g->devices.size(); ++i)
        if (!d_bufs[i]) return rt_set_error_message("rt_group_reduce_sum: null buffer for member " + std::to_string(i));
    try {
        const Rccl& r = rccl();
        // Whatever way this call ends, the RCCL group it opened is closed again (a group left open would swallow
        // this thread's next collective) and the caller's device is current again.
        struct Scope {
            const Rccl& r;
            int prev = -1;
            bool open = false;
            explicit Scope(const Rccl& rc) : r(rc) { (void)hipGetDevice(&prev); }
            ~Scope() {
                if (open) (void)r.GroupEnd();
                if (prev >= 0) (void)hipSetDevice(prev);
            }
        } scope(r);
        check(r.GroupStart(), "ncclGroupStart");
        scope.open = true;
        for (size_t i = 0; i < g->devices.size(); ++i) {
            if (hipSetDevice(g->devices[i]) != hipSuccess) throw std::runtime_error("hipSetDevice failed");
            check(r.Reduce(d_bufs[i], d_bufs[i], count, ncclFloat, ncclSum, root, g->comms[i],
                           streams ? (hipStream_t)streams[i] : nullptr),
                  "ncclReduce");
        }
        scope.open = false;  // closed here, with its result checked
        check(r.GroupEnd(), "ncclGroupEnd");
        return 0;
    } catch (const std::exception& e) {
        return rt_set_error_message(std::string("rt_group_reduce_sum: ") + e.what());
    }
}

void rt_group_destroy(rt_group* g) {
    if (!g) return;
    try {
        for (ncclComm_t c : g->comms)
            if (c) (void)rccl().CommDestroy(c);
    } catch (...) {
    }
    delete g;
}

}  // extern "C"
