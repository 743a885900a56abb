// Worker threads of the host side (OBJ line parse, the two tree builds, the flattenings) and the knobs they take.
//
// Two rules, both from the cold-start crash of round 2:
//  * a worker reads no environment variable.  getenv is only safe while no other thread changes the environment,
//    and the HIP runtime's initialisation does change it (libhsa-runtime64 and librocprofiler-register import
//    setenv).  Every RT_* knob of scene creation is read once, by BuildKnobs::from_environment on the thread that
//    calls rt_scene_create, before the first worker exists, and handed down by value;
//  * a worker thread — the pool's, a team's — has (with RT_SEGV_TRACE set) an alternate signal stack before it does
//    any work, so that a fault on it — a stack overflow included — prints its stack instead of ending the process
//    silently.
#pragma once
#include <cstdint>
#include <algorithm>
#include <atomic>
#include <cstddef>
#include <functional>
#include <future>
#include <memory>
#include <new>
#include <system_error>
#include <thread>
#include <utility>
#include <vector>

namespace rt {

struct BuildKnobs {
    uint32_t leaf = 0;             // RT_LEAF: render-tree leaf size, 0 = the measured default
    double traversal_cost = 1.0;   // RT_CT: render-tree SAH traversal cost relative to one primitive test
    bool device_build = false;     // RT_BUILD=device
    bool debug = false;            // RT_BUILD_DEBUG: phase times on stderr
    int small = 0;                 // RT_BUILD_SMALL: device build, small-node threshold, 0 = default
    bool one_thread = false;       // RT_BUILD_THREADS=1: scene creation on the calling thread alone (same trees: a test holds them to it)
    static BuildKnobs from_environment();
};

bool segv_trace_enabled();  // RT_SEGV_TRACE was set when the library was loaded
// No worker threads for the work that follows (BuildKnobs::one_thread; process-wide, set from the knob by every
// rt_load_scene and rt_scene_create): tasks run where they are submitted, teams have one member, chunked loops one chunk.
void set_workers_disabled(bool off);
bool workers_disabled();
// this thread's alternate signal stack for as long as the object lives (nothing unless segv_trace_enabled())
struct AltSignalStack {
    void* mem = nullptr;
    AltSignalStack();
    ~AltSignalStack();
    AltSignalStack(const AltSignalStack&) = delete;
    AltSignalStack& operator=(const AltSignalStack&) = delete;
};

// A thread of its own for f, now: for the members of a team (run_team), which must run side by side.
template <class F>
auto run_async(F&& f) {
    return std::async(std::launch::async, [fn = std::forward<F>(f)]() mutable {
        AltSignalStack alt;
        return fn();
    });
}

// Everything else the set-up does beside the calling thread — parser blocks, forked subtrees, chunked loops, the
// render tree's build — is a task for the process's pool: up to 15 threads, started at the first task and kept (a
// thread per task it was at first: some sixty thread starts per scene, and as many malloc arenas left behind).
// submit() queues f — or runs it at once when workers are disabled; wait() returns when the task is done and
// rethrows what it threw.  A waiter does not sleep while there is work: it runs its own task itself if no pool
// thread has taken it yet, then other queued tasks — so a task may submit and wait for tasks of its own, nothing
// ever waits for a free thread, and a pool without a single thread (none could be started; a forked child process)
// only means that the waiters do the work.
struct PoolTask;
using TaskRef = std::shared_ptr<PoolTask>;
TaskRef pool_submit(std::function<void()> f);
void pool_wait(const TaskRef& task);
// A submitted task that is waited for when the scope ends, whatever ends it (a task works on its submitter's locals);
// get() waits now and rethrows what the task threw.
struct ScopedTask {
    TaskRef task;
    explicit ScopedTask(std::function<void()> f) : task(pool_submit(std::move(f))) {}
    ScopedTask(const ScopedTask&) = delete;
    ScopedTask& operator=(const ScopedTask&) = delete;
    void get() {
        const TaskRef t = std::move(task);
        task = nullptr;
        if (t) pool_wait(t);
    }
    ~ScopedTask() {
        if (!task) return;
        try {
            pool_wait(task);
        } catch (...) {
        }
    }
};

// How many forked tasks of the tree builds may be queued or running in this process at a time (both trees
// together).  A fork beyond it is simply not made: the parent builds that child itself, and the tree is the same
// either way.
constexpr int kMaxForkedTasks = 24;
struct ForkSlot {
    bool taken;
    ForkSlot();   // takes a slot if one is free
    ~ForkSlot();  // gives it back
    ForkSlot(const ForkSlot&) = delete;
    ForkSlot& operator=(const ForkSlot&) = delete;
};
// `child` as a task if a slot is free, else here and now; then `rest` here; both done when it returns.
template <class Child, class Rest>
void fork_join(Child&& child, Rest&& rest) {
    ForkSlot slot;
    TaskRef job;
    if (slot.taken) job = pool_submit([&child]() { child(); });
    else child();
    try {
        rest();
    } catch (...) {
        if (job) {
            try {
                pool_wait(job);
            } catch (...) {
            }
        }
        throw;
    }
    if (job) pool_wait(job);
}

// Memory for the workers' arrays and for what they leave with the scene.  Blocks of kOsBlockBytes and more come
// straight from the operating system and go straight back (mmap / munmap): a worker thread's malloc arena keeps
// what is freed in it — glibc trims a thread arena only past a threshold that rises with every large block freed,
// 64 MB in the end — and with the render tree's node array (14 MB on lucy.yaml), the copy the f64 parity arrays
// are made from (9 MB) and a tree piece or a sort's scratch of a few MB each made on some forty short-lived
// threads, a process that loaded scene after scene held 500 MB of such memory at rest (110 arena heaps of
// 17-23 MB; RSS 430 -> 690 MB over 300 loads of the shipped scenes, 100-170 MB with MALLOC_ARENA_MAX=2).
// With the blocks of 1 MB and more out of the arenas: 330 MB; of 384 KB and more: 210 MB, of 128 KB and more:
// 105 MB — but fresh pages from the OS are zeroed pages, and a scene's creation takes 2 ms more (17.5 -> 19.5)
// and 4 ms more: the threshold is where creation time does not move.  Smaller blocks are malloc's.
constexpr size_t kOsBlockBytes = 1 << 20;
void* os_block(size_t bytes);            // throws std::bad_alloc
void os_block_free(void* p, size_t bytes) noexcept;
template <class T>
struct OsBlocks {
    using value_type = T;
    OsBlocks() = default;
    template <class U>
    OsBlocks(const OsBlocks<U>&) noexcept {}
    T* allocate(size_t n) {
        const size_t bytes = n * sizeof(T);
        return static_cast<T*>(bytes >= kOsBlockBytes ? os_block(bytes) : ::operator new(bytes));
    }
    void deallocate(T* p, size_t n) noexcept {
        const size_t bytes = n * sizeof(T);
        if (bytes >= kOsBlockBytes) os_block_free(p, bytes);
        else ::operator delete(p);
    }
    template <class U>
    bool operator==(const OsBlocks<U>&) const noexcept { return true; }
    template <class U>
    bool operator!=(const OsBlocks<U>&) const noexcept { return false; }
};
// The same, and resize(n) / a sized constructor leave the new elements' memory as it is — for arrays of plain data
// that the next loop fills on several threads (the first touch of the pages included), where value-initialising
// them first would be one more pass, on one thread.  Every element must be assigned before it is read.
template <class T>
struct Uninit : OsBlocks<T> {
    template <class U>
    struct rebind {
        using other = Uninit<U>;
    };
    Uninit() = default;
    template <class U>
    Uninit(const Uninit<U>&) noexcept {}
    template <class U>
    void construct(U*) noexcept {}
    template <class U, class... A>
    void construct(U* p, A&&... a) {
        ::new ((void*)p) U(std::forward<A>(a)...);
    }
};

// [0, n) in contiguous chunks on up to `max_tasks` threads (the caller's included); fn(begin, end) writes disjoint
// elements.  Small ranges stay on the calling thread.
template <class F>
void parallel_chunks(size_t n, size_t min_chunk, size_t max_tasks, F&& fn) {
    const size_t tasks = workers_disabled() ? 1 : std::max<size_t>(1, std::min(max_tasks, n / std::max<size_t>(min_chunk, 1)));
    if (tasks == 1) return fn((size_t)0, n);
    std::vector<TaskRef> jobs;
    for (size_t t = 1; t < tasks; ++t) jobs.push_back(pool_submit([&fn, n, t, tasks]() { fn(n * t / tasks, n * (t + 1) / tasks); }));
    std::exception_ptr failed;
    try {
        fn((size_t)0, n / tasks);
    } catch (...) {
        failed = std::current_exception();
    }
    for (auto& j : jobs) {  // every chunk is done before anything unwinds past what they write
        try {
            pool_wait(j);
        } catch (...) {
            if (!failed) failed = std::current_exception();
        }
    }
    if (failed) std::rethrow_exception(failed);
}

// A team of threads (the caller is member 0) that runs fn(member, team) and meets at team.barrier() between the
// phases of one piece of work: the top-level sort and partition of the tree builds, whose phases are too short to
// start threads for each of them.  A member waiting at the barrier spins briefly, then yields.
struct Team {
    size_t members;
    explicit Team(size_t n) : members(n) {}
    void barrier() {
        const uint32_t phase = phase_.load(std::memory_order_acquire);
        if (arrived_.fetch_add(1, std::memory_order_acq_rel) + 1 == members) {
            arrived_.store(0, std::memory_order_relaxed);
            phase_.store(phase + 1, std::memory_order_release);
            return;
        }
        for (uint32_t spins = 0; phase_.load(std::memory_order_acquire) == phase; ++spins)
            if (spins > 256) std::this_thread::yield();
    }
    // [0, n) in `members` contiguous chunks: member m's
    std::pair<size_t, size_t> chunk(size_t m, size_t n) const { return {n * m / members, n * (m + 1) / members}; }

  private:
    std::atomic<uint32_t> arrived_{0}, phase_{0};
};
template <class F>
void run_team(size_t members, F&& fn) {
    Team team(workers_disabled() ? 1 : std::max<size_t>(members, 1));
    // the members start their work together or not at all: one that could not be started would be waited for at
    // the first barrier for ever
    std::atomic<int> go{-1};  // -1: wait, 0: leave, 1: run
    std::vector<std::future<void>> jobs;
    try {
        for (size_t m = 1; m < team.members; ++m)
            jobs.push_back(run_async([&fn, &team, &go, m]() {
                while (go.load(std::memory_order_acquire) < 0) std::this_thread::yield();
                if (go.load(std::memory_order_acquire) == 1) fn(m, team);
            }));
    } catch (...) {
        go.store(0, std::memory_order_release);
        for (auto& j : jobs) j.wait();
        throw;
    }
    go.store(1, std::memory_order_release);
    fn((size_t)0, team);
    for (auto& j : jobs) j.get();
}
// (a team is all or nothing: when a member's thread cannot be started, run_team throws std::system_error before any
// member has run, and the caller does the work on its own)

}  // namespace rt
