#include "worker.h"

#include <execinfo.h>
#include <pthread.h>
#include <signal.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <cstdlib>
#include <cstring>
#include <string>

namespace rt {

BuildKnobs BuildKnobs::from_environment() {
    BuildKnobs k;
    if (const char* v = getenv("RT_LEAF")) k.leaf = (uint32_t)std::max(0, atoi(v));
    if (const char* v = getenv("RT_CT")) k.traversal_cost = atof(v);
    if (const char* v = getenv("RT_BUILD")) k.device_build = std::string(v) == "device";
    k.debug = getenv("RT_BUILD_DEBUG") != nullptr;
    if (const char* v = getenv("RT_BUILD_SMALL")) k.small = atoi(v);
    if (const char* v = getenv("RT_BUILD_THREADS")) k.one_thread = atoi(v) == 1;
    return k;
}

namespace {
std::atomic<int> g_forked_tasks{0};
std::atomic<bool> g_workers_disabled{false};
}  // namespace

// ---- the pool
struct PoolTask {
    std::function<void()> fn;
    std::exception_ptr error;
    std::atomic<int> state{0};  // 0 queued, 1 taken, 2 done
};

namespace {

struct Pool {
    std::mutex mutex;
    std::condition_variable work, done;  // work: a task was queued (or the pool ends); done: a task finished
    std::deque<TaskRef> queue;
    bool started = false;

    static void run(const TaskRef& t) {  // t->state is 1, set by the caller under the mutex
        try {
            t->fn();
        } catch (...) {
            t->error = std::current_exception();
        }
        t->fn = nullptr;
    }
    void finish(const TaskRef& t) {
        {
            std::lock_guard<std::mutex> lock(mutex);
            t->state.store(2, std::memory_order_release);
        }
        done.notify_all();
    }
    void thread_main() {
        AltSignalStack alt;
        std::unique_lock<std::mutex> lock(mutex);
        while (true) {
            work.wait(lock, [this]() { return !queue.empty(); });
            TaskRef t = std::move(queue.front());  // the oldest: the largest subtrees were forked first
            queue.pop_front();
            t->state.store(1, std::memory_order_relaxed);
            lock.unlock();
            run(t);
            finish(t);
            lock.lock();
        }
    }
    void start_threads() {  // under the mutex
        started = true;
        const unsigned want = std::min(15u, std::max(2u, std::thread::hardware_concurrency()) - 1u);
        for (unsigned i = 0; i < want; ++i) {
            try {
                std::thread([this]() { thread_main(); }).detach();
            } catch (const std::system_error&) {  // as many as there are to be had
                break;
            }
        }
    }
};
// The pool lives until the process ends and its threads are never joined: they sleep between scenes and go with the
// process.  (A destructor at exit would have to wake and join them — and in a child process made by fork() there is
// nothing to join: only the forking thread exists there.  The child starts a pool of its own at its first task.)
Pool* g_pool = nullptr;
std::once_flag g_pool_once;
Pool& pool() {
    std::call_once(g_pool_once, []() {
        g_pool = new Pool;
        // fork(): taken with the pool at rest (its mutex held across the call); the child gets a pool of its own — the
        // parent's is left as it was copied, mutex held, its condition variables knowing of sleepers that do not exist
        pthread_atfork([]() { g_pool->mutex.lock(); }, []() { g_pool->mutex.unlock(); }, []() { g_pool = new Pool; });
    });
    return *g_pool;
}

}  // namespace

TaskRef pool_submit(std::function<void()> f) {
    TaskRef t = std::make_shared<PoolTask>();
    t->fn = std::move(f);
    if (workers_disabled()) {
        t->state.store(1, std::memory_order_relaxed);
        Pool::run(t);
        t->state.store(2, std::memory_order_release);
        return t;
    }
    Pool& p = pool();
    {
        std::lock_guard<std::mutex> lock(p.mutex);
        if (!p.started) p.start_threads();
        p.queue.push_back(t);
    }
    p.work.notify_one();
    return t;
}

void pool_wait(const TaskRef& task) {
    Pool& p = pool();
    std::unique_lock<std::mutex> lock(p.mutex);
    while (task->state.load(std::memory_order_acquire) != 2) {
        TaskRef mine;
        if (task->state.load(std::memory_order_relaxed) == 0) {  // nobody has taken it: it is this thread's
            auto it = std::find(p.queue.begin(), p.queue.end(), task);
            if (it != p.queue.end()) p.queue.erase(it);
            mine = task;
        } else if (!p.queue.empty()) {  // somebody is at it: help with the newest task meanwhile
            mine = std::move(p.queue.back());
            p.queue.pop_back();
        }
        if (mine) {
            mine->state.store(1, std::memory_order_relaxed);
            lock.unlock();
            Pool::run(mine);
            p.finish(mine);
            lock.lock();
        } else {
            p.done.wait(lock);  // (woken by every task that finishes, and a newly queued one is found on the way round)
        }
    }
    lock.unlock();
    if (task->error) std::rethrow_exception(task->error);
}

void* os_block(size_t bytes) {
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) throw std::bad_alloc();
    return p;
}
void os_block_free(void* p, size_t bytes) noexcept { munmap(p, bytes); }

void set_workers_disabled(bool off) { g_workers_disabled.store(off, std::memory_order_release); }
bool workers_disabled() { return g_workers_disabled.load(std::memory_order_acquire); }

ForkSlot::ForkSlot() : taken(false) {
    if (workers_disabled()) return;
    taken = g_forked_tasks.fetch_add(1, std::memory_order_acq_rel) < kMaxForkedTasks;
    if (!taken) g_forked_tasks.fetch_sub(1, std::memory_order_acq_rel);
}
ForkSlot::~ForkSlot() {
    if (taken) g_forked_tasks.fetch_sub(1, std::memory_order_acq_rel);
}

namespace {

constexpr size_t kAltStackBytes = 1 << 16;
bool g_trace_on = false;  // written once, at library load

// Debugging aid (RT_SEGV_TRACE=1): the faulting thread's stack on SIGSEGV / SIGBUS / SIGABRT.  The message goes out
// before the unwinder runs (an unwinder that faults on a damaged stack must not take the message with it), and the
// handler runs on the thread's alternate stack where it has one (the loading thread, the pool's threads, a team's members).  A
// fault on a thread that blocks the signal — the HIP runtime's own threads block all of them at their entry — never
// gets here: the kernel ends the process with the default action.
void fatal_signal(int sig) {
    const char head[] = "librt_amd: fatal signal on thread ";
    (void)!write(2, head, sizeof head - 1);
    char num[24];
    int n = 0;
    unsigned long tid = (unsigned long)gettid();
    do num[n++] = (char)('0' + tid % 10), tid /= 10;
    while (tid);
    std::reverse(num, num + n);
    (void)!write(2, num, (size_t)n);
    const char tail[] = ", its stack:\n";
    (void)!write(2, tail, sizeof tail - 1);
    void* frames[64];
    const int depth = backtrace(frames, 64);
    backtrace_symbols_fd(frames, depth, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

struct SegvTrace {
    SegvTrace() {
        if (!getenv("RT_SEGV_TRACE")) return;  // at library load, on the loading thread
        g_trace_on = true;
        static AltSignalStack loader_stack;  // the loading thread keeps its alternate stack
        struct sigaction sa;
        std::memset(&sa, 0, sizeof sa);
        sa.sa_handler = fatal_signal;
        sa.sa_flags = SA_ONSTACK;
        for (int sig : {SIGSEGV, SIGBUS, SIGABRT}) sigaction(sig, &sa, nullptr);
    }
} g_segv_trace;

}  // namespace

bool segv_trace_enabled() { return g_trace_on; }

AltSignalStack::AltSignalStack() {
    if (!g_trace_on) return;
    mem = std::malloc(kAltStackBytes);
    if (!mem) return;
    stack_t ss;
    std::memset(&ss, 0, sizeof ss);
    ss.ss_sp = mem, ss.ss_size = kAltStackBytes;
    if (sigaltstack(&ss, nullptr) != 0) std::free(mem), mem = nullptr;
}

AltSignalStack::~AltSignalStack() {
    if (!mem) return;
    stack_t ss;
    std::memset(&ss, 0, sizeof ss);
    ss.ss_flags = SS_DISABLE;
    sigaltstack(&ss, nullptr);
    std::free(mem);
}

}  // namespace rt
