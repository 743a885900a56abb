#include "worker.h"

#include <execinfo.h>
#include <signal.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
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
// handler runs on the thread's alternate stack where it has one (the loading thread, every run_async worker).  A
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
