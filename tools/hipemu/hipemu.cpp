// hipemu runtime: runs one workgroup at a time; every work-item is a ucontext fiber.  See hip/hip_runtime.h.
#include "hip/hip_runtime.h"
#undef threadIdx
#undef blockIdx
#undef blockDim
#undef gridDim
#include <sys/mman.h>
#include <deque>
#include <map>
#include <vector>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

namespace hipemu {
Idx g_tid, g_bid;
dim3 g_bdim, g_gdim;

namespace {
enum State { READY, AT_BARRIER, AT_WAVE, DONE };
constexpr size_t STACK = 256 * 1024;
struct Dma { void* dst; int bytes; unsigned char data[16]; };
struct Fiber {
    void* rsp;                 // saved stack pointer while the fiber is switched out
    State st;
    Idx tid;
    WaveReduce reduce;         // set by wave_collective: run ONCE per wave when every live lane has arrived
    std::deque<Dma> dma;       // this lane's DMA instructions in flight (deferred completion model)
    alignas(32) unsigned char deposit[DEPOSIT];
};
struct Wave {
    alignas(32) unsigned char snap[64][DEPOSIT];
    alignas(32) unsigned char result[64][RESULT];
    unsigned long long snap_mask;
};
std::vector<Fiber> fibers;
std::vector<void*> stacks;
std::vector<Wave> waves;
void* sched_rsp = nullptr;
int cur = -1, nthreads = 0;
const std::function<void()>* body_fn = nullptr;
bool in_kernel = false;
bool reverse_order = false;    // run the ready work-items of a workgroup from the last to the first (and the workgroups of a grid likewise)
bool strict_barrier = false;   // abort when a barrier is released although part of the workgroup has ended (the hardware allows it)

// A minimal x86-64 SysV context switch (callee-saved registers + stack pointer): ucontext's swapcontext makes a sigprocmask
// system call per switch, and a wave collective is 64 switches.
extern "C" void hipemu_switch(void** save_rsp, void* load_rsp);
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size hipemu_switch, .-hipemu_switch
)");
void retire(Fiber& f, size_t keep) {
    while (f.dma.size() > keep) {
        const Dma& d = f.dma.front();
        memcpy(d.dst, d.data, d.bytes);
        f.dma.pop_front();
    }
}
void fiber_main() {
    (*body_fn)();
    retire(fibers[cur], 0);    // a kernel's memory operations have all completed when it ends
    fibers[cur].st = DONE;
    hipemu_switch(&fibers[cur].rsp, sched_rsp);
    abort();                   // a finished fiber is never resumed
}
void yield(State s) {
    fibers[cur].st = s;
    hipemu_switch(&fibers[cur].rsp, sched_rsp);
}
void* stack_for(int i) {
    while ((int)stacks.size() <= i) {
        void* p = mmap(nullptr, STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) { fprintf(stderr, "hipemu: cannot map a fiber stack\n"); abort(); }
        stacks.push_back(p);
    }
    return stacks[i];
}

void run_block() {
    const int n = nthreads;
    if ((int)fibers.size() < n) fibers.resize(n);
    const int nw = (n + 63) / 64;
    if ((int)waves.size() < nw) waves.resize(nw);
    for (int i = 0; i < n; ++i) {
        Fiber& f = fibers[i];
        // fresh stack: six zeroed callee-saved registers, the entry point as the return address, and a slot that leaves the
        // stack pointer congruent to 8 mod 16 at the entry, as after a call
        void** top = reinterpret_cast<void**>(static_cast<char*>(stack_for(i)) + STACK);
        top[-1] = nullptr;
        top[-2] = reinterpret_cast<void*>(&fiber_main);
        for (int r = 3; r <= 8; ++r) top[-r] = nullptr;
        f.rsp = &top[-8];
        f.reduce = nullptr;
        f.dma.clear();
        f.st = READY;
        f.tid.x = i % g_bdim.x;
        f.tid.y = (i / g_bdim.x) % g_bdim.y;
        f.tid.z = i / (g_bdim.x * g_bdim.y);
    }
    int live = n;
    while (live > 0) {
        bool progressed = false;
        for (int ii = 0; ii < (n + 63) / 64 * 64; ++ii) {
            // which ready WAVE runs first must not matter to a correct kernel; the lanes of a wave keep their order (the hardware
            // serialises the lanes of one LDS atomic in lane order: kernels may rely on that)
            const int nwv = (n + 63) / 64;
            const int i = reverse_order ? (nwv - 1 - ii / 64) * 64 + (ii & 63) : ii;
            if (i >= n) continue;                                                          // a lane beyond a ragged last wave
            if (fibers[i].st != READY) continue;
            cur = i;
            g_tid = fibers[i].tid;
            hipemu_switch(&sched_rsp, fibers[i].rsp);
            progressed = true;
            if (fibers[i].st == DONE) --live;
        }
        // wave collectives: release a wave once all of its live lanes have arrived
        for (int w = 0; w < nw; ++w) {
            int waiting = 0, alive = 0;
            const int lo = w * 64, hi = std::min(n, lo + 64);
            for (int i = lo; i < hi; ++i) {
                if (fibers[i].st == AT_WAVE) ++waiting;
                if (fibers[i].st != DONE) ++alive;
            }
            if (waiting > 0 && waiting == alive) {
                waves[w].snap_mask = 0;
                WaveReduce red = nullptr;
                for (int i = lo; i < hi; ++i)
                    if (fibers[i].st == AT_WAVE) {
                        memcpy(waves[w].snap[i - lo], fibers[i].deposit, DEPOSIT);
                        waves[w].snap_mask |= 1ull << (i - lo);
                        fibers[i].st = READY;
                        if (fibers[i].reduce) red = fibers[i].reduce;
                        fibers[i].reduce = nullptr;
                    }
                if (red) red(waves[w].snap, waves[w].snap_mask, waves[w].result);
                progressed = true;
            }
        }
        // workgroup barrier: release once every live work-item waits at it
        int at_bar = 0;
        for (int i = 0; i < n; ++i) at_bar += fibers[i].st == AT_BARRIER;
        if (at_bar > 0 && at_bar == live) {
            if (strict_barrier && live < n) {
                fprintf(stderr, "hipemu: strict barriers: block (%u,%u,%u) releases a barrier with %d of %d work-items already ended\n",
                        g_bid.x, g_bid.y, g_bid.z, n - live, n);
                abort();
            }
            for (int i = 0; i < n; ++i)
                if (fibers[i].st == AT_BARRIER) fibers[i].st = READY;
            progressed = true;
        }
        if (!progressed) {
            fprintf(stderr, "hipemu: deadlock in block (%u,%u,%u): %d live work-items, %d at the barrier; the others wait in a wave "
                            "collective that their wave mates never reach (divergent collective / barrier)\n",
                    g_bid.x, g_bid.y, g_bid.z, live, at_bar);
            abort();
        }
    }
}
}  // namespace

namespace {
void on_segv(int sig) {
    void* bt[64];
    const int n = backtrace(bt, 64);
    fprintf(stderr, "hipemu: signal %d in block (%u,%u,%u) work-item %d\n", sig, g_bid.x, g_bid.y, g_bid.z, cur);
    backtrace_symbols_fd(bt, n, 2);
    _exit(139);
}
}  // namespace

// ---- guarded device allocations ---------------------------------------------------------------------------------
namespace {
constexpr size_t GUARD = 4096;
constexpr unsigned char GUARD_BYTE = 0xA5;
std::map<char*, size_t>& live_allocs() { static std::map<char*, size_t> m; return m; }
unsigned long long launch_count = 0;
bool zone_clean(const char* z) {
    for (size_t i = 0; i < GUARD; ++i) if ((unsigned char)z[i] != GUARD_BYTE) return false;
    return true;
}
void check_guards(const char* when) {
    for (auto& kv : live_allocs()) {
        const bool lo = zone_clean(kv.first - GUARD), hi = zone_clean(kv.first + kv.second);
        if (lo && hi) continue;
        const char* z = lo ? kv.first + kv.second : kv.first - GUARD;
        size_t first = 0;
        while ((unsigned char)z[first] == GUARD_BYTE) ++first;
        size_t last = GUARD - 1;
        while ((unsigned char)z[last] == GUARD_BYTE) --last;
        fprintf(stderr, "hipemu: %s: write %s a device allocation of %zu bytes (bytes %zu..%zu of the %s red zone), launch #%llu\n", when,
                lo ? "past the end of" : "before the start of", kv.second, first, last, lo ? "upper" : "lower", launch_count);
        void* bt[48];
        backtrace_symbols_fd(bt, backtrace(bt, 48), 2);
        abort();
    }
}
}  // namespace
void* dev_alloc(size_t n) {
    const size_t body = (n + 255) / 256 * 256;
    char* raw = (char*)aligned_alloc(4096, (body + 2 * GUARD + 4095) / 4096 * 4096);
    if (!raw) return nullptr;
    memset(raw, GUARD_BYTE, GUARD);
    memset(raw + GUARD + n, GUARD_BYTE, body - n + GUARD);        // the red zone starts at the first byte past the request
    live_allocs()[raw + GUARD] = n;
    return raw + GUARD;
}
void dev_free(void* p) {
    if (!p) return;
    auto it = live_allocs().find((char*)p);
    if (it == live_allocs().end()) { fprintf(stderr, "hipemu: hipFree of a pointer hipMalloc did not return\n"); abort(); }
    check_guards("hipFree");
    live_allocs().erase(it);
    free((char*)p - GUARD);
}

void launch(dim3 grid, dim3 block, size_t, const std::function<void()>& body) {
    static bool traced = false;
    if (!traced && getenv("HIPEMU_TRACE")) {       // a backtrace instead of a bare crash (set before the first launch)
        static char alt[1 << 16];
        stack_t ss{alt, 0, sizeof(alt)};
        sigaltstack(&ss, nullptr);
        struct sigaction sa{};
        sa.sa_handler = on_segv;
        sa.sa_flags = SA_ONSTACK;
        sigaction(SIGSEGV, &sa, nullptr);
        sigaction(SIGBUS, &sa, nullptr);
        sigaction(SIGILL, &sa, nullptr);       // -fsanitize-trap builds
        traced = true;
    }
    static const bool log = getenv("HIPEMU_LOG") != nullptr;
    static const bool env_once = [] {            // HIPEMU_ORDER=reverse: the same as hipemu_set_reverse_order(1), for whole test runs
        const char* o = getenv("HIPEMU_ORDER");
        if (o && !strcmp(o, "reverse")) reverse_order = true;
        return true;
    }();
    (void)env_once;
    if (log) fprintf(stderr, "hipemu: launch grid (%u,%u,%u) block (%u,%u,%u)\n", grid.x, grid.y, grid.z, block.x, block.y, block.z);
    if (in_kernel) { fprintf(stderr, "hipemu: nested launch\n"); abort(); }
    nthreads = (int)(block.x * block.y * block.z);
    if (nthreads <= 0 || nthreads > 1024) { fprintf(stderr, "hipemu: bad block size %d\n", nthreads); abort(); }
    in_kernel = true;
    body_fn = &body;
    g_bdim = block;
    g_gdim = grid;
    const unsigned long long nblk = (unsigned long long)grid.x * grid.y * grid.z;
    for (unsigned long long b = 0; b < nblk; ++b) {
        const unsigned long long lin = reverse_order ? nblk - 1 - b : b;
        g_bid = Idx{(unsigned)(lin % grid.x), (unsigned)((lin / grid.x) % grid.y), (unsigned)(lin / ((unsigned long long)grid.x * grid.y))};
        run_block();
    }
    in_kernel = false;
    body_fn = nullptr;
    ++launch_count;
    check_guards("kernel launch");
}

void syncthreads() { yield(AT_BARRIER); }

void wave_exchange(const void* mine, size_t bytes) {
    if (bytes > DEPOSIT) { fprintf(stderr, "hipemu: wave exchange of %zu bytes\n", bytes); abort(); }
    memcpy(fibers[cur].deposit, mine, bytes);
    yield(AT_WAVE);
}
const void* wave_collective(const void* mine, size_t bytes, WaveReduce reduce) {
    if (bytes > DEPOSIT) { fprintf(stderr, "hipemu: wave collective of %zu bytes\n", bytes); abort(); }
    memcpy(fibers[cur].deposit, mine, bytes);
    fibers[cur].reduce = reduce;
    yield(AT_WAVE);
    return waves[cur / 64].result[cur & 63];
}
int dma_eager = -1;          // -1: read HIPEMU_DMA at the first DMA; hipemu_set_dma_eager() overrides
void set_strict_barrier(bool on) { strict_barrier = on; }
void set_reverse_order(bool on) { reverse_order = on; }
void wave_sync() { yield(AT_WAVE); }
void dma_issue(const void* src, void* dst, int bytes) {
    if (dma_eager < 0) dma_eager = getenv("HIPEMU_DMA") && !strcmp(getenv("HIPEMU_DMA"), "eager");
    if (dma_eager || bytes > 16) { memcpy(dst, src, bytes); return; }
    Dma d;
    d.dst = dst; d.bytes = bytes;
    memcpy(d.data, src, bytes);
    fibers[cur].dma.push_back(d);
}
void waitcnt_vm(int n) { if (cur >= 0 && in_kernel) retire(fibers[cur], n < 0 ? 0 : (size_t)n); }
const void* wave_slot(int lane) {
    if (lane < 0 || lane > 63) return nullptr;
    const Wave& w = waves[cur / 64];
    return ((w.snap_mask >> lane) & 1) ? w.snap[lane] : nullptr;
}
int lane_id() { return cur & 63; }
unsigned long long wave_live_mask() {
    unsigned long long m = 0;
    const int lo = (cur / 64) * 64, hi = std::min(nthreads, lo + 64);
    for (int i = lo; i < hi; ++i)
        if (fibers[i].st != DONE) m |= 1ull << (i - lo);
    return m;
}
}  // namespace hipemu

// tests switch the DMA completion model between kernels (deferred: the latest legal completion; eager: the earliest)
extern "C" void hipemu_set_dma_eager(int on) { hipemu::dma_eager = on ? 1 : 0; }
// every work-item of a workgroup must take part in every barrier (checks that role-split kernels mirror each other's barriers)
// execution order of work-items and workgroups reversed: results of a kernel that claims a fixed summation order must not change
extern "C" void hipemu_set_reverse_order(int on) { hipemu::set_reverse_order(on != 0); }
extern "C" void hipemu_set_strict_barrier(int on) { hipemu::set_strict_barrier(on != 0); }
