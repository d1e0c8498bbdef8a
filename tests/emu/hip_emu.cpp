// Coroutine scheduler behind tests/emu/hip_emu.hpp.  TEST INFRASTRUCTURE ONLY.
#include "hip_emu.hpp"

#include <stdio.h>
#include <sys/mman.h>

#include <atomic>
#include <thread>
#include <vector>

extern "C" void emu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
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
.size emu_switch,.-emu_switch
)");

namespace emu {
namespace {
constexpr int kMaxThreads = 1024;
constexpr size_t kStack = 256 * 1024;
constexpr int kSlots = 2;
constexpr int kMaxXchg = kLaneStride;

struct Wave {
    int live = 0, arrived = 0;
    unsigned gen = 0;
    alignas(16) unsigned char buf[kSlots][64][kMaxXchg];
};

struct Co {
    void* sp = nullptr;
    bool done = true;
    ThreadInfo info;
    unsigned xseq = 0;
};

struct Worker {
    std::vector<Co> co;
    std::vector<Wave> waves;
    char* stacks = nullptr;
    void* main_sp = nullptr;
    int nthreads = 0, live = 0, arrived = 0, current = -1;
    unsigned gen = 0;
    const std::function<void()>* body = nullptr;
};

thread_local Worker* tw = nullptr;

void finish_current();

void trampoline() {
    Worker* w = tw;
    (*w->body)();
    finish_current();
    abort();  // never reached
}

int next_live(Worker* w, int from) {
    for (int i = 1; i <= w->nthreads; ++i) {
        int j = (from + i) % w->nthreads;
        if (!w->co[j].done) return j;
    }
    return -1;
}

void yield() {
    Worker* w = tw;
    int me = w->current;
    int nx = next_live(w, me);
    if (nx < 0 || nx == me) return;
    w->current = nx;
    emu_switch(&w->co[me].sp, w->co[nx].sp);
}

void release_checks(Worker* w) {
    if (w->live > 0 && w->arrived >= w->live) {
        w->arrived = 0;
        w->gen++;
    }
    for (auto& wv : w->waves)
        if (wv.live > 0 && wv.arrived >= wv.live) {
            wv.arrived = 0;
            wv.gen++;
        }
}

void finish_current() {
    Worker* w = tw;
    int me = w->current;
    Co& c = w->co[me];
    c.done = true;
    w->live--;
    w->waves[c.info.wave].live--;
    release_checks(w);
    int nx = next_live(w, me);
    void* dummy;
    if (nx < 0) {
        w->current = -1;
        emu_switch(&dummy, w->main_sp);
    } else {
        w->current = nx;
        emu_switch(&dummy, w->co[nx].sp);
    }
}

void run_block(Worker* w, dim3 grid, dim3 block, unsigned bx, unsigned by, unsigned bz) {
    const int nt = (int)(block.x * block.y * block.z);
    w->nthreads = nt;
    w->live = nt;
    w->arrived = 0;
    w->gen = 0;
    const int nw = (nt + 63) / 64;
    w->waves.assign(nw, Wave());
    if ((int)w->co.size() < nt) w->co.resize(nt);
    for (int t = 0; t < nt; ++t) {
        Co& c = w->co[t];
        c.done = false;
        c.xseq = 0;
        c.info.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
        c.info.bid = dim3(bx, by, bz);
        c.info.bdim = block;
        c.info.gdim = grid;
        c.info.lane = t & 63;
        c.info.wave = t >> 6;
        w->waves[t >> 6].live++;
        uintptr_t top = (uintptr_t)(w->stacks + (size_t)(t + 1) * kStack);
        top &= ~(uintptr_t)15;
        void** sp = (void**)top;
        *--sp = nullptr;                 // fake return address of trampoline
        *--sp = (void*)&trampoline;      // popped by `ret`
        for (int i = 0; i < 6; ++i) *--sp = nullptr;
        c.sp = sp;
    }
    w->current = 0;
    emu_switch(&w->main_sp, w->co[0].sp);
}
}  // namespace

ThreadInfo* cur() { return &tw->co[tw->current].info; }

void syncthreads() {
    Worker* w = tw;
    unsigned g = w->gen;
    w->arrived++;
    if (w->arrived >= w->live) {
        w->arrived = 0;
        w->gen++;
        return;
    }
    while (w->gen == g) yield();
}

static void wave_sync() {
    Worker* w = tw;
    Wave& wv = w->waves[w->co[w->current].info.wave];
    unsigned g = wv.gen;
    wv.arrived++;
    if (wv.arrived >= wv.live) {
        wv.arrived = 0;
        wv.gen++;
        return;
    }
    while (wv.gen == g) yield();
}

const unsigned char* exchange(const void* mine, int bytes) {
    if (bytes > kMaxXchg) abort();
    Worker* w = tw;
    Co& c = w->co[w->current];
    Wave& wv = w->waves[c.info.wave];
    int slot = (int)(c.xseq++ % kSlots);
    memcpy(wv.buf[slot][c.info.lane], mine, bytes);
    wave_sync();
    return &wv.buf[slot][0][0];
}

float shfl_xor_f(float v, int mask) {
    const unsigned char* all = exchange(&v, 4);
    int src = (cur()->lane ^ mask) & 63;
    float r;
    memcpy(&r, all + (size_t)src * kMaxXchg, 4);
    return r;
}
int shfl_xor_i(int v, int mask) {
    const unsigned char* all = exchange(&v, 4);
    int src = (cur()->lane ^ mask) & 63;
    int r;
    memcpy(&r, all + (size_t)src * kMaxXchg, 4);
    return r;
}

void launch(dim3 grid, dim3 block, std::function<void()> body) {
    const uint64_t nblocks = (uint64_t)grid.x * grid.y * grid.z;
    const int nt = (int)(block.x * block.y * block.z);
    if (nblocks == 0 || nt == 0) return;
    if (nt > kMaxThreads) abort();
    unsigned hw = std::thread::hardware_concurrency();
    int nworkers = (int)std::min<uint64_t>(nblocks, hw ? hw : 4);
    std::atomic<uint64_t> next{0};
    auto work = [&]() {
        Worker w;
        w.stacks = (char*)mmap(nullptr, (size_t)nt * kStack, PROT_READ | PROT_WRITE,
                               MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (w.stacks == (char*)MAP_FAILED) abort();
        w.body = &body;
        tw = &w;
        for (;;) {
            uint64_t b = next.fetch_add(1);
            if (b >= nblocks) break;
            unsigned bx = (unsigned)(b % grid.x);
            unsigned by = (unsigned)((b / grid.x) % grid.y);
            unsigned bz = (unsigned)(b / ((uint64_t)grid.x * grid.y));
            run_block(&w, grid, block, bx, by, bz);
        }
        munmap(w.stacks, (size_t)nt * kStack);
        tw = nullptr;
    };
    if (nworkers <= 1) {
        std::thread t(work);  // own thread: keeps thread_local __shared__ isolated
        t.join();
    } else {
        std::vector<std::thread> ts;
        for (int i = 0; i < nworkers; ++i) ts.emplace_back(work);
        for (auto& t : ts) t.join();
    }
}
}  // namespace emu

// fp16 <-> fp32 (software) for the emulated f16_t
float emu_half_to_float(uint16_t h) {
    uint32_t s = (h >> 15) & 1, e = (h >> 10) & 31, m = h & 1023, u;
    if (e == 0) {
        if (m == 0) u = s << 31;
        else {
            int ex = -1;
            do { ex++; m <<= 1; } while (!(m & 1024));
            u = (s << 31) | ((uint32_t)(127 - 15 - ex) << 23) | ((m & 1023) << 13);
        }
    } else if (e == 31) u = (s << 31) | 0x7f800000u | (m << 13);
    else u = (s << 31) | ((e - 15 + 127) << 23) | (m << 13);
    float f;
    memcpy(&f, &u, 4);
    return f;
}
uint16_t emu_float_to_half(float f) {
    _Float16 h = (_Float16)f;
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}
