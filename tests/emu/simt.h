// simt.h -- a tiny host SIMT emulator (TEST INFRASTRUCTURE ONLY).
//
// Runs a wave-cooperative kernel body -- the same source hipcc compiles for gfx950 -- on the CPU:
// the 64 lanes of a wavefront are 64 coroutines that the scheduler resumes round-robin; a wave
// collective (ballot / shuffle / barrier) parks the calling lane until every live lane of the wave
// has arrived, so collectives see exactly the values a real wavefront would exchange.  Blocks are
// one wavefront wide and are run one after another, in an order the caller picks (ascending,
// descending or shuffled) to exercise the encoder's speculative, order-independent sweeps.
#pragma once
#if !defined(__x86_64__)
#error "the SIMT emulator's context switch is written for x86-64"
#endif
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#if defined(__SANITIZE_THREAD__)  // the race check (tests/race): the lanes' coroutines are fibers to ThreadSanitizer
#include <sanitizer/tsan_interface.h>
#define ORZ_SIMT_TSAN 1
#endif

extern "C" void orz_simt_switch(void** save_sp, void* new_sp);
asm(R"(
.text
.globl orz_simt_switch
.type orz_simt_switch,@function
orz_simt_switch:
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
.size orz_simt_switch,.-orz_simt_switch
)");

namespace simt {

constexpr int kLanes = 64;
constexpr size_t kStack = 256 * 1024;

struct Wave;
struct Lane {
    void* fiber = nullptr;
    void* sp = nullptr;
    bool done = true;
    uint32_t gen = 0;  // collectives this lane has entered
};

struct Wave {
    Lane lanes[kLanes];
    void* sched_sp = nullptr;
    int cur = -1;
    // collective deposit slots, double-buffered by generation parity, tagged with the generation
    uint64_t val[2][kLanes];
    uint32_t tag[2][kLanes];
    uint8_t* stacks = nullptr;
    void (*entry)(void*) = nullptr;
    void* entry_arg = nullptr;
    std::vector<uint8_t> lds;
    uint32_t block = 0;
    void* sched_fiber = nullptr;

    Wave() {
        stacks = (uint8_t*)std::malloc(kStack * kLanes);
        std::memset(tag, 0xff, sizeof tag);
#if defined(ORZ_SIMT_TSAN)
        for (int i = 0; i < kLanes; i++) lanes[i].fiber = __tsan_create_fiber(0);
#endif
    }
    ~Wave() { std::free(stacks); }
};

inline Wave*& current_wave() {
    static thread_local Wave* w = nullptr;
    return w;
}

inline void lane_trampoline() {
    Wave* w = current_wave();
    int me = w->cur;
    w->entry(w->entry_arg);
    w->lanes[me].done = true;
    void* dummy;
#if defined(ORZ_SIMT_TSAN)
    __tsan_switch_to_fiber(w->sched_fiber, 0);
#endif
    orz_simt_switch(&dummy, w->sched_sp);  // never returns
    std::abort();
}

// context handed to kernel bodies on the host
struct WaveCtx {
    Wave* w;
    uint32_t lane_;
    uint32_t lane() const { return lane_; }
    uint32_t block() const { return w->block; }
    uint8_t* lds() const { return w->lds.data(); }

    void park() {
#if defined(ORZ_SIMT_TSAN)
        __tsan_switch_to_fiber(w->sched_fiber, 0);
#endif
        orz_simt_switch(&w->lanes[lane_].sp, w->sched_sp);
    }
    uint32_t deposit(uint64_t v) {
        Lane& l = w->lanes[lane_];
        uint32_t g = l.gen++;
        w->val[g & 1][lane_] = v;
        w->tag[g & 1][lane_] = g;
        park();
        return g;
    }
    uint64_t ballot(bool p) {
        uint32_t g = deposit(p ? 1 : 0);
        uint64_t m = 0;
        for (int i = 0; i < kLanes; i++)
            if (w->tag[g & 1][i] == g && w->val[g & 1][i]) m |= 1ull << i;
        return m;
    }
    uint32_t bcast(uint32_t v, uint32_t src) {
        uint32_t g = deposit(v);
        return w->tag[g & 1][src] == g ? (uint32_t)w->val[g & 1][src] : 0;
    }
    uint32_t shfl(uint32_t v, uint32_t src) { return bcast(v, src); }
    uint64_t bcast64(uint64_t v, uint32_t src) {
        uint32_t g = deposit(v);
        return w->tag[g & 1][src] == g ? w->val[g & 1][src] : 0;
    }
    void sync() { deposit(0); }
    unsigned long long clock() const { return 0; }
    unsigned long long wallclock() const { return 0; }
};

enum Order { kAscending = 0, kDescending = 1, kShuffled = 2 };

// Run `nblocks` one-wave blocks of kernel body `k` (k(WaveCtx&) is the per-lane code).
// (`first`, `stride`: the blocks first, first + stride, ... of the order -- the race check runs a launch's blocks on several host threads)
template <class K>
void launch_waves(size_t nblocks, const K& k, size_t lds_bytes, Order order = kAscending, uint64_t seed = 1, size_t first = 0, size_t stride = 1) {
    static thread_local Wave* wave = nullptr;
    if (!wave) wave = new Wave();
    Wave* w = wave;
    current_wave() = w;
#if defined(ORZ_SIMT_TSAN)
    w->sched_fiber = __tsan_get_current_fiber();
#endif
    if (w->lds.size() < lds_bytes) w->lds.resize(lds_bytes);
    struct Arg {
        const K* k;
        Wave* w;
    } arg{&k, w};
    w->entry_arg = &arg;
    w->entry = [](void* p) {
        Arg* a = (Arg*)p;
        WaveCtx ctx{a->w, (uint32_t)a->w->cur};
        (*a->k)(ctx);
    };
    std::vector<uint32_t> ord(nblocks);
    for (size_t i = 0; i < nblocks; i++) ord[i] = (uint32_t)i;
    if (order == kDescending) {
        for (size_t i = 0; i < nblocks; i++) ord[i] = (uint32_t)(nblocks - 1 - i);
    } else if (order == kShuffled) {
        uint64_t s = seed * 0x9E3779B97F4A7C15ull + 12345;
        for (size_t i = nblocks; i > 1; i--) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            size_t j = (size_t)(s % i);
            uint32_t t = ord[i - 1]; ord[i - 1] = ord[j]; ord[j] = t;
        }
    }
    for (size_t bi = first; bi < nblocks; bi += stride) {
        w->block = ord[bi];
        std::memset(w->tag, 0xff, sizeof w->tag);
        for (int i = 0; i < kLanes; i++) {
            Lane& l = w->lanes[i];
            l.done = false;
            l.gen = 0;
            // fresh stack: six callee-saved slots, then the return address of the switch
            uint8_t* top = w->stacks + kStack * (size_t)(i + 1);
            uintptr_t sp = ((uintptr_t)top & ~(uintptr_t)15) - 8;  // as if `call`ed: rsp % 16 == 8 at entry
            sp -= 8;
            *(void**)sp = (void*)&lane_trampoline;
            sp -= 6 * 8;
            std::memset((void*)sp, 0, 6 * 8);
            l.sp = (void*)sp;
        }
        bool live = true;
        while (live) {
            live = false;
            for (int i = 0; i < kLanes; i++) {
                Lane& l = w->lanes[i];
                if (l.done) continue;
                w->cur = i;
#if defined(ORZ_SIMT_TSAN)
                __tsan_switch_to_fiber(l.fiber, 0);
#endif
                orz_simt_switch(&w->sched_sp, l.sp);
                if (!l.done) live = true;
            }
            // every lane still alive must be parked in the SAME collective: wave collectives under
            // divergent control flow would deadlock or mis-pair on real hardware
            uint32_t g = 0;
            bool have = false;
            for (int i = 0; i < kLanes; i++) {
                if (w->lanes[i].done) continue;
                if (!have) { g = w->lanes[i].gen; have = true; }
                else if (w->lanes[i].gen != g) {
                    std::fprintf(stderr, "simt: divergent wave collective in block %u (lane %d at %u, expected %u)\n",
                                 w->block, i, w->lanes[i].gen, g);
                    std::abort();
                }
            }
        }
    }
}

}  // namespace simt
