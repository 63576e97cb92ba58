// Fiber scheduler behind tests/emu/hip_emu.h (TEST INFRASTRUCTURE ONLY; see that header).
#include "hip_emu.h"

#include <sys/mman.h>
#include <vector>

emu_uint3 threadIdx, blockIdx;
dim3 blockDim, gridDim;
extern "C" {
alignas(256) char i2i_smem[160 * 1024];
}

// Minimal x86-64 context switch (callee-saved registers + stack pointer).
extern "C" void emu_switch(void** save_sp, void* new_sp);
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
)");

namespace emu {
namespace {
constexpr size_t kStack = 256 * 1024;
constexpr int kMaxThreads = 1024;

struct Fiber {
    void* sp = nullptr;
    bool done = false;
    // what it waits for: 0 nothing, 1 wave collective (gen), 2 block barrier (gen)
    int wait_kind = 0;
    unsigned long wait_gen = 0;
    emu_uint3 tid;
};
struct Wave {
    unsigned long gen = 0;
    int arrived = 0;
    int live = 0;
    alignas(16) char in[64 * kMaxIn];
    alignas(16) char out[64 * kMaxOut];
};

char* g_stacks = nullptr;
Fiber g_fib[kMaxThreads];
Wave g_wave[kMaxThreads / 64];
unsigned long g_block_gen = 0;
int g_block_arrived = 0, g_block_live = 0;
int g_cur = -1, g_nthreads = 0;
void* g_sched_sp = nullptr;
const std::function<void()>* g_body = nullptr;

void yield_to_sched() { emu_switch(&g_fib[g_cur].sp, g_sched_sp); }
void retire_all_of_current();

void fiber_main() {
    (*g_body)();
    retire_all_of_current();                 // s_endpgm: everything still in flight lands
    Fiber& f = g_fib[g_cur];
    f.done = true;
    Wave& w = g_wave[g_cur / 64];
    w.live--;
    g_block_live--;
    // a lane that exits may complete a pending rendezvous of the others
    if (w.live > 0 && w.arrived == w.live) {
        fprintf(stderr, "emu: lane exited while its wave waits in a collective (divergent collective)\n");
        abort();
    }
    if (g_block_live > 0 && g_block_arrived == g_block_live) { g_block_arrived = 0; g_block_gen++; }
    yield_to_sched();
    abort();  // never resumed
}

void trampoline() { fiber_main(); }

void init_fiber(int t) {
    char* top = g_stacks + (size_t)(t + 1) * kStack;
    // layout expected by emu_switch's epilogue: 6 popped registers then the return address.
    // After `ret`, rsp must be == 8 (mod 16) as at a normal function entry.
    uintptr_t sp = ((uintptr_t)top & ~(uintptr_t)15) - 8;
    void** s = (void**)sp;
    *--s = (void*)&trampoline;          // return address
    for (int i = 0; i < 6; ++i) *--s = nullptr;
    g_fib[t].sp = (void*)s;
    g_fib[t].done = false;
    g_fib[t].wait_kind = 0;
}

bool runnable(const Fiber& f, int t) {
    if (f.done) return false;
    if (f.wait_kind == 1) return g_wave[t / 64].gen != f.wait_gen;
    if (f.wait_kind == 2) return g_block_gen != f.wait_gen;
    return true;
}
}  // namespace

int lane_id() { return g_cur & 63; }

// ---- asynchronous-memory model: one FIFO of pending copies per lane (see hip_emu.h)
namespace {
struct Pending { const void* src; void* dst; int bytes; };
std::vector<Pending> g_pend[kMaxThreads];
size_t g_pend_head[kMaxThreads];
bool g_async = false;
void pend_retire(int t, size_t keep) {
    std::vector<Pending>& q = g_pend[t];
    while (q.size() - g_pend_head[t] > keep) {
        const Pending& e = q[g_pend_head[t]++];
        if (e.bytes > 0) memcpy(e.dst, e.src, (size_t)e.bytes);
    }
    if (g_pend_head[t] == q.size()) { q.clear(); g_pend_head[t] = 0; }
}
}  // namespace
bool vmem_async() { return g_async; }
void vmem_defer(const void* src, void* dst, int bytes) { g_pend[g_cur].push_back({src, dst, bytes}); }
void vmem_note(int n) { if (g_async) for (int i = 0; i < n; ++i) g_pend[g_cur].push_back({nullptr, nullptr, 0}); }
int g_drop_barrier = 0;      // I2I_EMU_DROP_BARRIER=n: every thread skips its n-th barrier (self-test: the schedules must notice)
int g_barrier_calls[kMaxThreads];
int g_order = 0;
bool g_skew = false;         // I2I_EMU_ORDER set: run-ahead scheduling (see launch)
unsigned long long g_rng = 1;
int g_wait_bias = 0;     // I2I_EMU_WAIT_BIAS: added to every count (self-test of the model: > 0 must break the DMA kernels)
void vmem_wait(int n) { if (g_async) pend_retire(g_cur, (size_t)(n + g_wait_bias < 0 ? 0 : n + g_wait_bias)); }
namespace { void retire_all_of_current() { if (g_async) pend_retire(g_cur, 0); } }

void wave_collective(const void* in, size_t in_bytes, void* out, size_t out_bytes,
                     void (*fn)(const char*, char*, void*), void* ctx) {
    if (in_bytes > (size_t)kMaxIn || out_bytes > (size_t)kMaxOut) { fprintf(stderr, "emu: collective payload too large\n"); abort(); }
    const int me = g_cur, lane = me & 63;
    Wave& w = g_wave[me / 64];
    memcpy(w.in + lane * kMaxIn, in, in_bytes);
    w.arrived++;
    if (w.arrived == w.live) {
        if (w.live != 64 && w.live != (g_nthreads - (me / 64) * 64 < 64 ? g_nthreads - (me / 64) * 64 : 64)) {
            fprintf(stderr, "emu: collective with exited lanes\n"); abort();
        }
        fn(w.in, w.out, ctx);
        w.arrived = 0;
        w.gen++;
    } else {
        g_fib[me].wait_kind = 1;
        g_fib[me].wait_gen = w.gen;
        yield_to_sched();
        g_fib[me].wait_kind = 0;
    }
    memcpy(out, w.out + lane * kMaxOut, out_bytes);
}

void block_barrier() {
    const int me = g_cur;
    if (g_drop_barrier > 0 && ++g_barrier_calls[me] == g_drop_barrier) return;     // self-test knob: this barrier is "forgotten"
    g_block_arrived++;
    if (g_block_arrived == g_block_live) {
        g_block_arrived = 0;
        g_block_gen++;
        return;
    }
    g_fib[me].wait_kind = 2;
    g_fib[me].wait_gen = g_block_gen;
    yield_to_sched();
    g_fib[me].wait_kind = 0;
}

void launch(const std::function<void()>& body, dim3 grid, dim3 block, size_t smem) {
    const int nt = (int)(block.x * block.y * block.z);
    if (nt > kMaxThreads || smem > sizeof(i2i_smem)) { fprintf(stderr, "emu: launch too large (%d threads, %zu B LDS)\n", nt, smem); abort(); }
    if (!g_stacks) {
        g_stacks = (char*)mmap(nullptr, kStack * kMaxThreads, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (g_stacks == (char*)MAP_FAILED) { perror("mmap"); abort(); }
    }
    g_body = &body;
    g_nthreads = nt;
    {
        const char* e = getenv("I2I_EMU_ASYNC");
        g_async = e && atoi(e) != 0;
        const char* b = getenv("I2I_EMU_WAIT_BIAS");
        g_wait_bias = b ? atoi(b) : 0;
        const char* d = getenv("I2I_EMU_DROP_BARRIER");
        g_drop_barrier = d ? atoi(d) : 0;
        const char* o = getenv("I2I_EMU_ORDER");
        g_order = o ? atoi(o) : 0;
        g_skew = o != nullptr;
        g_rng = 0x9E3779B97F4A7C15ull ^ (unsigned long long)g_order;
        for (int t = 0; t < nt; ++t) { g_pend[t].clear(); g_pend_head[t] = 0; }
    }
    blockDim = block;
    gridDim = grid;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                for (int t = 0; t < nt; ++t) {
                    init_fiber(t);
                    g_fib[t].tid = {(unsigned)(t % block.x), (unsigned)((t / block.x) % block.y), (unsigned)(t / (block.x * block.y))};
                }
                for (int w = 0; w < (nt + 63) / 64; ++w) {
                    g_wave[w].gen = 0; g_wave[w].arrived = 0;
                    g_wave[w].live = (nt - w * 64) < 64 ? (nt - w * 64) : 64;
                }
                g_block_gen = 0; g_block_arrived = 0; g_block_live = nt;
                for (int t = 0; t < nt; ++t) g_barrier_calls[t] = 0;
                // poison LDS so reads of unwritten bytes show up as NaNs / garbage
                memset(i2i_smem, 0xFF, smem ? smem : 1);
                int remaining = nt;
                const int nwaves = (nt + 63) / 64;
                while (remaining > 0) {
                    bool progress = false;
                    // wave order of this sweep (I2I_EMU_ORDER): 0 ascending, 1 descending, n > 1 pseudo-random (seed n).  A fiber
                    // runs until it blocks, so the order decides which wave races ahead between two barriers: a missing
                    // barrier that the ascending order happens to hide shows up under another one.
                    int order[kMaxThreads / 64];
                    for (int w = 0; w < nwaves; ++w) order[w] = w;
                    if (g_order == 1) for (int w = 0; w < nwaves; ++w) order[w] = nwaves - 1 - w;
                    else if (g_order > 1)
                        for (int w = nwaves - 1; w > 0; --w) {
                            g_rng = g_rng * 6364136223846793005ull + 1442695040888963407ull;
                            const int k = (int)((g_rng >> 33) % (unsigned)(w + 1));
                            const int tmp = order[w]; order[w] = order[k]; order[k] = tmp;
                        }
                    // Each wave runs until it can go no further (all lanes blocked at a BLOCK barrier, or done) before the next one
                    // starts: wave-level collectives (MFMA, shuffles) complete inside the inner loop, so the leading wave is a whole
                    // barrier interval ahead of the others -- the widest skew real hardware can show between two barriers.  With
                    // I2I_EMU_ORDER unset the classic lock-step sweep (one pass over all waves) is kept.
                    for (int wi = 0; wi < nwaves; ++wi) {
                        bool wave_progress = true;
                        while (wave_progress) {
                            wave_progress = false;
                            for (int l = 0; l < 64; ++l) {
                                const int t = order[wi] * 64 + l;
                                if (t >= nt) continue;
                                Fiber& f = g_fib[t];
                                if (!runnable(f, t)) continue;
                                g_cur = t;
                                threadIdx = f.tid;
                                blockIdx = {bx, by, bz};
                                emu_switch(&g_sched_sp, f.sp);
                                progress = true;
                                wave_progress = true;
                                if (f.done) remaining--;
                            }
                            if (!g_skew) break;            // lock-step mode: one pass per wave and sweep
                        }
                    }
                    if (!progress) { fprintf(stderr, "emu: deadlock in block (%u,%u,%u)\n", bx, by, bz); abort(); }
                }
            }
    g_cur = -1;
}
}  // namespace emu
