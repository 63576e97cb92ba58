// Wave-level CPU emulator for the HIP kernel sources (TEST INFRASTRUCTURE ONLY).
//
// The product kernels under img2img-turbo_amd/csrc/*.hip are written for gfx950 only and contain no
// host/device dual paths.  There is no GPU in the build container and GPU minutes are scarce, so the
// tests compile the *same* sources a second time with the host clang and this header force-included
// (`-include hip_emu.h`, with tests/emu/ first on the include path so <hip/hip_runtime.h> resolves to
// a stub).  Every thread of a workgroup becomes a fiber; __syncthreads(), cross-lane shuffles and the
// MFMA builtins are wave/block rendezvous points that reproduce the gfx950 lane->element layouts
// (cdna_hip_programming.md section 3).  This checks indexing, tiling, LDS layout and epilogue logic
// bit-for-bit against the oracle on the CPU.  It says nothing about performance and is never loaded
// by the product (img2img-turbo_amd/_capi.py loads only the hipcc-built library).
//
// Synchronisation is checked on adversarial schedules selected by environment variables (read per launch):
//   I2I_EMU_ASYNC=1          LDS-DMA copies / hidden loads land only at the s_waitcnt vmcnt(N) that retires them (latest
//                            completion the in-order counter allows; the default is the earliest: at issue)
//   I2I_EMU_ORDER=0|1|n      run-ahead wave scheduling: a wave runs until it blocks at a block barrier before the next one
//                            starts (ascending / descending / shuffled with seed n); unset = lock-step sweep
//   I2I_EMU_WAIT_BIAS=k      self-test: every vmcnt wait k operations too generous  (the kernels must then fail)
//   I2I_EMU_DROP_BARRIER=n   self-test: every thread skips its n-th barrier          (the kernels must then fail)
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define I2I_EMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__
#ifndef __restrict__
#define __restrict__
#endif

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct emu_uint3 { unsigned x, y, z; };
extern emu_uint3 threadIdx, blockIdx;
extern dim3 blockDim, gridDim;
extern "C" char i2i_smem[];   // the one dynamic-LDS array every kernel carves from

typedef void* hipStream_t;
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;
typedef void* hipEvent_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorUnknown = 999 };
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }

namespace emu {
void launch(const std::function<void()>& body, dim3 grid, dim3 block, size_t smem);
// Wave rendezvous: every live lane deposits `in`; the last arriver runs fn(all_in, all_out) once.
// in/out slots are kMaxIn / kMaxOut bytes per lane.
constexpr int kMaxIn = 160, kMaxOut = 96;
void wave_collective(const void* in, size_t in_bytes, void* out, size_t out_bytes,
                     void (*fn)(const char* in_slots, char* out_slots, void* ctx), void* ctx);
void block_barrier();
int lane_id();
// Asynchronous-memory model (I2I_EMU_ASYNC=1): LDS-DMA copies and the loads the kernels hide from the compiler are QUEUED per
// lane and only performed by the s_waitcnt vmcnt(N) that retires them (oldest first, until N remain) -- the latest completion
// the in-order vmcnt counter allows.  The default (everything completes at issue) is the earliest one.  A kernel that reads
// DMA data before the wait that covers it, or waits with too large a count, passes the default run and fails this one.
bool vmem_async();
void vmem_defer(const void* src, void* dst, int bytes);   // a queued 16-byte (or smaller) copy
void vmem_note(int n);                                     // n compiler-visible VMEM ops the kernel counts in its waits (no data)
void vmem_wait(int n);                                     // s_waitcnt vmcnt(n)
}  // namespace emu

namespace emu {
template <class K, class... Args>
static inline void launch_kernel(K kern, dim3 grid, dim3 block, size_t smem, Args... args) {
    launch([&]() { kern(args...); }, grid, block, smem);
}
}  // namespace emu
#define hipLaunchKernelGGL(kern, grid, block, smem, stream, ...) \
    emu::launch_kernel(kern, (grid), (block), (smem), __VA_ARGS__)

static inline void __syncthreads() { emu::block_barrier(); }

// ---------------------------------------------------------------- cross-lane
template <class T>
static inline T emu_shfl_generic(T v, int arg, int mode) {
    static_assert(sizeof(T) <= 8, "shfl payload");
    struct In { unsigned long long bits; int arg; int mode; } in;
    in.bits = 0; memcpy(&in.bits, &v, sizeof(T)); in.arg = arg; in.mode = mode;
    unsigned long long out = 0;
    emu::wave_collective(&in, sizeof(in), &out, sizeof(out),
        [](const char* is, char* os, void*) {
            for (int l = 0; l < 64; ++l) {
                const In* me = (const In*)(is + l * emu::kMaxIn);
                int src = l;
                if (me->mode == 0) src = l ^ me->arg;          // xor
                else if (me->mode == 1) src = me->arg & 63;    // idx
                else if (me->mode == 2) src = (l + me->arg < 64) ? l + me->arg : l;  // down
                const In* s = (const In*)(is + src * emu::kMaxIn);
                memcpy(os + l * emu::kMaxOut, &s->bits, 8);
            }
        }, nullptr);
    T r; memcpy(&r, &out, sizeof(T)); return r;
}
template <class T> static inline T __shfl_xor(T v, int mask, int = 64) { return emu_shfl_generic(v, mask, 0); }
template <class T> static inline T __shfl(T v, int lane, int = 64) { return emu_shfl_generic(v, lane, 1); }
template <class T> static inline T __shfl_down(T v, int d, int = 64) { return emu_shfl_generic(v, d, 2); }
static inline int __builtin_amdgcn_readfirstlane(int v) { return emu_shfl_generic(v, 0, 1); }
static inline void __builtin_amdgcn_s_barrier() { emu::block_barrier(); }
namespace emu {
// v_permlane32_swap_b32 a, b: lanes 32-63 of a exchange with lanes 0-31 of b.
static inline void permlane32_swap(float& a, float& b) {
    struct In { float a, b; } in = {a, b};
    float out[2];
    wave_collective(&in, sizeof(in), out, sizeof(out),
        [](const char* is, char* os, void*) {
            for (int l = 0; l < 64; ++l) {
                const In* me = (const In*)(is + l * kMaxIn);
                float* o = (float*)(os + l * kMaxOut);
                if (l < 32) { o[0] = me->a; o[1] = ((const In*)(is + (l + 32) * kMaxIn))->a; }
                else        { o[0] = ((const In*)(is + (l - 32) * kMaxIn))->b; o[1] = me->b; }
            }
        }, nullptr);
    a = out[0]; b = out[1];
}
// global_load_lds_dwordx4: LDS destination = wave-uniform base + lane * 16; executed synchronously here.
static inline void global_load_lds16(const void* g, void* lds_wave_base) {
    if (vmem_async()) vmem_defer(g, (char*)lds_wave_base + 16 * lane_id(), 16);
    else memcpy((char*)lds_wave_base + 16 * lane_id(), g, 16);
}
}  // namespace emu
static inline void __builtin_amdgcn_s_setprio(int) {}
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline void __builtin_amdgcn_sched_group_barrier(int, int, int) {}

// ---------------------------------------------------------------- MFMA (gfx950 layouts)
typedef __bf16 emu_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 emu_f16x8 __attribute__((ext_vector_type(8)));
typedef float emu_f32x4 __attribute__((ext_vector_type(4)));
typedef float emu_f32x16 __attribute__((ext_vector_type(16)));

// 16x16xK: A lane l holds A[i=l&15][k=(l>>4)*E+j]; B lane l holds B[k=(l>>4)*E+j][n=l&15];
// D lane l reg r = D[row=(l>>4)*4+r][col=l&15].   (E = 8 for 16-bit K=32, E = 1 for f32 K=4)
template <int E>
static inline emu_f32x4 emu_mfma16(const float* a, const float* b, emu_f32x4 c) {
    struct In { float a[8], b[8], c[4]; } in;
    for (int j = 0; j < E; ++j) { in.a[j] = a[j]; in.b[j] = b[j]; }
    for (int r = 0; r < 4; ++r) in.c[r] = c[r];
    float out[4];
    emu::wave_collective(&in, sizeof(in), out, sizeof(out),
        [](const char* is, char* os, void*) {
            for (int l = 0; l < 64; ++l) {
                float* o = (float*)(os + l * emu::kMaxOut);
                const In* me = (const In*)(is + l * emu::kMaxIn);
                int col = l & 15;
                for (int r = 0; r < 4; ++r) {
                    int row = (l >> 4) * 4 + r;
                    float acc = me->c[r];
                    for (int q = 0; q < 4; ++q) {   // k-ordered fmaf chain
                        const In* la = (const In*)(is + (q * 16 + row) * emu::kMaxIn);
                        const In* lb = (const In*)(is + (q * 16 + col) * emu::kMaxIn);
                        for (int j = 0; j < E; ++j) acc = fmaf(la->a[j], lb->b[j], acc);
                    }
                    o[r] = acc;
                }
            }
        }, nullptr);
    emu_f32x4 d = {out[0], out[1], out[2], out[3]};
    return d;
}
static inline emu_f32x4 __builtin_amdgcn_mfma_f32_16x16x32_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x4 c, int, int, int) {
    float fa[8], fb[8];
    for (int j = 0; j < 8; ++j) { fa[j] = (float)a[j]; fb[j] = (float)b[j]; }
    return emu_mfma16<8>(fa, fb, c);
}
static inline emu_f32x4 __builtin_amdgcn_mfma_f32_16x16x32_f16(emu_f16x8 a, emu_f16x8 b, emu_f32x4 c, int, int, int) {
    float fa[8], fb[8];
    for (int j = 0; j < 8; ++j) { fa[j] = (float)a[j]; fb[j] = (float)b[j]; }
    return emu_mfma16<8>(fa, fb, c);
}
static inline emu_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, emu_f32x4 c, int, int, int) {
    return emu_mfma16<1>(&a, &b, c);
}

// 32x32xK: A lane l holds A[i=l&31][k=(l>>5)*E+j]; B[k=(l>>5)*E+j][n=l&31];
// D reg r (0..15): row=(r&3)+8*(r>>2)+4*(l>>5), col=l&31.
template <int E>
static inline emu_f32x16 emu_mfma32(const float* a, const float* b, emu_f32x16 c) {
    struct In { float a[8], b[8], c[16]; } in;
    for (int j = 0; j < E; ++j) { in.a[j] = a[j]; in.b[j] = b[j]; }
    for (int r = 0; r < 16; ++r) in.c[r] = c[r];
    float out[16];
    emu::wave_collective(&in, sizeof(in), out, sizeof(out),
        [](const char* is, char* os, void*) {
            for (int l = 0; l < 64; ++l) {
                float* o = (float*)(os + l * emu::kMaxOut);
                const In* me = (const In*)(is + l * emu::kMaxIn);
                int col = l & 31;
                for (int r = 0; r < 16; ++r) {
                    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                    float acc = me->c[r];
                    for (int q = 0; q < 2; ++q) {
                        const In* la = (const In*)(is + (q * 32 + row) * emu::kMaxIn);
                        const In* lb = (const In*)(is + (q * 32 + col) * emu::kMaxIn);
                        for (int j = 0; j < E; ++j) acc = fmaf(la->a[j], lb->b[j], acc);
                    }
                    o[r] = acc;
                }
            }
        }, nullptr);
    emu_f32x16 d;
    for (int r = 0; r < 16; ++r) d[r] = out[r];
    return d;
}
static inline emu_f32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x16 c, int, int, int) {
    float fa[8], fb[8];
    for (int j = 0; j < 8; ++j) { fa[j] = (float)a[j]; fb[j] = (float)b[j]; }
    return emu_mfma32<8>(fa, fb, c);
}
static inline emu_f32x16 __builtin_amdgcn_mfma_f32_32x32x16_f16(emu_f16x8 a, emu_f16x8 b, emu_f32x16 c, int, int, int) {
    float fa[8], fb[8];
    for (int j = 0; j < 8; ++j) { fa[j] = (float)a[j]; fb[j] = (float)b[j]; }
    return emu_mfma32<8>(fa, fb, c);
}
static inline emu_f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, emu_f32x16 c, int, int, int) {
    return emu_mfma32<1>(&a, &b, c);
}

// ---------------------------------------------------------------- math spelled the device way
#define __expf(x) expf(x)
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline float __fdividef(float a, float b) { return a / b; }

// ---------------------------------------------------------------- runtime stubs used by capi
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
enum hipMemcpyKind { hipMemcpyDefault = 4, hipMemcpyDeviceToDevice = 3 };
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
