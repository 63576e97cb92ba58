// Device-side vocabulary shared by the gfx950 kernels: element types, 16-byte chunks, MFMA wrappers.
//
// Every contraction operand moves as 16-byte "chunks" (8 x bf16/f16 or 4 x f32): one global_load_dwordx4
// per lane, one ds_write_b128 / ds_read_b128 per chunk, and one chunk per lane feeds
//   - 16-bit: ONE v_mfma_f32_16x16x32_{bf16,f16}   (lane l holds k = 8*(l>>4) .. +8)
//   - f32   : FOUR v_mfma_f32_16x16x4_f32, MFMA j consuming element j of the chunk.  Any k->lane
//             assignment is valid as long as A and B use the same one, so the f32 path keeps the
//             wide LDS reads of the 16-bit path (exact f32 = fmaf chain; the 1e-3 parity mode).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/i2i_turbo.h"

extern "C" __shared__ __attribute__((aligned(16))) char i2i_smem[];   // the only LDS object (G17)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int EPC = 4;            // elements per 16-byte chunk
    static constexpr int DT = I2I_F32;
    typedef f32x4 chunk_t;
};
template <> struct Elem<__bf16> {
    static constexpr int EPC = 8;
    static constexpr int DT = I2I_BF16;
    typedef bf16x8 chunk_t;
};
template <> struct Elem<_Float16> {
    static constexpr int EPC = 8;
    static constexpr int DT = I2I_F16;
    typedef f16x8 chunk_t;
};

template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return (T)v; }

template <typename T>
__device__ __forceinline__ typename Elem<T>::chunk_t zero_chunk() {
    typename Elem<T>::chunk_t z;
#pragma unroll
    for (int j = 0; j < Elem<T>::EPC; ++j) z[j] = (T)0.0f;
    return z;
}

// acc(16x16 f32 fragment) += A-chunk x B-chunk
__device__ __forceinline__ f32x4 mma_chunk(f32x4 a, f32x4 b, f32x4 acc) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
    return acc;
}
__device__ __forceinline__ f32x4 mma_chunk(bf16x8 a, bf16x8 b, f32x4 acc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma_chunk(f16x8 a, f16x8 b, f32x4 acc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
}

// x * sigmoid(x) with v_exp_f32 + v_rcp_f32 (1 ulp each; the IEEE division sequence costs ~10 more VALU per element)
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
#ifdef I2I_EMU
__device__ __forceinline__ float exp2_fast(float x) { return exp2f(x); }
__device__ __forceinline__ bool wave_any(bool c) {           // true in every lane iff c holds in some lane of the wave
    int v = c ? 1 : 0;
    for (int m = 1; m < 64; m <<= 1) v |= __shfl_xor(v, m);
    return v != 0;
}
#else
__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32 (2^x, ~1 ulp)
__device__ __forceinline__ bool wave_any(bool c) { return __builtin_amdgcn_ballot_w64(c) != 0; }
#endif
// Exact-form GELU (F.gelu default, erf based) with erf from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, i.e. fp32
// round-off class): one v_rcp_f32 + one v_exp_f32 + 8 FMAs instead of libm's erff (~40 VALU) -- the GEGLU epilogue
// of the ff.net.0 GEMMs evaluates it 4 times per accumulator fragment and was VALU bound.
__device__ __forceinline__ float erf_as_f(float x) {
    // explicit fused multiply-adds: the library is built with -ffp-contract=off (bit-stable reductions), which turned this
    // Horner chain into 10 multiplies + 7 adds per value -- 32 VALU per GEGLU output measured in the wide GEMM's epilogue,
    // where the gate activation is the bound of the K = 320 layers.  10 VALU + v_rcp + v_exp now.
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, ax, 1.0f));
    const float poly = t * __builtin_fmaf(t, __builtin_fmaf(t, __builtin_fmaf(t, __builtin_fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float r = __builtin_fmaf(-poly, exp2_fast(ax * (ax * -1.44269504088896341f)), 1.0f);
    return copysignf(r, x);
}
// gelu(x) = 0.5 x (1 + erf(x / sqrt 2)) with the 1/sqrt 2 folded into the constants of the erf above (p = 0.3275911 / sqrt 2,
// exponent -x^2 / 2): 12 VALU + v_rcp + v_exp
__device__ __forceinline__ float gelu_erf_f(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.23164189f, ax, 1.0f));
    const float poly = t * __builtin_fmaf(t, __builtin_fmaf(t, __builtin_fmaf(t, __builtin_fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float r = __builtin_fmaf(-poly, exp2_fast(ax * (ax * -0.72134752044448170f)), 1.0f);
    const float h = 0.5f * x;
    return __builtin_fmaf(h, copysignf(r, x), h);
}

// LDS tile geometry common to A and B tiles: rows of 128 bytes = 8 chunks; chunk kc of row r is stored
// at physical chunk kc ^ ((r >> 1) & 7).  With this XOR a ds_read_b128 wave access (lane -> row l&15,
// chunk 4*kg + (l>>4)) touches 16 distinct 16-byte slots in each of its four 16-lane service groups
// (cdna_hip_programming.md section 2 / T2), and the 8-lane ds_write_b128 groups stay conflict-free.
__device__ __forceinline__ int lds_chunk_off(int row, int kc) { return row * 128 + ((kc ^ ((row >> 1) & 7)) << 4); }

// Offset-robust variant: chunk kc of row r at physical chunk kc ^ (((r >> 1) & 3) << 1).  A fragment read of
// 16 CONSECUTIVE rows starting at ANY row (the shifted halo rows of the 3x3 taps) stays conflict-free in all
// four ds_read_b128 service groups (brute-forced over every start with tools/lds_bank_model.py; the 3-bit
// XOR above is 2-way conflicted for starts that are not multiples of 4).
__device__ __forceinline__ int lds_swz2(int row) { return ((row >> 1) & 3) << 1; }
__device__ __forceinline__ int lds_chunk_off2(int row, int kc) { return row * 128 + ((kc ^ lds_swz2(row)) << 4); }

// epilogue activation selector (i2i_igemm_params.act_out): 1 GELU (erf form), 2 quick_gelu
__device__ __forceinline__ float act_out_f(float x, int kind) {
    if (kind == 1) return gelu_erf_f(x);
    if (kind == 2) return x * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x));
    return x;
}

// Nearest-neighbour source index of upsampled coordinate i (F.interpolate(mode="nearest")): exact 2x is a shift; an
// explicit output size `up` != 2*in uses ATen's rule min(floor(i * (float)in/up), in-1).
__device__ __forceinline__ int up_src(int i, int in, int up, int ups) {
    if (up == (in << ups)) return i >> ups;
    const int j = (int)floorf((float)i * ((float)in / (float)up));
    return j < in - 1 ? j : in - 1;
}

// ---- explicit synchronisation for LDS-DMA pipelines (cdna_hip_programming.md section 5: raw s_barrier +
// counted waits; __syncthreads() would drain the DMA queue with vmcnt(0) at every barrier).
// The CPU emulator (tests/emu, I2I_EMU) executes copies synchronously, so the waits are no-ops there.
#ifdef I2I_EMU
__device__ __forceinline__ void note_vmem(int n) { emu::vmem_note(n); }    // n compiler-visible VMEM ops a counted wait relies on
__device__ __forceinline__ void wait_lgkm0() {}
template <int N> __device__ __forceinline__ void wait_vmcnt() { emu::vmem_wait(N); }      // retires queued copies in the async model
__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) { emu::global_load_lds16(g, lds_wave_base); }
__device__ __forceinline__ void glds16_sv(const char* sbase, unsigned voff, void* lds_wave_base) { emu::global_load_lds16(sbase + voff, lds_wave_base); }
#else
__device__ __forceinline__ void note_vmem(int) {}
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// global_load_lds_dwordx4: 16 bytes per lane, global address per lane, LDS destination = wave-uniform base
// + lane * 16 (M0); no VGPR round trip, no ds_write.
__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void glds16_sv(const char* sbase, unsigned voff, void* lds_wave_base) { glds16(sbase + voff, lds_wave_base); }
#endif
// A 16-byte global load hipcc does NOT count (cdna_hip_programming.md 5.7 form (ii)): beside an LDS-DMA pipeline
// the compiler drains vmcnt(0) before/after every ordinary VGPR load it can see, which would serialise the DMA
// ring.  The caller owns the completion: a counted wait_vmcnt<N>() followed by reg_fence() on every destination
// before its first use; between load and fence the destination must not be touched (keep the load unconditional,
// no phi).  sbase must be provably wave-uniform (SGPR pair), voff a 32-bit byte offset.
#ifdef I2I_EMU
template <typename C> __device__ __forceinline__ void gload16_uncounted(C& dst, const char* sbase, unsigned voff) {
    if (emu::vmem_async()) emu::vmem_defer(sbase + voff, &dst, (int)sizeof(C));      // lands at the wait that covers it
    else dst = *(const C*)(sbase + voff);
}
template <typename C> __device__ __forceinline__ void reg_fence(C&) {}
#else
template <typename C> __device__ __forceinline__ void gload16_uncounted(C& dst, const char* sbase, unsigned voff) {
    // s_nop 4: hipcc pads nothing inside an asm string, and the SGPR base / VGPR offset are usually written by the
    // SALU / VALU instructions just ahead (SALU-write -> VMEM-read of an SGPR needs 5 wait states)
    // readfirstlane makes the base provably uniform for the "s" constraint (free when it already lives in SGPRs)
    const uint64_t b = (uint64_t)(uintptr_t)sbase;
    const uint64_t ub = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) |
                        (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(ub) : "memory");
}
template <typename C> __device__ __forceinline__ void reg_fence(C& x) { asm volatile("" : "+v"(x)); }
#endif
// workgroup barrier that orders LDS traffic only (the caller places the vmcnt wait it needs)
__device__ __forceinline__ void lds_barrier() {
    wait_lgkm0();
    __builtin_amdgcn_s_barrier();
    wait_lgkm0();
}

// ---- 16-byte epilogue stores for 16-bit outputs (cdna_hip_programming.md T21).  With the MFMA operands swapped
// a lane of quad lq owns 4 consecutive channels of one pixel per 16-channel fragment (an 8-byte store).  The
// weight rows of every fragment are loaded PERMUTED (frag_row_perm: quads 1 and 2 exchanged) so that lane
// quads lq and lq+2 -- which v_permlane32_swap can exchange -- own ADJACENT channel quads; one half exchange
// per accumulator register over a fragment pair (j, j+1) then leaves 8 consecutive channels in every lane:
// lanes 0-31 finish fragment j, lanes 32-63 fragment j+1, channel base (lq & 1) * 8.
__device__ __forceinline__ int frag_row_perm(int lr) { return (lr & 3) | ((lr & 4) << 1) | ((lr & 8) >> 1); }
__device__ __forceinline__ int frag_quad_of_lane(int lq) { return ((lq & 1) << 1) | (lq >> 1); }   // channel quad a lane quad owns
#ifdef I2I_EMU
__device__ __forceinline__ void half_swap(float& a, float& b) { emu::permlane32_swap(a, b); }
#else
__device__ __forceinline__ void half_swap(float& a, float& b) {     // a[lanes 32-63] <-> b[lanes 0-31]
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    a = __builtin_bit_cast(float, (unsigned)r[0]);
    b = __builtin_bit_cast(float, (unsigned)r[1]);
}
#endif
// v[0..7] = the 8 consecutive channels this lane owns after the exchange of fragments a (j) and b (j+1)
__device__ __forceinline__ void widen_pair(f32x4 a, f32x4 b, float v[8]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float x = a[r], y = b[r];
        half_swap(x, y);
        v[r] = x;
        v[4 + r] = y;
    }
}

// max / sum over the four lanes {l, l^16, l^32, l^48} that hold one MFMA column: two VALU swaps
// (v_permlane16_swap / v_permlane32_swap, gfx950) instead of two ds_bpermute round trips through the LDS queue
#ifdef I2I_EMU
__device__ __forceinline__ float quad_max(float v) { v = fmaxf(v, __shfl_xor(v, 16)); return fmaxf(v, __shfl_xor(v, 32)); }
__device__ __forceinline__ float quad_sum(float v) { v += __shfl_xor(v, 16); return v + __shfl_xor(v, 32); }
#else
__device__ __forceinline__ float quad_max(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);       // {rows 0,0,2,2}, {rows 1,1,3,3}
    const float m = fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
    const unsigned w = __builtin_bit_cast(unsigned, m);
    r = __builtin_amdgcn_permlane32_swap(w, w, false, false);            // {lo half, lo half}, {hi half, hi half}
    return fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}
__device__ __forceinline__ float quad_sum(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float m = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
    const unsigned w = __builtin_bit_cast(unsigned, m);
    r = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
#endif

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}
