// 3x3 stride-1 pad-1 implicit-GEMM convolution, "wide" operating point for gfx950: 32x32x16 MFMA, wave tiles of
// 128 pixels x 128 channels (4 x 4 fragments = 256 accumulator registers, the whole AGPR file) at ONE wave per SIMD
// (128 x 64 / 64 x 128 wave tiles at two waves per SIMD spilled with the GroupNorm prologue and were dropped).  Same replaced reference ops as conv3x3.hip (F.conv2d of diffusers ResnetBlock2D /
// Upsample2D with the preceding F.group_norm + F.silu, torch.cat and nearest-2x folded into the operand staging,
// bias / residual / next GroupNorm's partial sums in the epilogue); 16-bit dtypes only (the exact-f32 parity mode
// and planes narrower than 32 stay on conv3x3_halo_kernel).
//
// Why a second operating point: the 64x64 wave tile on 16x16x32 MFMAs at two waves per SIMD is
// issue bound -- per 32 MFMAs (512 matrix cycles) a wave issues 16 ds_read_b128, its DMA share, waits and a barrier.
// Here a k16 step is 16 MFMAs of 32 cycles (512 matrix cycles) against 8 ds_read_b128: half the LDS traffic and half
// the MFMA instructions per FLOP, and with 8 issue slots per MFMA and nobody else on the SIMD the GroupNorm+SiLU of
// the NEXT slab's halo runs in the MFMA shadow instead of in a serial hand-over.
//
// Workgroup = TH x 32 output pixels of one image x BN channels; WM x WN waves, wave = FM tile rows (one 32-pixel
// fragment each) x FN 32-channel fragments.  K loop = (64-channel slab) x (9 taps) x (4 k16 steps):
//   * pixels: the (TH+2) x 34 halo of a slab is staged once (16-byte loads hidden from the compiler's waitcnt
//     bookkeeping -> GN affine + SiLU in registers, spread over the taps -> ds_write_b128 at the slab hand-over);
//     tap (dy,dx) reads 32 consecutive 128-byte LDS rows starting at (row+dy)*34+dx, lanes 0-31 the even chunk of the
//     k16 step and lanes 32-63 the odd one.  XOR swizzle chunk ^ ((row>>1)&7): conflict-free for every start row
//     (tools/lds_bank_model.py).
//   * weights [N][9*Cin]: LDS-DMA into a 3-deep ring of [BN][64] slabs, swizzle carried by the per-lane source
//     address, issued right after the step barrier that frees the slot and waited for two steps later with a counted
//     vmcnt.  Every wave issues the same VMEM operations in every window, branch-free (tail steps re-fetch valid
//     weights that nobody reads), so the counts are exact.
//   * ONE s_barrier per step, between k16 steps 2 and 3; fragment reads run one k16 step ahead of their MFMAs.
//   * Every MFMA gap is written out in the source and fenced (sched_barrier(0)): MFMA, one fragment read, four INDEPENDENT
//     VALU / transcendental instructions of the staged GroupNorm+SiLU transform, a DMA piece / halo store / hidden load.  At one
//     wave per SIMD a dependent instruction pair stalls the whole SIMD for its latency: the transform runs as eight stages of
//     four elements, a stage per gap (round 5; the list scheduler's own order was one serial chain per element).
//   * MFMA operands swapped (A = weight rows) so an accumulator lane owns 4 consecutive channels of one pixel per
//     register quad; quads are half-exchanged with v_permlane32_swap so every lane stores 16 bytes.
#include <stdlib.h>

#include "i2i_dev.h"
#include "launch.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mma32(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mma32(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

template <int V> struct icw { static constexpr int value = V; };
template <int N, class F> __device__ __forceinline__ void static_for_w(F&& f) {
    if constexpr (N > 0) {
        static_for_w<N - 1>(f);
        f(icw<N - 1>{});
    }
}

__device__ __forceinline__ int swz3(int row) { return (row >> 1) & 7; }

// All lanes of the wave have executed everything before this point (LDS traffic of one wave is processed in issue
// order on the hardware; the CPU emulator runs lanes as independent fibers and needs the rendezvous).
#ifdef I2I_EMU
__device__ __forceinline__ void wave_sync() { (void)__shfl_xor(0, 1); }
#else
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_wave_barrier(); }
#endif

// Re-defines a lane-constant for the optimiser at this point: without it LICM hoists every address / index derived from
// it out of the tile and slab loops (dozens of registers live across the whole kernel -> spills, and a spill reload's
// compiler-inserted vmcnt(0) would also break the hand-counted waits).  Emits no instruction.
#ifdef I2I_EMU
template <typename V> __device__ __forceinline__ void opaque(V&) {}
#else
template <typename V> __device__ __forceinline__ void opaque(V& x) { asm volatile("" : "+v"(x)); }
#endif

// One accumulator value, read out of its AGPR where the epilogue uses it.  With plain uses hipcc gives the accumulators' loop-exit values
// the VGPR class (every use is a VALU instruction) and copies ALL 256 of them out of the AGPRs at the top of the epilogue: three quarters
// fit, the rest are shuffled between AGPRs (75 v_accvgpr_mov) and spilled, and the spill reloads' vmcnt(0) then wait for the tile's own
// global stores.  An "a"-constraint operand keeps the value where the MFMA left it until this instruction.  (The last MFMA and the first
// read are a workgroup barrier and dozens of instructions apart: no hand-placed wait states needed.)
#ifdef I2I_EMU
__device__ __forceinline__ float acc_rd(float a) { return a; }
#else
__device__ __forceinline__ float acc_rd(float a) {
    float v;
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
    return v;
}
#endif

// 0.0f written into an AGPR here and now (see the prologue)
#ifdef I2I_EMU
__device__ __forceinline__ float acc_zero() { return 0.f; }
#else
__device__ __forceinline__ float acc_zero() {
    float a;
    asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(a));
    return a;
}
#endif

// c + a.x*b.x + a.y*b.y in fp32 (v_dot2c_f32_bf16 / v_dot2c_f32_f16)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
#ifdef I2I_EMU
__device__ __forceinline__ float dot2acc(bf16x2_t a, bf16x2_t b, float c) { return fmaf((float)a[1], (float)b[1], fmaf((float)a[0], (float)b[0], c)); }
__device__ __forceinline__ float dot2acc(f16x2_t a, f16x2_t b, float c) { return fmaf((float)a[1], (float)b[1], fmaf((float)a[0], (float)b[0], c)); }
#else
__device__ __forceinline__ float dot2acc(bf16x2_t a, bf16x2_t b, float c) { return __builtin_amdgcn_fdot2_f32_bf16(a, b, c, false); }
__device__ __forceinline__ float dot2acc(f16x2_t a, f16x2_t b, float c) { return __builtin_amdgcn_fdot2(a, b, c, false); }
#endif

#ifdef I2I_EMU
__device__ __forceinline__ int mul24(int a, int b) { return a * b; }
#else
__device__ __forceinline__ int mul24(int a, int b) { return __mul24(a, b); }       // v_mul_i32_i24: full rate (operands < 2^23)
#endif

constexpr int W32_TW = 32, W32_CK = 64, W32_RING = 3, W32_MAX_CIN = 1024;
// weight ring of the sub-pixel form: 4 taps per slab, two stages (a four-deep ring on the 128-channel tile measured the same:
// profiles/r5k_subpix_ring_and_tile.log)
constexpr int w32_spx_ring(int) { return 2; }
constexpr int SPX_ROW_CHUNKS = W32_TW / 4;              // row layout of the epilogue: a lane owns chunk c16 of pixels p4, p4 + 4, ...: 8 per 32-pixel row
// bytes of one k16 plane of the halo image: rows of 32 bytes, padded to 32 (mod 128)
constexpr int w32_plane_px(int halo_px) { return ((halo_px * 32 + 127) / 128) * 128 + 32; }
constexpr int w32_plane(int th, bool subpix = false) { return w32_plane_px(subpix ? (th + 1) * (W32_TW + 1) : (th + 2) * (W32_TW + 2)); }

// GN: GroupNorm affine + SiLU applied while staging (p.gn_ss != nullptr, p.act == 1); otherwise raw staging.
// RES: residual tensor added in the epilogue (p.res != nullptr).
// SUBPIX: the sub-pixel form of "nearest-2x upsample, then 3x3 conv" (diffusers Upsample2D; conv3x3.hip has the same form on
// its tiles): output pixel (2y+a, 2x+b) only sees a 2x2 neighbourhood of the SOURCE plane, so each output parity (a, b) is a
// 2x2 convolution of the source with pre-summed tap weights (packer.subpixel_weights, [4][N][2][2][cin]) -- 4/9 of the MFMA
// work.  A workgroup owns TH x 32 SOURCE positions of one parity: halo (TH+1) x 33 with its origin shifted by the parity,
// 4 taps per slab, 2-deep weight ring (4 % 2 == 0 keeps the ring slot a compile-time constant), outputs scattered to
// (2y+a, 2x+b) as full 256-byte lines.  The Upsample2D conv has neither a norm in front nor a residual.
// K2: the launch carries a second contraction (p.k2_a; its code is compiled into these instantiations only, so the plain ones
// keep their register allocation).
template <typename T, int TH, int BN, int WM, int WN, bool GN, bool RES, bool SUBPIX = false, bool K2 = false>
__global__ __launch_bounds__(WM* WN * 64, (WM * WN) / 4) void conv3x3_w32_kernel(const i2i_igemm_params p) {
    static_assert(!SUBPIX || (!GN && !RES), "");
    static_assert(!K2 || !RES, "a launch with a second contraction has no residual (the contraction replaces it)");
    constexpr int KS = SUBPIX ? 2 : 3, NPAR = SUBPIX ? 4 : 1;
    constexpr int TW = W32_TW, CK = W32_CK, RING = SUBPIX ? w32_spx_ring(BN) : W32_RING, NTAPS = KS * KS;
    static_assert(NTAPS % RING == 0, "the ring slot of a tap is a compile-time constant");
    constexpr int NW = WM * WN, NT = NW * 64;
    constexpr int HW2 = TW + KS - 1, HALO = (TH + KS - 1) * HW2;
    static_assert(TH % WM == 0 && BN % (32 * WN) == 0, "");
    constexpr int FM = TH / WM, WTN = BN / WN, FN = WTN / 32;
    constexpr int HPT = (HALO * 8 + NT - 1) / NT;      // halo chunks per thread per slab
    constexpr int NPIECE = BN / 8;                     // 1-KiB LDS-DMA pieces per weight slab
    static_assert(NPIECE % NW == 0, "every wave issues the same number of DMA pieces");
    constexpr int BPW = NPIECE / NW;
    constexpr int PLANE = w32_plane(TH, SUBPIX);
    typedef typename Elem<T>::chunk_t chunk_t;
    static_assert(Elem<T>::EPC == 8, "16-bit dtypes only");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // SGPR: LDS-DMA destinations are scalar arithmetic
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lh = lane >> 5;
    const int kc = tid & 7;
    // Measurement hook (csrc/build.py --tag trace --defs=-DI2I_TRACE=1; never in the product build): every wave sums the
    // shader cycles (s_memtime) it spends per pipeline segment and writes them to p.ws [workgroup][wave][16]:
    // 0 prologue, 1 k16 steps 0-2, 2 counted vmcnt wait, 3 step barrier, 4 window + k16 step 3, 5 slab-end barrier + first
    // reads, 6 epilogue.  p.splitk carries ablation bits (results are WRONG with any set): 1 no stores, 8 no MFMAs.
#ifdef I2I_TRACE
    unsigned tr_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned tr_t = (unsigned)__builtin_amdgcn_s_memtime();
#define W32_TR(k) do { const unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime(); tr_acc[k] += t_ - tr_t; tr_t = t_; } while (0)
#define W32_ABL(bit) ((p.splitk & (bit)) != 0)
#else
#define W32_TR(k) do { } while (0)
#define W32_ABL(bit) false
#endif

    // ---- XCD-aware tile id: workgroup b runs on XCD b & 7; every XCD gets one contiguous run of (spatial tile, channel tile)
    // pairs, channel tiles fastest, so neighbouring spatial tiles (shared halo rows) and the channel tiles of one spatial tile
    // (same halo) meet in one L2.  (Round 3 also pinned ONE channel tile per XCD so that each L2 streams half / a quarter of the
    // weight matrix; the interleaved same-box A/B of round 4 put it at -0.04 +- 0.4 % of a step -- inside the noise -- and it
    // was removed: profiles/r4_ab_xcdtn_fuse_skip.log.)
    const int pl_h = SUBPIX ? p.hin : p.ho, pl_w = SUBPIX ? p.win : p.wo;       // the plane the tiles walk
    const int tiles_x = (pl_w + TW - 1) / TW, tiles_y = (pl_h + TH - 1) / TH;
    const int ntn = (p.N + BN - 1) / BN;
    int tn, bid;
    {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        tn = bid % ntn; bid /= ntn;
    }
    int pa = 0, pb = 0;                                  // output parity (row, column) of this workgroup: fastest tile index
    if (SUBPIX) { pa = (bid >> 1) & 1; pb = bid & 1; bid >>= 2; }
    const int tx0 = (bid % tiles_x) * TW; bid /= tiles_x;
    const int ty0 = (bid % tiles_y) * TH;
    const int img = bid / tiles_y;
    const int n0 = tn * BN;

    const T* __restrict__ a0 = (const T*)p.a0;
    const T* __restrict__ a1 = (const T*)p.a1;
    const int cin = p.c0 + p.c1;
    const T* __restrict__ bw = (const T*)p.b + (SUBPIX ? (int64_t)(pa * 2 + pb) * p.N * p.ldb : 0);
    // (the tiles walk the SOURCE plane in both forms: the plain form takes no upsampled source -- the planner sends Upsample2D to the
    // sub-pixel form and index-map sizes to the halo conv --, so a halo pixel's address is two integer operations, not up_src()'s float
    // divisions and branches: they were 700 of the prologue's 2 900 instructions, in front of the first load)
    const int hin_up = p.hin, win_up = p.win;

    // LDS map.  Halo: FOUR planes, one per k16 step (channels 16*kk .. +15 of the slab), rows of 32 bytes = 2 chunks,
    // chunk h of row r at physical chunk h ^ ((r>>3)&1): a fragment read (32 consecutive rows, lanes 0-31 chunk 0,
    // lanes 32-63 chunk 1) is conflict-free for every start row, and the k16 step is a pure immediate offset.  PLANE
    // = 32 (mod 128) so that the eight chunks of one pixel (8 consecutive lanes of a ds_write_b128) hit distinct banks.
    // | 4 planes | 1 KiB dummy (stores of the out-of-range lanes of the last chunk row) | weight ring | GN consts | bias |
    constexpr int HS0 = 0;
    constexpr int DUM0 = 4 * PLANE;
    constexpr int BS0 = DUM0 + 1024;
    constexpr int SS0 = BS0 + RING * BN * 128;          // [cin][2] fp32
    const int BI0 = SS0 + (GN ? cin * 8 : 0);           // [BN] fp32
    char* Bs = i2i_smem + BS0;
    const bool bias_lds = p.bias_mode == 1;

    // ---- this thread's halo chunks: chunk id v = tid + j*NT -> halo pixel v>>3, chunk kc = tid&7 (constant).
    // hpix = pixel index inside image `img` (0 for zero padding, flagged in padmask: conv pads the ACTIVATED tensor).
    unsigned hpix[HPT], padmask = 0;
#pragma unroll
    for (int j = 0; j < HPT; ++j) {
        const int hp = (tid >> 3) + j * (NT / 8);
        unsigned pix = 0;
        bool pad = true;
        if (hp < HALO) {
            const int hy = hp / HW2, hx = hp - hy * HW2;
            const int iy = ty0 + hy + pa - 1, ix = tx0 + hx + pb - 1;      // coordinates in the (upsampled) input plane; SUBPIX: source plane
            if ((unsigned)iy < (unsigned)hin_up && (unsigned)ix < (unsigned)win_up) {
                pix = (unsigned)(iy * p.win + ix);
                pad = false;
            }
        }
        hpix[j] = pix;
        padmask |= (pad ? 1u : 0u) << j;
    }
    const char* img0 = (const char*)a0 + (int64_t)img * p.hin * p.win * p.lda0 * (int)sizeof(T);
    const char* img1 = (const char*)a1 + (int64_t)img * p.hin * p.win * p.lda1 * (int)sizeof(T);
    // LDS store address of chunk j: st_off + j * (NT/8)*32 (row bit 3 does not change with j); the last chunk row
    // may run past the halo: those lanes store into the dummy block
    const int st_off = HS0 + (kc >> 1) * PLANE + (tid >> 3) * 32 + (((kc & 1) ^ (((tid >> 3) >> 3) & 1)) << 4);
    const int st_last = ((tid >> 3) + (HPT - 1) * (NT / 8) < HALO) ? st_off + (HPT - 1) * (NT / 8) * 32 : DUM0 + lane * 16;

    // ---- weight DMA: piece pc = wave + q*NW covers LDS rows pc*8 .. +7; lane -> row pc*8 + (lane>>3), physical
    // chunk lane&7 = source chunk (lane&7) ^ swz3(row).  Rows past N are clamped (their columns are never stored).
    unsigned b_voff[BPW];
#pragma unroll
    for (int q = 0; q < BPW; ++q) {
        const int row = (wave + q * NW) * 8 + (lane >> 3);
        int n = n0 + row;
        n = n < p.N ? n : p.N - 1;
        b_voff[q] = (unsigned)(n * p.ldb + (((lane & 7) ^ swz3(row)) * 8)) * (unsigned)sizeof(T);
    }
    const int nslab = cin / CK;
    // step index (slab*9 + tap) may run past the last one at the tail: those fetch slab 0 again (valid, never read)
    auto b_dma_q = [&](int slab, int tap, int buf, int q) __attribute__((always_inline)) {
        const int sl = slab < nslab ? slab : 0;
        const char* src = (const char*)(bw + (tap * cin + sl * CK));
        glds16_sv(src, b_voff[q], Bs + buf * BN * 128 + (wave + q * NW) * 1024);
    };

    chunk_t rh[HPT];
    // One 16-byte load per call, always and branch-free (padding lanes read pixel 0 and are zeroed by the transform)
    auto halo_load = [&](int slab, int j, bool hidden) __attribute__((always_inline)) {
        const int ci = slab * CK;
        const char* base = ci < p.c0 ? img0 + ci * (int)sizeof(T) : img1 + (ci - p.c0) * (int)sizeof(T);
        const unsigned ldb = (unsigned)(ci < p.c0 ? p.lda0 : p.lda1) * (unsigned)sizeof(T);
        const unsigned voff = hpix[j] * ldb + (unsigned)kc * 16u;
        if (hidden) gload16_uncounted(rh[j], base, voff);
        else rh[j] = *(const chunk_t*)(base + voff);
    };
    // GroupNorm affine + SiLU of one parked chunk, in place; padding chunks become exact zeros
    float ssr[16];                                      // (scale, shift) of this thread's 8 channels, slab being staged
    auto load_ssr = [&](int slab) __attribute__((always_inline)) {
        if constexpr (GN) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *(const f32x4*)(i2i_smem + SS0 + (slab * CK + kc * 8) * 8 + q * 16);
                ssr[4 * q + 0] = v[0]; ssr[4 * q + 1] = v[1]; ssr[4 * q + 2] = v[2]; ssr[4 * q + 3] = v[3];
            }
        }
    };
    auto halo_xform = [&](int j) __attribute__((always_inline)) {
        chunk_t c = rh[j];
        if constexpr (GN) {
#pragma unroll
            for (int e = 0; e < 8; ++e) c[e] = from_f32<T>(silu_f(__builtin_fmaf(to_f32<T>(c[e]), ssr[2 * e], ssr[2 * e + 1])));
        }
        rh[j] = ((padmask >> j) & 1u) ? zero_chunk<T>() : c;
    };
    // The same transform as SIXTEEN pieces for the k16 steps of the K loop (one piece per MFMA gap): two half chunks of 4 channels,
    // each as eight stages of 4 independent instructions -- convert, scale/shift, * -log2 e, exp2, + 1, rcp, * y, pack + zero the
    // padding.  An instruction and its consumer are then always one MFMA (32 cycles) apart.  The plain loop above compiles to one
    // serial dependency chain per element (cvt -> fma -> mul -> exp -> add -> rcp -> mul, each waiting for its predecessor's
    // latency), and at one wave per SIMD nothing else issues while the wave waits: those k16 steps ran at ~2x their MFMA time.
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    float xy[2][4], xt[2][4];                           // two chunks in flight in the prologue (sc = 0 / 1), one in the K loop
    chunk_t xc[2];
    auto xform_piece = [&](auto jc, auto mc, auto scc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value, m = decltype(mc)::value, h = m >> 3, st = m & 7, sc = decltype(scc)::value;
        if constexpr (m == 0) { reg_fence(rh[j]); xc[sc] = rh[j]; }
        if constexpr (st == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) xy[sc][e] = to_f32<T>(xc[sc][4 * h + e]);
        } else if constexpr (st == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) xy[sc][e] = __builtin_fmaf(xy[sc][e], ssr[2 * (4 * h + e)], ssr[2 * (4 * h + e) + 1]);
        } else if constexpr (st == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) xt[sc][e] = xy[sc][e] * -1.44269504088896341f;
        } else if constexpr (st == 3) {
#pragma unroll
            for (int e = 0; e < 4; ++e) xt[sc][e] = exp2_fast(xt[sc][e]);
        } else if constexpr (st == 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) xt[sc][e] = xt[sc][e] + 1.0f;
        } else if constexpr (st == 5) {
#pragma unroll
            for (int e = 0; e < 4; ++e) xt[sc][e] = __builtin_amdgcn_rcpf(xt[sc][e]);
        } else if constexpr (st == 6) {
#pragma unroll
            for (int e = 0; e < 4; ++e) xy[sc][e] = xy[sc][e] * xt[sc][e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) xc[sc][4 * h + e] = from_f32<T>(xy[sc][e]);
            if constexpr (h == 0) opaque(xc[sc]);      // (keeps the two v_cvt_pk of this half in this gap: they would sink to the chunk's last use)
            if constexpr (h == 1) {
                u32x4_t u = __builtin_bit_cast(u32x4_t, xc[sc]);
                const bool pad = (padmask >> j) & 1u;
#pragma unroll
                for (int r = 0; r < 4; ++r) u[r] = pad ? 0u : u[r];
                rh[j] = __builtin_bit_cast(chunk_t, u);
            }
        }
    };
    auto halo_store = [&](int j) __attribute__((always_inline)) {
        *(chunk_t*)(i2i_smem + (j == HPT - 1 ? st_last : st_off + j * (NT / 8) * 32)) = rh[j];
    };

    f32x16 acc[FM][FN];

    // ---- prologue: weights of steps 0..2, GN constants of every input channel, bias, halo of slab 0
#pragma unroll
    for (int t = 0; t < RING; ++t)
#pragma unroll
        for (int q = 0; q < BPW; ++q) b_dma_q(0, t, t, q);
    if constexpr (GN) {
        for (int pc = wave; pc * 128 < cin; pc += NW)      // 1 KiB = (scale, shift) of 128 channels per piece
            if (pc * 128 + lane * 2 < cin) glds16(p.gn_ss + ((int64_t)img * cin + pc * 128 + lane * 2) * 2, i2i_smem + SS0 + pc * 1024);
    }
    if (!bias_lds) {                                    // no bias: the epilogue adds zeros
        for (int t = tid; t < BN; t += NT) ((float*)(i2i_smem + BI0))[t] = 0.f;
    }
    if (bias_lds && wave == NW - 1) {
#pragma unroll
        for (int q = 0; q < (BN + 255) / 256; ++q) {
            if (q * 256 + lane * 4 < BN) {
                int n = n0 + q * 256 + lane * 4;
                n = n + 4 <= p.N ? n : p.N - 4;          // lanes past a ragged tile's end re-read its last quad (never stored)
                glds16(p.bias + n, i2i_smem + BI0 + q * 1024);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < HPT; ++j) halo_load(0, j, true);
    // Hidden loads with hand-counted waits, as in the K loop (beside the DMAs hipcc waits vmcnt(0) for a load it can see).  The 256
    // accumulator writes go into their shadow (acc_zero(): volatile, so they stay in front of the wait; as plain constants hipcc
    // re-materialises them in front of the first MFMA, after the second barrier).  The wait below lets exactly the HPT halo loads stay in flight -- every DMA (weights of steps
    // 0..2, GroupNorm constants, bias) is older --, and the transform takes the chunks in arrival order, two at a time.
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = K2 ? 0.f : acc_zero();      // (second-contraction kernels: pinned AGPR writes cost them 230-280 spilled registers)
    __builtin_amdgcn_sched_barrier(0);
    wait_vmcnt<HPT>();
    lds_barrier();
    load_ssr(0);
    if constexpr (GN) {
        // two chunks at a time through the staged transform: every stage is 8 independent instructions (the element-by-element
        // form is a dependent chain of ~9 instructions per element, and nothing else issues on this SIMD while it waits)
        static_for_w<(HPT + 1) / 2>([&](auto pc) __attribute__((always_inline)) {
            constexpr int j0 = 2 * decltype(pc)::value, j1 = j0 + 1;
            wait_vmcnt<(HPT - j0 - 2 > 0 ? HPT - j0 - 2 : 0)>();
            reg_fence(rh[j0]);
            if constexpr (j1 < HPT) reg_fence(rh[j1]);
            static_for_w<16>([&](auto mc) __attribute__((always_inline)) {
                xform_piece(icw<j0>{}, mc, icw<0>{});
                if constexpr (j1 < HPT) xform_piece(icw<j1>{}, mc, icw<1>{});
                __builtin_amdgcn_sched_barrier(0);
            });
            halo_store(j0);
            if constexpr (j1 < HPT) halo_store(j1);
        });
    } else {
        static_for_w<HPT>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            wait_vmcnt<HPT - j - 1>();
            reg_fence(rh[j]);
            halo_xform(j);
            halo_store(j);
        });
    }
    lds_barrier();

    // ---- per-lane LDS read bases.  Pixel fragment row = u + c with u = wm*FM*34 + l31 (lane) and c = (i+dy)*34 + dx
    // (compile time); bit 3 of row u + c depends on (u + c) mod 16 only: 16 bases x_off[c & 15]; the row part of c and
    // the k16 plane enter as the immediate c*32 + kk*PLANE.
    int x_off[16];
    {
        const int u = wm * FM * HW2 + l31;
#pragma unroll
        for (int m = 0; m < 16; ++m) x_off[m] = HS0 + u * 32 + ((lh ^ (((u + m) >> 3) & 1)) << 4);
    }
    const int w_off = BS0 + (wn * WTN + l31) * 128 + ((lh ^ swz3(l31)) << 4);     // fragment j: + j*4096 (swizzle unchanged)

    chunk_t xf[2][FM], wf[2][FN];
    auto xread = [&](int tap, int i, int kk) __attribute__((always_inline)) -> chunk_t {
        const int c = (i + tap / KS) * HW2 + tap % KS;
        return *(const chunk_t*)(i2i_smem + x_off[c & 15] + (kk * PLANE + c * 32));
    };
    auto wread = [&](int buf, int j, int kk) __attribute__((always_inline)) -> chunk_t {
        return *(const chunk_t*)(i2i_smem + (w_off ^ (kk << 5)) + buf * BN * 128 + j * 4096);
    };

    // Halo chunks of the NEXT slab: loaded in the windows of taps 0..5 (chunks t, t+6, ...: a load issued after P_t is
    // covered by the counted wait of P_{t+2}), transformed during step t+3 (q-th chunk of the window beside k16 step q),
    // stored after P_8 -- when every fragment read of the current halo has completed -- in the shadow of the slab's
    // last 16 MFMAs.  ONE extra barrier per slab, no serial hand-over.
    constexpr int LW = SUBPIX ? 2 : 6;
    auto nh = [](int t) constexpr { int c = 0; for (int j = t; j < HPT && t >= 0 && t < LW; j += LW) ++c; return c; };
    static_assert(SUBPIX || HPT <= 4 * LW, "halo chunks do not fit the windows of taps 0..5 / the four k16 steps");

    // One k16 step = NMM = 16 MFMAs (i-major over the wave's FM x FN fragments).
    // jc: icw<j> = the parked chunk this step transforms (GN kernels; -1 = none); post(icw<k>), k < NPO: LDS writers of the step
    // (DMA pieces, halo stores); wload(icw<k>), k < NWL: the window's hidden loads.  Every MFMA gap is its own scheduling region
    // (sched_barrier(0)): gap m = MFMA m, fragment read m of the next k16 step (weights first: the i-major MFMA order needs every
    // weight fragment and x[0] at once), then the post pieces and hidden loads dealt over the gaps without a read, and piece m of
    // the transform.  (sched_group_barrier pipelines over the whole step left this to the list scheduler's register-pressure
    // heuristics, which clumped 30 .. 80 instructions in front of single MFMAs in half of the instantiations.)
    auto kstep = [&](auto tapc, auto kkc, auto jc, auto&& post, auto npost_c, auto&& wload, auto nwl_c) __attribute__((always_inline)) {
        constexpr int tap = decltype(tapc)::value, kk = decltype(kkc)::value, jx = decltype(jc)::value;
        constexpr int cur = kk & 1, nxt = cur ^ 1;
        constexpr bool xnext = kk < 3 || tap < NTAPS - 1;          // not across the slab hand-over
        constexpr int ntap = kk < 3 ? tap : tap + 1, nkk = (kk + 1) & 3;
        constexpr int NRD = FN + (xnext ? FM : 0), NMM = FM * FN, NPO = decltype(npost_c)::value, NWL = decltype(nwl_c)::value;
        constexpr bool HP = jx >= 0 && GN;
        static_assert(NRD < NMM && NMM == 16, "");
        constexpr int G = NMM - NRD;                              // gaps without a fragment read
        if constexpr (jx >= 0 && !GN) { reg_fence(rh[jx]); halo_xform(jx); }      // (raw staging: only the padding is zeroed)
        __builtin_amdgcn_sched_barrier(0);
        static_for_w<NMM>([&](auto mc) __attribute__((always_inline)) {
            constexpr int m = decltype(mc)::value, i = m / FN, j = m % FN;
            if (!W32_ABL(8)) acc[i][j] = mma32(wf[cur][j], xf[cur][i], acc[i][j]);
            if constexpr (m < NRD) {
                // read order: w0, x0, w1 .. w(FN-1), x1 .. x(FM-1)
                if constexpr (m == 0) wf[nxt][0] = wread(ntap % RING, 0, nkk);
                else if constexpr (xnext && m == 1) xf[nxt][0] = xread(ntap, 0, nkk);
                else if constexpr (m < (xnext ? 1 : 0) + FN) wf[nxt][m - (xnext ? 1 : 0)] = wread(ntap % RING, m - (xnext ? 1 : 0), nkk);
                else xf[nxt][m - FN] = xread(ntap, m - FN, nkk);
            } else {
                static_for_w<(NPO + G - 1) / G>([&](auto rc) __attribute__((always_inline)) {
                    constexpr int k = decltype(rc)::value * G + (m - NRD);
                    if constexpr (k < NPO) post(icw<k>{});
                });
                static_for_w<(NWL + G - 1) / G>([&](auto rc) __attribute__((always_inline)) {
                    constexpr int k = decltype(rc)::value * G + (NMM - 1 - m);
                    if constexpr (k < NWL) wload(icw<k>{});
                });
            }
            if constexpr (HP) xform_piece(icw<jx>{}, mc, icw<0>{});
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    constexpr int DMA_OPS = BPW;
    // RES: the LAST slab has no next halo to stage; its hidden-load slots (same count per window, so every counted wait stays
    // exact) fetch the first NPRE tile rows of the RESIDUAL in the epilogue's row layout instead of dummy halo chunks: the
    // epilogue finds them in rh[] with their HBM latency already paid under the slab's MFMAs (it used to expose one load
    // latency per tile row at one row of look-ahead: conv2 of a 128-channel resnet ran 0.12 - 0.19 ms behind its conv1).
    constexpr int RCH = SPX_ROW_CHUNKS;                   // 16-byte chunks of one 32-pixel tile row per lane (row layout)
#ifndef W32_NPRE_MAX
#define W32_NPRE_MAX 4                                    // (0: the A/B baseline -- every residual row is fetched by the epilogue)
#endif
    constexpr int NPRE0 = (RES && !SUBPIX) ? (HPT / RCH < FM ? HPT / RCH : FM) : 0;
    constexpr int NPRE = NPRE0 < W32_NPRE_MAX ? NPRE0 : W32_NPRE_MAX;
    unsigned rp_lane = 0u, rp_okk = 0u;                   // per lane: byte offset of (pixel p4, chunk c16) in a residual row; bit k: pixel k*4 + p4 and the chunk are inside
    auto res_prefetch = [&](int j) __attribute__((always_inline)) {      // j = i * RCH + k: chunk k of tile row i
        const int i = j / RCH, k = j - i * RCH;
        const int oy = ty0 + wm * FM + i;
        const bool rowok = oy < p.ho;
        const char* rbase = (const char*)((const T*)p.res + (((int64_t)img * p.ho + (rowok ? oy : 0)) * p.wo + tx0) * p.ldr);      // uniform
        const bool ok = rowok && ((rp_okk >> k) & 1u);
        gload16_uncounted(rh[j], rbase, ok ? rp_lane + (unsigned)(k * 4 * p.ldr) * (unsigned)sizeof(T) : 0u);
    };
    auto step = [&](int slab, auto tapc, auto lastc) __attribute__((always_inline)) {
        constexpr int tap = decltype(tapc)::value;
        constexpr bool LAST = decltype(lastc)::value && NPRE > 0;
        // chunks loaded in window tap-3 landed before P_{tap-1}: transform the q-th beside k16 step q
        // (SUBPIX: four taps per slab and no transform -- the chunks are fenced and their padding zeroed at the store)
        auto jq = [&](int q) constexpr { return (!SUBPIX && !LAST && tap >= 3 && tap - 3 + q * LW < HPT) ? tap - 3 + q * LW : -1; };
        auto nonek = [](auto) __attribute__((always_inline)) {};
        kstep(tapc, icw<0>{}, icw<jq(0)>{}, nonek, icw<0>{}, nonek, icw<0>{});
        kstep(tapc, icw<1>{}, icw<jq(1)>{}, nonek, icw<0>{}, nonek, icw<0>{});
        kstep(tapc, icw<2>{}, icw<jq(2)>{}, nonek, icw<0>{}, nonek, icw<0>{});
        __builtin_amdgcn_sched_barrier(0);
        W32_TR(1);
        // -- P_s: publishes B[s+1] (issued after P_{s-2}).  Outstanding VMEM allowed = the window issued after P_{s-1}:
        //    its halo loads and its DMA batch.
        //    (two-deep ring: the awaited batch is the youngest operation in flight)
        wait_vmcnt<RING == 2 ? 0 : DMA_OPS + nh(tap - 1)>();
        W32_TR(2);
        lds_barrier();
        W32_TR(3);
        // -- window after P_s, beside the MFMAs of k16 step 3: next slab's halo chunks (hidden loads; from this slab again when
        //    there is no next one, the count per window never changes), the DMA of B[s+3] into the ring slot step s just released
        //    and, at tap 8, the stores of the next slab's halo
        const int hs = slab + 1 < nslab ? slab + 1 : slab;
        constexpr int NWLD = tap < LW ? nh(tap) : 0;
        auto wl = [&](auto kc) __attribute__((always_inline)) {
            constexpr int j = tap + decltype(kc)::value * LW;
            if constexpr (LAST && j < NPRE * RCH) res_prefetch(j);
            else halo_load(hs, j, true);
        };
        constexpr int NST = (tap == NTAPS - 1 && !LAST) ? HPT : 0;
        auto po = [&](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value;
            if constexpr (k < NST) {
                if constexpr (SUBPIX) { reg_fence(rh[k]); halo_xform(k); }
                halo_store(k);
            } else {
                constexpr int q = k - NST;
                if constexpr (tap + RING < NTAPS) b_dma_q(slab, tap + RING, tap % RING, q);
                else b_dma_q(slab + 1, tap + RING - NTAPS, tap % RING, q);
            }
        };
        kstep(tapc, icw<3>{}, icw<jq(3)>{}, po, icw<NST + BPW>{}, wl, icw<NWLD>{});
        __builtin_amdgcn_sched_barrier(0);
        W32_TR(4);
    };

    // first fragments of the first step
    W32_TR(0);
#pragma unroll
    for (int i = 0; i < FM; ++i) xf[0][i] = xread(0, i, 0);
#pragma unroll
    for (int j = 0; j < FN; ++j) wf[0][j] = wread(0, j, 0);

    // (with a residual the last slab is a second copy of the nine steps: no next halo, residual rows in its load slots)
    const int nslab_main = NPRE > 0 ? nslab - 1 : nslab;
    for (int slab = 0; slab < nslab_main; ++slab) {
        load_ssr(slab + 1 < nslab ? slab + 1 : slab);     // constants of the slab whose halo is transformed during this one (from tap 3 on)
        __builtin_amdgcn_sched_barrier(0);                // (its ds_reads must not take the fragment reads' slots in the pinned schedule)
        static_for_w<NTAPS>([&](auto tc) __attribute__((always_inline)) { step(slab, tc, icw<0>{}); });
        lds_barrier();                                    // the next slab's halo (stored after P_8) is complete
#pragma unroll
        for (int i = 0; i < FM; ++i) xf[0][i] = xread(0, i, 0);
        W32_TR(5);
    }
    if constexpr (NPRE > 0) {
        {
            int pl = lane;
            opaque(pl);                                   // (derived here, not hoisted across the main loop)
            const int c16 = pl & 15, p4 = pl >> 4;
            const int nrb = n0 + wn * WTN + c16 * 8;
            rp_lane = (unsigned)(p4 * p.ldr + nrb) * (unsigned)sizeof(T);
#pragma unroll
            for (int k = 0; k < RCH; ++k) rp_okk |= (nrb < p.N && tx0 + k * 4 + p4 < p.wo) ? (1u << k) : 0u;
        }
        __builtin_amdgcn_sched_barrier(0);
        static_for_w<NTAPS>([&](auto tc) __attribute__((always_inline)) { step(nslab - 1, tc, icw<1>{}); });
        W32_TR(5);
    }
    // the tail windows issued DMA and (unused) halo loads: everything must have landed before LDS / registers are reused
    wait_vmcnt<0>();
#pragma unroll
    for (int j = 0; j < HPT; ++j) reg_fence(rh[j]);

    lds_barrier();                                       // every wave is done with the halo planes / the ring: the staging blocks reuse them
    if constexpr (NPRE > 0) {
        // residual rows fetched by the last slab -> their staging blocks, in the epilogue's row layout (pixel k*4 + p4, chunk c16 at
        // p4*256 + ((c16 ^ p4) << 4) + k*1024, bits 6-7 ^ (k*4 & 12)); rh[] is dead from here on
        int sl = lane;
        opaque(sl);
        const int c16 = sl & 15, p4 = sl >> 4;
        char* const blk = i2i_smem + wave * ((NPRE + 1) * W32_TW * 256);
        const int rl = p4 * 256 + ((c16 ^ p4) << 4);
#pragma unroll
        for (int r = 0; r < NPRE; ++r)
#pragma unroll
            for (int k = 0; k < RCH; ++k) *(chunk_t*)(blk + r * (W32_TW * 256) + ((rl + k * 1024) ^ (((k * 4) & 12) << 4))) = rh[r * RCH + k];
        wave_sync();
    }

    // ---- second contraction (p.k2_a): acc += k2_a[pixel][0..k2_c) . k2_b[n][0..k2_c)^T with the pixels taken at the OUTPUT
    // positions of this tile (sub-pixel form: of this parity).  Two users: the decoder's `sample = sample + skip_conv_i(skip *
    // gamma)` (src/model.py:41-43) folded into the Upsample2D conv that produces `sample`, and a ResnetBlock2D's
    // `conv_shortcut(x) + conv2(...)` (diffusers resnet.py: output = shortcut(input) + hidden) folded into its conv2 -- the 1x1
    // convolution costs 64-channel slabs of one tap here instead of a launch of its own plus a write and a re-read of its
    // output.  Per slab: weights by LDS-DMA into ring slots 0 / 1, the tile's TH x 32 pixels (RAW: no GroupNorm on this operand)
    // into the halo planes at (row, column) = tap (0, 0) positions, four k16 steps; the next slab's loads fly under them.
    if constexpr (K2) {
        if (p.k2_a) {
            int t2 = tid;
            opaque(t2);                                   // nothing below shares a value with the staging constants of the main loop
            const int kc2 = t2 & 7, ln2 = t2 & 63;
            const char* k2img = (const char*)p.k2_a + (int64_t)img * p.ho * p.wo * p.k2_lda * (int)sizeof(T);
            unsigned k2pix[HPT], k2ok = 0;
#pragma unroll
            for (int j = 0; j < HPT; ++j) {
                const int hp = (t2 >> 3) + j * (NT / 8);
                const int hy = hp / HW2, hx = hp - hy * HW2;
                const bool ok = hy < TH && hx < TW && ty0 + hy < pl_h && tx0 + hx < pl_w;
                const int oy2 = SUBPIX ? 2 * (ty0 + hy) + pa : ty0 + hy, ox2 = SUBPIX ? 2 * (tx0 + hx) + pb : tx0 + hx;
                k2pix[j] = ok ? (unsigned)(oy2 * p.wo + ox2) * (unsigned)p.k2_lda * (unsigned)sizeof(T) + (unsigned)kc2 * 16u : 0u;
                k2ok |= (ok ? 1u : 0u) << j;
            }
            unsigned k2w[BPW];
#pragma unroll
            for (int q = 0; q < BPW; ++q) {
                const int row = (wave + q * NW) * 8 + (ln2 >> 3);
                int n = n0 + row;
                n = n < p.N ? n : p.N - 1;
                k2w[q] = (unsigned)(n * p.k2_ldb + (((ln2 & 7) ^ swz3(row)) * 8)) * (unsigned)sizeof(T);
            }
            const int nsl2 = p.k2_c / CK;
            // slab sl+1's pixels (registers) and weights (the other ring slot) are requested before slab sl's MFMAs
            chunk_t rk[HPT];
            auto k2_issue = [&](int sl) __attribute__((always_inline)) {
#pragma unroll
                for (int q = 0; q < BPW; ++q)
                    glds16_sv((const char*)((const T*)p.k2_b + sl * CK), k2w[q], Bs + (sl & 1) * BN * 128 + (wave + q * NW) * 1024);
#pragma unroll
                for (int j = 0; j < HPT; ++j) rk[j] = *(const chunk_t*)(k2img + (size_t)(k2pix[j] + (unsigned)(sl * CK * (int)sizeof(T))));
            };
            k2_issue(0);
            for (int sl = 0; sl < nsl2; ++sl) {
                wait_vmcnt<0>();
#pragma unroll
                for (int j = 0; j < HPT; ++j) {
                    rh[j] = ((k2ok >> j) & 1u) ? rk[j] : zero_chunk<T>();
                    halo_store(j);
                }
                lds_barrier();
                if (sl + 1 < nsl2) k2_issue(sl + 1);
                const int wsl = (sl & 1) * BN * 128;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                    for (int j = 0; j < FN; ++j) wf[0][j] = *(const chunk_t*)(i2i_smem + (w_off ^ (kk << 5)) + wsl + j * 4096);
#pragma unroll
                    for (int i = 0; i < FM; ++i) xf[0][i] = xread(0, i, kk);
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int j = 0; j < FN; ++j) acc[i][j] = mma32(wf[0][j], xf[0][i], acc[i][j]);
                }
                lds_barrier();                            // the planes / the slot are free for the next slab (or the epilogue's staging)
            }
        }
    }
    // ---- epilogue of one tile.  acc[i][j][r]: pixel (tile row wm*FM+i, column l31), channels j*32 + 8*(r>>2) + 4*lh + (r&3):
    // per register quad a lane owns 4 consecutive channels = an 8-byte piece of the pixel's 256-byte row (the wave's 128
    // channels).  The pieces go straight into a wave-private LDS image of SPX pixel rows (16-byte chunk c of pixel px at
    // c ^ (px & 15): conflict-free for the column-of-pixels piece writes and for the row reads), the image is read back as
    // whole rows -- 16 lanes x 16 bytes per pixel -- for full-line global stores, and the GroupNorm partial sums of the
    // STORED values are taken from the same registers with v_dot2c (2 channels per instruction; the finest group is a
    // quad).  The residual is loaded in the row layout (full lines), staged through the same image and added in fp32
    // before the one rounding.
    constexpr int SPX = 32, NRND = 1;                    // a whole 32-pixel tile row per staging round
    static_assert(WTN == 128 && BS0 + RING * BN * 128 >= NW * (NPRE + 1) * SPX * 256, "the staging blocks fit in front of the GroupNorm constants / the bias");
    typedef T tx4 __attribute__((ext_vector_type(4)));
    typedef T tx2 __attribute__((ext_vector_type(2)));
    const T* __restrict__ res = (const T*)p.res;
    const bool do_stats = p.gn_part != nullptr;
    auto epilogue = [&](auto fullc, int img, int ty0, int tx0) __attribute__((always_inline)) {
        // FULL: the tile lies inside the plane and the channel range (every tile of the model's shapes at 512 x 512): no store predicates
        // -- hipcc turns each into an exec-mask branch around its store, which also makes its vmcnt bookkeeping pessimistic (a residual
        // row's wait became vmcnt(0): the previous rows' STORES had to complete first) -- and no zeroing of masked chunks
        constexpr bool FULL = decltype(fullc)::value != 0;
        // every lane constant of the epilogue is re-derived from an opaque copy of the lane id: otherwise LICM computes
        // the staging / store addresses once per kernel and they sit in (spilled) registers across the whole tile stream
        int elane = lane;
        opaque(elane);
        const int l31 = elane & 31, lh = elane >> 5;
        const int c16 = elane & 15, p4 = elane >> 4;          // row-layout role: chunk c16 of pixels p4, p4+4, ...
        // the halo planes / the ring are dead: NPRE + 1 staging blocks of 8 KiB per wave from the LDS base (block r < NPRE already
        // holds residual row r, stored right after the K loop; block NPRE serves the other rows)
        char* const stg0 = i2i_smem + wave * ((NPRE + 1) * SPX * 256);
        const int nrb = n0 + wn * WTN + c16 * 8;              // first channel of the row-layout chunk
        const bool nok = FULL || nrb < p.N;
        // byte offsets of this lane's row-layout chunk inside the staging image (pixel k*4 + p4) and of its pieces
        const int rl_off = p4 * 256 + ((c16 ^ p4) << 4);      // + k*1024 + ((k*4) & 12) folded below: (k*4+p4)&15 = (k*4 & 12) | p4
        constexpr int PXS = SUBPIX ? 2 : 1;               // output pixels per tile pixel along a row (sub-pixel form: every other one)
        const unsigned o_lane = (unsigned)(PXS * p4 * p.ldc + nrb), r_lane = RES ? (unsigned)(p4 * p.ldr + nrb) : 0u;
        tx2 ones;
        ones[0] = (T)1.0f; ones[1] = (T)1.0f;
        float gs0 = 0.f, gq0 = 0.f, gs1 = 0.f, gq1 = 0.f;     // (sum, sum of squares) of the chunk's two channel quads
        // The residual is requested one (16 x 32 x 128 tile) or two (8 x 32 x 256) tile rows ahead of its use; the staging
        // registers of the K loop are dead here.  Same-box A/B against fetching each row at its use (W32_RDEPTH=0):
        // +3..5 % on 128 -> 128 @ 512^2 at depth 1, +2 % on 512 -> 512 @ 128^2 at depth 2, nothing on 256 -> 256
        // (profiles/r3m_ab_residual_prefetch_fused_splitk_merged_cross_kv.log).
        static_assert(NRND == 1, "");
#ifndef W32_RDEPTH
#define W32_RDEPTH (BN == 128 ? 1 : 2)
#endif
        constexpr bool RAHEAD = W32_RDEPTH > 0;              // (0: fetch each row at its use -- the A/B baseline)
        // (rows the last slab fetched count towards the look-ahead: more rows in registers than this and hipcc, which parks whole
        // rows of accumulators in VGPRs, spills the residual right after loading it)
        constexpr int RDEPTH0 = !RAHEAD ? 1 : (FM < W32_RDEPTH ? FM : W32_RDEPTH);
        constexpr int RDEPTH = RDEPTH0 - NPRE > 1 ? RDEPTH0 - NPRE : 1;
        // Rows 0 .. NPRE-1 of the residual were fetched by the last slab and already sit in their own staging blocks (below);
        // the other rows go through rr[]: row NPRE + d is requested up front, row i + RDEPTH as soon as row i has been staged.
        static_assert(SPX / 4 == SPX_ROW_CHUNKS, "");
        chunk_t rr[RES ? RDEPTH : 1][SPX / 4];
        auto res_load = [&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            if constexpr (RES && i >= NPRE && i < FM) {
                const int oy = ty0 + wm * FM + i;
                const bool rowok = FULL || oy < p.ho;
                const T* const rbase = res + (((int64_t)img * p.ho + (rowok ? oy : 0)) * p.wo + tx0) * p.ldr;      // uniform
#pragma unroll
                for (int k = 0; k < SPX / 4; ++k) {
                    const bool ok = FULL || (rowok && nok && tx0 + k * 4 + p4 < p.wo);
                    // (uniform 64-bit base + 32-bit BYTE offset per lane: the scalar-base form of the load, no 64-bit VALU address per chunk)
                    rr[(i - NPRE) % RDEPTH][k] = *(const chunk_t*)((const char*)rbase + (ok ? ((unsigned)(k * 4 * p.ldr) + r_lane) * (unsigned)sizeof(T) : 0u));
                }
            }
        };
        // (row j >= NPRE is requested RDEPTH rows ahead of its use: up front when j < RDEPTH, at the start of row j - RDEPTH when that
        // row came staged from the last slab, right after row j - RDEPTH has been staged otherwise)
        if constexpr (RAHEAD) static_for_w<RDEPTH>([&](auto ic) __attribute__((always_inline)) { res_load(ic); });      // (rows < NPRE: no-op)
        static_for_w<FM>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            char* const stg = stg0 + (i < NPRE ? i : NPRE) * (SPX * 256);
            __builtin_amdgcn_sched_barrier(0);                // one tile row at a time (register budget)
            // (16 x 32 x 128 tiles: at the row's start; the 8 x 32 x 256 tile requests it between (b) and (c) -- at the start of its
            // first row hipcc spills the freshly loaded row around the accumulator copies)
            if constexpr (RES && RAHEAD && i < NPRE && i + RDEPTH >= NPRE && BN == 128) res_load(icw<i + RDEPTH>{});
            const int sy = ty0 + wm * FM + i;                 // row in the plane the tiles walk
            const bool rowok = FULL || sy < pl_h;
            const int oy = SUBPIX ? 2 * sy + pa : sy;
            const int64_t rowpix = ((int64_t)img * p.ho + (rowok ? oy : 0)) * p.wo + (SUBPIX ? 2 * tx0 + pb : tx0);     // uniform
            T* const obase = (T*)p.c + rowpix * p.ldc;
            int blane = lh * 16;                              // (re-derived per row: else the 16 bias quads of the wave stay in 64 registers across the rows)
            opaque(blane);
#pragma unroll
            for (int rd = 0; rd < NRND; ++rd) {
                if constexpr (RES && i >= NPRE) {              // (a) residual of the row's pixels, row layout -> stage
                    if constexpr (!RAHEAD) res_load(icw<i>{});
#pragma unroll
                    for (int k = 0; k < SPX / 4; ++k) *(chunk_t*)(stg + ((rl_off + k * 1024) ^ (((k * 4) & 12) << 4))) = rr[(i - NPRE) % RDEPTH][k];
                    wave_sync();
                    if constexpr (RAHEAD) res_load(icw<i + RDEPTH>{});      // the registers are free again: row i + RDEPTH
                }
                if (NRND == 1 || (l31 / SPX) == rd) {          // (b) the lanes of this round's pixel columns finish their pieces
                    const int px = l31 % SPX;
                    char* const prow = stg + px * 256 + 8 * lh;
                    const int pswz = (px & 15) << 4;
                    // The bias quads (and the staged residual quads) of channel fragment j + 1 are read BEFORE fragment j's pieces are
                    // written: hipcc cannot tell the staging image from the bias block (both LDS, dynamic offsets), so a read placed after
                    // a piece's store waits for it -- the first form (read, wait, 4 FMAs, store, per piece) exposed one LDS round trip per
                    // piece, 64-128 per tile (W32_EPI_BATCH=0: that form, the A/B baseline).
#ifndef W32_EPI_BATCH
#define W32_EPI_BATCH 1
#endif
                    // (with a residual the next row's chunks are in flight in rr[] as well: the quads of fragment j are then read at the top of
                    // ITS region -- one round trip per fragment, half the registers -- instead of one fragment ahead)
                    constexpr int PF = RES ? 0 : 1, NB = PF + 1;
                    f32x4 bqv[NB][4];
                    tx4 rqv[NB][4];
                    auto quads = [&](auto jc) __attribute__((always_inline)) {
                        constexpr int j = decltype(jc)::value;
#pragma unroll
                        for (int q = 0; q < 4; ++q) bqv[j % NB][q] = *(const f32x4*)(i2i_smem + BI0 + (wn * WTN + j * 32 + 8 * q) * 4 + blane);
                        if constexpr (RES) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) rqv[j % NB][q] = *(const tx4*)(prow + (((j * 4 + q) << 4) ^ pswz));
                        }
                    };
                    if constexpr (W32_EPI_BATCH && PF) quads(icw<0>{});
                    static_for_w<FN>([&](auto jc) __attribute__((always_inline)) {
                        constexpr int j = decltype(jc)::value;
                        __builtin_amdgcn_sched_barrier(0);    // four pieces at a time: 16 accumulators + 2 x 16 bias values live
                        if constexpr (W32_EPI_BATCH && j + PF < FN) quads(icw<j + PF>{});
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            char* const a = prow + (((j * 4 + q) << 4) ^ pswz);
                            f32x4 bq;
                            if constexpr (W32_EPI_BATCH) bq = bqv[j % NB][q];
                            else bq = *(const f32x4*)(i2i_smem + BI0 + (wn * WTN + j * 32 + 8 * q) * 4 + blane);
                            float v[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(p.alpha, W32_EPI_BATCH ? acc_rd(acc[i][j][4 * q + r]) : acc[i][j][4 * q + r], bq[r]);
                            if constexpr (RES) {
                                tx4 r4;
                                if constexpr (W32_EPI_BATCH) r4 = rqv[j % NB][q];
                                else r4 = *(const tx4*)a;
#pragma unroll
                                for (int r = 0; r < 4; ++r) v[r] += to_f32<T>(r4[r]);
                            }
                            tx4 o4;
#pragma unroll
                            for (int r = 0; r < 4; ++r) o4[r] = from_f32<T>(v[r]);
                            *(tx4*)a = o4;
                        }
                    });
                }
                wave_sync();
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (RES && RAHEAD && i < NPRE && i + RDEPTH >= NPRE && BN != 128) res_load(icw<i + RDEPTH>{});
                // (c) whole rows back: full-line stores + statistics of what is stored
                const unsigned o_lane_b = o_lane * (unsigned)sizeof(T);      // (byte offset: scalar row base + 32-bit lane offset in the store)
                // (the row chunks are read CB at a time and pinned: hipcc otherwise sinks each read into its store's predicate branch --
                // read, lgkmcnt(0), store, eight LDS round trips per tile row; four at a time beside a residual row in flight)
                constexpr int CB = !W32_EPI_BATCH ? 1 : (RES ? 4 : SPX / 4);
                chunk_t cq[CB];
#pragma unroll
                for (int k = 0; k < SPX / 4; ++k) {
                    if (k % CB == 0) {
#pragma unroll
                        for (int kk = 0; kk < CB; ++kk) cq[kk] = *(const chunk_t*)(stg + ((rl_off + (k + kk) * 1024) ^ ((((k + kk) * 4) & 12) << 4)));
                        if constexpr (W32_EPI_BATCH) {
#pragma unroll
                            for (int kk = 0; kk < CB; ++kk) opaque(cq[kk]);
                        }
                    }
                    const int pxr = rd * SPX + k * 4;
                    chunk_t c = cq[k % CB];
                    const bool ok = FULL || (rowok && nok && tx0 + pxr + p4 < pl_w);
                    if (ok && !W32_ABL(1)) *(chunk_t*)((char*)(obase + (unsigned)(PXS * pxr * p.ldc)) + o_lane_b) = c;
                    if (!FULL && !ok) c = zero_chunk<T>();
                    tx2 d0, d1, d2, d3;
                    d0[0] = c[0]; d0[1] = c[1]; d1[0] = c[2]; d1[1] = c[3]; d2[0] = c[4]; d2[1] = c[5]; d3[0] = c[6]; d3[1] = c[7];
                    gs0 = dot2acc(d0, ones, gs0); gs0 = dot2acc(d1, ones, gs0);
                    gq0 = dot2acc(d0, d0, gq0);   gq0 = dot2acc(d1, d1, gq0);
                    gs1 = dot2acc(d2, ones, gs1); gs1 = dot2acc(d3, ones, gs1);
                    gq1 = dot2acc(d2, d2, gq1);   gq1 = dot2acc(d3, d3, gq1);
                }
                wave_sync();                                   // (emulator) the image is free for the next round
            }
        });
        // ---- GroupNorm partial sums: lane -> the 4 lanes sharing a chunk (2 shuffles) -> wave (its staging block) ->
        // workgroup -> one slot per (tile, group).  Fixed order: deterministic.
        if (do_stats) {
            gs0 += __shfl_xor(gs0, 16); gs0 += __shfl_xor(gs0, 32);
            gq0 += __shfl_xor(gq0, 16); gq0 += __shfl_xor(gq0, 32);
            gs1 += __shfl_xor(gs1, 16); gs1 += __shfl_xor(gs1, 32);
            gq1 += __shfl_xor(gq1, 16); gq1 += __shfl_xor(gq1, 32);
            float* st = (float*)stg0;                        // [32 quads][2]
            if (p4 == 0) {
                st[(c16 * 2 + 0) * 2 + 0] = gs0; st[(c16 * 2 + 0) * 2 + 1] = gq0;
                st[(c16 * 2 + 1) * 2 + 0] = gs1; st[(c16 * 2 + 1) * 2 + 1] = gq1;
            }
            lds_barrier();
            const int groups = p.gn_part_groups, cpg = p.N / groups;
            const int ng_tile = BN / cpg;
            const int etid = wave * 64 + elane;
            const int g = n0 / cpg + etid;
            if (etid < ng_tile && g < groups) {
                const int c0w = etid * cpg, wnn = c0w / WTN, q0 = (c0w - wnn * WTN) >> 2, nq = cpg >> 2;
                float S = 0.f, Q = 0.f;
                for (int wmm = 0; wmm < WM; ++wmm) {
                    const float* sw = (const float*)(i2i_smem + (wmm * WN + wnn) * ((NPRE + 1) * SPX * 256));
                    for (int q = q0; q < q0 + nq; ++q) { S += sw[q * 2]; Q += sw[q * 2 + 1]; }
                }
                const int tile_in_img = ((ty0 / TH) * tiles_x + tx0 / TW) * NPAR + pa * 2 + pb;      // (one slot per tile and parity)
                float* out = p.gn_part + (((int64_t)img * (tiles_x * tiles_y * NPAR) + tile_in_img) * groups + g) * 2;
                out[0] = S;
                out[1] = Q;
            }
            lds_barrier();                                   // the staging blocks are free again
        }
    };
#ifndef W32_EPI_FULL
#define W32_EPI_FULL 1
#endif
    if (W32_EPI_FULL && ty0 + TH <= pl_h && tx0 + TW <= pl_w && n0 + BN <= p.N) epilogue(icw<1>{}, img, ty0, tx0);      // (uniform)
    else epilogue(icw<0>{}, img, ty0, tx0);
#ifdef I2I_TRACE
    W32_TR(6);
    if (p.ws && lane == 0) {
        unsigned* o = (unsigned*)p.ws + ((size_t)blockIdx.x * NW + wave) * 16;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = tr_acc[k];
    }
#endif
#undef W32_TR
#undef W32_ABL
}

template <typename T, int TH, int BN, int WM, int WN>
int launch_w32(const i2i_igemm_params& p, hipStream_t s) {
    const int pl_h = p.subpix ? p.hin : p.ho, pl_w = p.subpix ? p.win : p.wo;
    const int nsp = ((pl_w + W32_TW - 1) / W32_TW) * ((pl_h + TH - 1) / TH) * p.nimg * (p.subpix ? 4 : 1), ntn = (p.N + BN - 1) / BN;
    const unsigned tiles = (unsigned)(nsp * ntn);
    const bool gn = p.gn_ss != nullptr;
    const size_t smem = 4 * w32_plane(TH) + 1024 + W32_RING * BN * 128 + (gn ? (size_t)(p.c0 + p.c1) * 8 : 0) + BN * 4;
    const dim3 g(tiles), b(WM * WN * 64);
    if (p.subpix) {
        const size_t smem_sp = 4 * w32_plane(TH, true) + 1024 + w32_spx_ring(BN) * BN * 128 + BN * 4;
        if (p.k2_a) hipLaunchKernelGGL((conv3x3_w32_kernel<T, TH, BN, WM, WN, false, false, true, true>), g, b, smem_sp, s, p);
        else hipLaunchKernelGGL((conv3x3_w32_kernel<T, TH, BN, WM, WN, false, false, true, false>), g, b, smem_sp, s, p);
        return i2i::check_launch("conv3x3_w32<SUBPIX>");
    }
    if (p.k2_a) {        // (eligibility: no residual beside a second contraction)
        if (gn) hipLaunchKernelGGL((conv3x3_w32_kernel<T, TH, BN, WM, WN, true, false, false, true>), g, b, smem, s, p);
        else hipLaunchKernelGGL((conv3x3_w32_kernel<T, TH, BN, WM, WN, false, false, false, true>), g, b, smem, s, p);
        return i2i::check_launch("conv3x3_w32<K2>");
    }
    if (gn && p.res) hipLaunchKernelGGL((conv3x3_w32_kernel<T, TH, BN, WM, WN, true, true>), g, b, smem, s, p);
    else if (gn) hipLaunchKernelGGL((conv3x3_w32_kernel<T, TH, BN, WM, WN, true, false>), g, b, smem, s, p);
    else if (p.res) hipLaunchKernelGGL((conv3x3_w32_kernel<T, TH, BN, WM, WN, false, true>), g, b, smem, s, p);
    else hipLaunchKernelGGL((conv3x3_w32_kernel<T, TH, BN, WM, WN, false, false>), g, b, smem, s, p);
    return i2i::check_launch("conv3x3_w32");
}

// tile ids 40..49 (i2i_igemm_params.tile): 40 = auto among the w32 configurations
//   41: 8 x 32 px x 256 ch, 4 waves (128 px x 128 ch each, one per SIMD)      42: 16 x 32 px x 128 ch, 4 waves (128 x 128)
// (8-wave forms of the same workgroup tiles -- 128 x 64 / 64 x 128 wave tiles at two waves per SIMD -- were built and
// measured in round 3: within +-3 % of these without the GroupNorm prologue, out of registers with it;
// profiles/r3_w32_ab_nogn.log.  Removed.)
int w32_cfg(const i2i_igemm_params& p) {
    int cfg = p.tile;
#ifndef W32_AUTO_41
#define W32_AUTO_41 1      // (A/B builds: 0 = the 16 x 32 x 128 tile everywhere; 2 = everywhere but the sub-pixel form)
#endif
    if (cfg == 0 || cfg == 40) cfg = (p.N % 256 == 0 && (W32_AUTO_41 == 1 || (W32_AUTO_41 == 2 && !p.subpix))) ? 41 : 42;
    return cfg;
}
void w32_cfg_geometry(int cfg, int* th, int* bn, int* wtn) {
    if (cfg == 41) { *th = 8; *bn = 256; *wtn = 128; }
    else { *th = 16; *bn = 128; *wtn = 128; }
}

}  // namespace

// ---- The four (dtype, tile) families of instantiations.  W32_PART (0 .. 3, csrc/build.py and tests/emu/build_emu.py) compiles ONE
// family per translation unit -- the whole file is 32 kernels of 35 - 65 KB of code each, 4 minutes of hipcc and 20 of host clang
// for the emulator in one unit; part 0 also holds the dispatcher and the eligibility rules.  Without W32_PART: everything.
namespace i2i {
int w32_launch_bf16_41(const i2i_igemm_params& p, hipStream_t s);
int w32_launch_bf16_42(const i2i_igemm_params& p, hipStream_t s);
int w32_launch_f16_41(const i2i_igemm_params& p, hipStream_t s);
int w32_launch_f16_42(const i2i_igemm_params& p, hipStream_t s);
#if !defined(W32_PART) || W32_PART == 0
int w32_launch_bf16_41(const i2i_igemm_params& p, hipStream_t s) { return launch_w32<__bf16, 8, 256, 2, 2>(p, s); }
#endif
#if !defined(W32_PART) || W32_PART == 1
int w32_launch_bf16_42(const i2i_igemm_params& p, hipStream_t s) { return launch_w32<__bf16, 16, 128, 4, 1>(p, s); }
#endif
#if !defined(W32_PART) || W32_PART == 2
int w32_launch_f16_41(const i2i_igemm_params& p, hipStream_t s) { return launch_w32<_Float16, 8, 256, 2, 2>(p, s); }
#endif
#if !defined(W32_PART) || W32_PART == 3
int w32_launch_f16_42(const i2i_igemm_params& p, hipStream_t s) { return launch_w32<_Float16, 16, 128, 4, 1>(p, s); }
#endif
}  // namespace i2i

#if !defined(W32_PART) || W32_PART == 0
namespace i2i {
// Eligibility: 16-bit dtype, 3x3 stride 1 pad 1 over a source of the output's size, 64-aligned channel
// counts, plane at least one 8 x 32 tile, at least 128 output channels, 16-byte epilogue vectors, GroupNorm only
// together with SiLU; the sub-pixel upsampler form on source planes of at least 8 x 32.
bool conv3x3_w32_eligible(const i2i_igemm_params& p, int dtype) {
    if (dtype != I2I_BF16 && dtype != I2I_F16) return false;
    if (p.ks != 3 || p.stride != 1 || p.pad != 1 || p.geglu || p.zcount > 1 || p.bias_mode == 2 || p.out_f32 || p.act_out) return false;
    if (p.c0 % W32_CK || p.c1 % W32_CK || (p.c0 + p.c1) < W32_CK || (p.c0 + p.c1) > W32_MAX_CIN) return false;
    if (p.N < 128 || p.N % 8 || p.ldc % 8 || (p.res && p.ldr % 8)) return false;
    if (p.k2_a && (p.res || !p.k2_b || p.k2_c < W32_CK || p.k2_c % W32_CK || p.k2_lda % 8 || p.k2_ldb % 8 || p.k2_lda < p.k2_c || p.k2_ldb < p.k2_c ||
                   (((uintptr_t)p.k2_a | (uintptr_t)p.k2_b) & 15))) return false;
    if (p.subpix) {      // sub-pixel upsampler: tiles walk the SOURCE plane, weights [4 parities][N][4*cin]; no norm, no residual
        if (p.ups != 1 || p.up_h || p.up_w || p.gn_ss || p.act || p.res || p.ldb != 4 * (p.c0 + p.c1)) return false;
        return p.win >= W32_TW && p.hin >= 8 && p.ho == 2 * p.hin && p.wo == 2 * p.win;
    }
    if (p.wo < W32_TW || p.ho < 8) return false;
    if (p.ups || p.up_h || p.up_w || p.ho != p.hin || p.wo != p.win) return false;      // (upsampled sources: the sub-pixel form above, or the halo conv)
    if (p.gn_ss && p.act != 1) return false;
    if (!p.gn_ss && p.act) return false;
    return true;
}
// tile == 0 routing: the wide tiles win on every VAE shape that fills the chip (profiles/r3_w32_ab_*.log: +20..24 % over
// the halo conv with the GroupNorm prologue); channel counts that are not multiples of 128 would waste a third of a tile
// and grids below ~7/8 of the CUs (UNet planes, batch 1) are better served by the 8x16 tiles of the halo conv.
bool conv3x3_w32_auto(const i2i_igemm_params& p, int dtype) {
    if (!conv3x3_w32_eligible(p, dtype) || p.N % 128) return false;
    int th, bn, wtn;
    w32_cfg_geometry(w32_cfg(p), &th, &bn, &wtn);
    const int pl_h = p.subpix ? p.hin : p.ho, pl_w = p.subpix ? p.win : p.wo;
    const long tiles = (long)p.nimg * ((pl_h + th - 1) / th) * ((pl_w + W32_TW - 1) / W32_TW) * ((p.N + bn - 1) / bn) * (p.subpix ? 4 : 1);
    return tiles >= 224;
}
int conv3x3_w32_gn_parts(const i2i_igemm_params& p, int dtype, int groups) {
    if (!conv3x3_w32_eligible(p, dtype) || groups < 1 || p.N % groups) return 0;
    const int cpg = p.N / groups;
    int th, bn, wtn;
    w32_cfg_geometry(w32_cfg(p), &th, &bn, &wtn);
    if (cpg % 4 || wtn % cpg || bn % cpg) return 0;        // the epilogue sums channel quads
    const int pl_h = p.subpix ? p.hin : p.ho, pl_w = p.subpix ? p.win : p.wo;
    return ((pl_w + W32_TW - 1) / W32_TW) * ((pl_h + th - 1) / th) * (p.subpix ? 4 : 1);      // one slot per tile (and parity)
}
int conv3x3_w32(const i2i_igemm_params& p, int dtype, hipStream_t s) {
    const int cfg = w32_cfg(p);
    if (cfg != 41 && cfg != 42) return fail(I2I_ERR_BAD_ARG, "conv3x3_w32: unknown tile config %d", p.tile);
    switch (dtype) {
        case I2I_BF16: return cfg == 41 ? w32_launch_bf16_41(p, s) : w32_launch_bf16_42(p, s);
        case I2I_F16: return cfg == 41 ? w32_launch_f16_41(p, s) : w32_launch_f16_42(p, s);
    }
    return fail(I2I_ERR_BAD_ARG, "conv3x3_w32: bad dtype");
}
}  // namespace i2i
#endif  // W32_PART == 0
