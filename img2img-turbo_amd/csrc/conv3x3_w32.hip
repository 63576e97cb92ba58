// 3x3 stride-1 pad-1 implicit-GEMM convolution, "wide" operating point for gfx950: 32x32x16 MFMA, wave tiles of
// 128 pixels x 128 channels (4 x 4 fragments = 256 accumulator registers, the whole AGPR file), ONE wave per SIMD, one
// persistent workgroup per CU walking a stream of tiles.  Same replaced reference ops as conv3x3.hip (F.conv2d of
// diffusers ResnetBlock2D with the preceding F.group_norm + F.silu and torch.cat folded into the operand staging, bias /
// residual / next GroupNorm's partial sums in the epilogue); 16-bit dtypes only (the exact-f32 parity mode, planes
// narrower than 32, upsampling gathers and grids that cannot fill the chip stay on conv3x3_halo_kernel).
//
// Why a second operating point: the 64x64 wave tile on 16x16x32 MFMAs at two waves per SIMD is
// issue bound -- per 32 MFMAs (512 matrix cycles) a wave issues 16 ds_read_b128, its DMA share, waits and a barrier.
// Here a k16 step is 16 MFMAs of 32 cycles (512 matrix cycles) against 8 ds_read_b128: half the LDS traffic and half
// the MFMA instructions per FLOP, and with 8 issue slots per MFMA and nobody else on the SIMD the GroupNorm+SiLU of
// the NEXT slab's halo runs in the MFMA shadow instead of in a serial hand-over.
//
// Workgroup tile = TH x 32 output pixels of one image x BN channels; WM x WN waves, wave = FM tile rows (one 32-pixel
// fragment each) x FN 32-channel fragments.  K loop = (64-channel slab) x (9 taps) x (4 k16 steps):
//   * pixels: the (TH+2) x 34 halo of a slab lives in FOUR LDS planes, one per k16 step (rows of 32 bytes, chunk XOR
//     bit 3 of the row): a fragment read (32 consecutive rows at ANY start, lanes 0-31 chunk 0, lanes 32-63 chunk 1)
//     is conflict-free and the k16 step is an immediate offset (tools/lds_bank_model.py).  The NEXT slab's halo is
//     loaded by 16-byte loads hidden from the compiler's waitcnt bookkeeping in the windows of taps 0..5, transformed
//     (GN affine + SiLU) in registers beside the MFMAs of taps 3..8 and stored after the slab's last step barrier.
//   * weights [N][9*Cin]: LDS-DMA into a 3-deep ring of [BN][64] slabs, swizzle carried by the per-lane source
//     address, issued right after the step barrier that frees the slot and waited for two steps later with a counted
//     vmcnt.  Every wave issues the same VMEM operations in every window, branch-free, so the counts are exact.
//   * ONE s_barrier per step (between k16 steps 2 and 3) + one per slab; fragment reads run one k16 step ahead.
//   * the slab stream is CONTINUOUS ACROSS TILES: during a tile's last slab the staged halo, the GroupNorm constants
//     and the ring's tail batches are those of the workgroup's next tile, so a tile costs its MFMAs + its epilogue.
//     With one workgroup per CU nothing else would cover a prologue (measured 20 % of a 2-slab tile, profiles/r3a_*).
//   * epilogue: accumulators (lane = 4 consecutive channels of one pixel per register quad, quads half-exchanged with
//     v_permlane32_swap) -> alpha, bias, residual -> 16-bit -> a wave-private LDS staging block -> read back as whole
//     256-byte pixel rows -> full-line global stores (the accumulator layout would store 32-byte pieces of 32 lines per
//     instruction: store-issue bound, 24 % of a 2-slab tile) and the GroupNorm partial sums of the stored values.
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

constexpr int W32_TW = 32, W32_CK = 64, W32_RING = 3;
// bytes of one k16 plane of the halo image: rows of 32 bytes, padded to 32 (mod 128)
constexpr int w32_plane(int th) { return (((th + 2) * (W32_TW + 2) * 32 + 127) / 128) * 128 + 32; }
// pixels per epilogue staging round per wave (a pixel = the wave's 128 channels = 256 bytes): what the LDS left beside
// the halo planes and the weight ring allows
constexpr int w32_stage_px(int bn) { return bn <= 128 ? 32 : 16; }
constexpr size_t w32_lds_bytes(int th, int bn, int nw) {
    return (size_t)4 * w32_plane(th) + 1024 + (size_t)W32_RING * bn * 128 + 1024 + (size_t)bn * 4 + (size_t)nw * w32_stage_px(bn) * 256;
}

// Tile stream of a launch (host side fills it).  Workgroup b runs on XCD b & 7.  ntn | 8: it serves channel tile XCD % ntn
// and, of the spatial tiles' (8 / ntn)-way split, run XCD / ntn: tiles lg, lg + lgroups, ... with lg = b >> 3.  Otherwise:
// channel tile (b>>3) % ntn, the 8-way split's run XCD, lg = (b>>3) / ntn.  Neighbouring tiles meet in one L2 either way.
struct w32_sched { int ntn, lgroups, tiles_x, tiles_y, nsp; };

// GN: GroupNorm affine + SiLU applied while staging (p.gn_ss != nullptr, p.act == 1); otherwise raw staging.
// RES: residual tensor added in the epilogue (p.res != nullptr).
template <typename T, int TH, int BN, int WM, int WN, bool GN, bool RES>
__global__ __launch_bounds__(WM* WN * 64, (WM * WN) / 4) void conv3x3_w32_kernel(const i2i_igemm_params p, const w32_sched sc) {
    constexpr int TW = W32_TW, CK = W32_CK, RING = W32_RING, NTAPS = 9;
    constexpr int NW = WM * WN, NT = NW * 64;
    constexpr int HW2 = TW + 2, HALO = (TH + 2) * HW2;
    static_assert(TH % WM == 0 && BN % (32 * WN) == 0, "");
    constexpr int FM = TH / WM, WTN = BN / WN, FN = WTN / 32;
    static_assert(WTN == 128, "the epilogue stages 256-byte pixel rows per wave");
    constexpr int HPT = (HALO * 8 + NT - 1) / NT;      // halo chunks per thread per slab
    constexpr int PPJ = NT / 8;                        // halo pixels covered per chunk index j
    constexpr int NPIECE = BN / 8;                     // 1-KiB LDS-DMA pieces per weight slab
    static_assert(NPIECE % NW == 0, "every wave issues the same number of DMA pieces");
    constexpr int BPW = NPIECE / NW;
    constexpr int PLANE = w32_plane(TH);
    constexpr int SPX = w32_stage_px(BN), NRND = 32 / SPX;       // staging rounds per tile row
    typedef typename Elem<T>::chunk_t chunk_t;
    static_assert(Elem<T>::EPC == 8, "16-bit dtypes only");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // SGPR: LDS-DMA destinations are scalar arithmetic
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lh = lane >> 5;
    const int kc = tid & 7;
    // Measurement hook (csrc/build.py --tag trace --defs=-DI2I_TRACE=1; never in the product build): every wave sums the
    // shader cycles (s_memtime) it spends per pipeline segment and writes them to p.ws [workgroup][wave][16]:
    // 0 prologue (once per workgroup), 1 k16 steps 0-2, 2 counted vmcnt wait, 3 step barrier, 4 window + k16 step 3,
    // 5 slab-end wait + barrier + first reads, 6 epilogue, 7 tile set-up (next tile decode, accumulator reset).
    // p.splitk carries ablation bits (results are WRONG with any set): 1 no stores, 4 no weight DMA in the stream, 8 no MFMAs,
    // 16 halo loads all from pixel 0 (cache hits), 32 no GroupNorm transform VALU, 64 no step barriers.
#ifdef I2I_TRACE
    unsigned tr_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned tr_t = (unsigned)__builtin_amdgcn_s_memtime();
#define W32_TR(k) do { const unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime(); tr_acc[k] += t_ - tr_t; tr_t = t_; } while (0)
#if I2I_TRACE >= 2
#define W32_TRS(k) W32_TR(k)      // per-step stamps: four s_memtime round trips per step, they dominate what they measure
#else
#define W32_TRS(k) do { } while (0)
#endif
#define W32_ABL(bit) ((p.splitk & (bit)) != 0)
#else
#define W32_TR(k) do { } while (0)
#define W32_TRS(k) do { } while (0)
#define W32_ABL(bit) false
#endif

    // ---- this workgroup's tile stream.  Channel tiles by XCD when their number divides 8: every CU of an XCD then streams
    // the SAME BN rows of the weight matrix (512 -> 512: 2.4 MB instead of 4.7 MB against a 4 MB L2; with both channel
    // tiles on one XCD the weight DMA alone cost 30 % of the kernel, profiles/r3g_w32_ablation.log).
    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3;
    int tn, lg, s_cur, cnt;
    if (8 % sc.ntn == 0) {
        tn = xcd % sc.ntn;
        lg = lb;
        const int ng = 8 / sc.ntn, grp = xcd / sc.ntn;
        const int q = sc.nsp / ng, r = sc.nsp % ng;
        s_cur = (grp < r ? grp * (q + 1) : r * (q + 1) + (grp - r) * q) + lg;
        cnt = q + (grp < r ? 1 : 0);
    } else {
        tn = lb % sc.ntn;
        lg = lb / sc.ntn;
        const int q = sc.nsp >> 3, r = sc.nsp & 7;
        s_cur = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + lg;
        cnt = q + (xcd < r ? 1 : 0);
    }
    if (lg >= cnt) return;                             // fewer tiles than workgroups (uniform)
    const int n0 = tn * BN;

    const T* __restrict__ bw = (const T*)p.b;
    const int cin = p.c0 + p.c1;
    const int nslab = cin / CK;
    const int64_t img_stride0 = (int64_t)p.hin * p.win * p.lda0 * (int)sizeof(T), img_stride1 = (int64_t)p.hin * p.win * p.lda1 * (int)sizeof(T);

    // LDS map: | 4 halo planes | 1 KiB dummy (stores of the out-of-range lanes of the last chunk row) | weight ring |
    //          | GroupNorm (scale, shift) of the slab being staged, 2 x 512 B | bias of the channel tile | staging, per wave |
    constexpr int HS0 = 0;
    constexpr int DUM0 = 4 * PLANE;
    constexpr int BS0 = DUM0 + 1024;
    constexpr int SS0 = BS0 + RING * BN * 128;
    constexpr int BI0 = SS0 + 1024;
    constexpr int STG0 = BI0 + BN * 4;
    char* Bs = i2i_smem + BS0;
    const bool bias_lds = p.bias_mode == 1;

    // ---- this thread's halo chunks: chunk id v = tid + j*NT -> halo pixel hp = v>>3 = (hy, hx), chunk kc = tid&7.
    // hrel[j] = hy*win + hx is tile independent; the pixel index inside the image is pb + hrel[j] with the tile's
    // pb = (ty0-1)*win + tx0-1 -- unless the pixel is zero padding (bit j of the tile's pad mask: conv pads the
    // ACTIVATED tensor, so padding lanes load pixel 0 and the transform replaces them by zeros).
    // One base + a per-lane wrap mask instead of HPT registers: hp advances by PPJ pixels per j; bit j of `wrapm` says
    // that step j -> j+1 crosses into the next halo row, which adds (win - 34) on top of the PPJ.
    int hp0 = tid >> 3, hy0 = hp0 / HW2, hx0 = hp0 - hy0 * HW2;
    static_assert(PPJ < HW2 || PPJ % HW2 < HW2, "");
    int hrel0 = hy0 * p.win + hx0;
    unsigned wrapm = 0;
    {
        int hx = hx0;
#pragma unroll
        for (int j = 0; j + 1 < HPT; ++j) {
            hx += PPJ % HW2;
            if (hx >= HW2) { hx -= HW2; wrapm |= 1u << j; }
        }
    }
    const int row_skip = p.win - HW2;
    auto hrel = [&](int j) __attribute__((always_inline)) -> int {   // hy_j*win + hx_j
        return hrel0 + j * (PPJ % HW2) + (j * (PPJ / HW2)) * p.win + mul24(__builtin_popcount(wrapm & ((1u << j) - 1u)), row_skip);
    };
    auto pad_of = [&](int ty0, int tx0) __attribute__((always_inline)) -> unsigned {
        unsigned m = 0;
        int hy = hy0, hx = hx0;
#pragma unroll
        for (int j = 0; j < HPT; ++j) {
            const int iy = ty0 + hy - 1, ix = tx0 + hx - 1;
            const bool in = hp0 + j * PPJ < HALO && (unsigned)iy < (unsigned)p.hin && (unsigned)ix < (unsigned)p.win;
            m |= (in ? 0u : 1u) << j;
            hx += PPJ % HW2; hy += PPJ / HW2;
            if (hx >= HW2) { hx -= HW2; ++hy; }
        }
        return m;
    };
    // LDS store address of chunk j: st_off + j*PPJ*32 (row bit 3 does not change with j); the last chunk row may run
    // past the halo: those lanes store into the dummy block
    int st_off = HS0 + (kc >> 1) * PLANE + hp0 * 32 + (((kc & 1) ^ ((hp0 >> 3) & 1)) << 4);
    int st_last = (hp0 + (HPT - 1) * PPJ < HALO) ? st_off + (HPT - 1) * PPJ * 32 : DUM0 + lane * 16;

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
    // slab index past the tile's last = the next tile's slabs (same channel tile: same weights)
    auto b_dma_q = [&](int slab, int tap, int buf, int q) __attribute__((always_inline)) {
        if (W32_ABL(4)) return;
        const int sl = slab < nslab ? slab : slab - nslab;
        const char* src = (const char*)(bw + (tap * cin + sl * CK));
        glds16_sv(src, b_voff[q], Bs + buf * BN * 128 + (wave + q * NW) * 1024);
    };

    // ---- the slab being STAGED (next slab of this tile, or slab 0 of the next tile): wave-uniform descriptor
    const char* sg_base = nullptr;                      // source pointer of its first channel, image included
    unsigned sg_ld = 0;                                 // pixel stride in bytes
    int sg_pb = 0;                                      // pixel base (ty0-1)*win + tx0-1
    unsigned sg_pad = 0;                                // zero-padding mask of this thread's chunks
    const float* sg_ss = nullptr;                       // its 64 (scale, shift) pairs
    auto set_stage = [&](int img, int ty0, int tx0, unsigned pad, int slab) __attribute__((always_inline)) {
        const int ci = slab * CK;
        sg_base = ci < p.c0 ? (const char*)p.a0 + img * img_stride0 + ci * (int)sizeof(T)
                            : (const char*)p.a1 + img * img_stride1 + (ci - p.c0) * (int)sizeof(T);
        sg_ld = (unsigned)(ci < p.c0 ? p.lda0 : p.lda1) * (unsigned)sizeof(T);
        sg_pb = (ty0 - 1) * p.win + tx0 - 1;
        sg_pad = pad;
        if constexpr (GN) sg_ss = p.gn_ss + ((int64_t)img * cin + ci) * 2;
    };

    chunk_t rh[HPT];
    // One 16-byte load per call, always and branch-free
    auto halo_load = [&](int j, bool hidden) __attribute__((always_inline)) {
        const unsigned pix = (unsigned)(sg_pb + hrel(j)) & (((sg_pad >> j) & 1u) - 1u);      // padding lanes: pixel 0, branch-free
        const unsigned voff = pix * sg_ld + (unsigned)kc * 16u;
        if (hidden) gload16_uncounted(rh[j], sg_base, W32_ABL(16) ? (unsigned)kc * 16u : voff);      // (never conditional: the asm's destination must not meet a phi)
        else rh[j] = *(const chunk_t*)(sg_base + voff);
    };
    auto ss_dma = [&]() __attribute__((always_inline)) {            // every wave writes the same 512 bytes: equal VMEM counts
        if constexpr (GN) {
            glds16(sg_ss + (lane & 31) * 4, i2i_smem + SS0);      // 64 lanes x 16 B: lanes 32-63 write a second copy (no exec mask, no branch)
        }
    };
    // GroupNorm affine + SiLU of one parked chunk, in place; padding chunks become exact zeros
    float ssr[16];                                      // (scale, shift) of this thread's 8 channels of the staged slab
    auto load_ssr = [&]() __attribute__((always_inline)) {
        if constexpr (GN) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *(const f32x4*)(i2i_smem + SS0 + kc * 64 + q * 16);
                ssr[4 * q + 0] = v[0]; ssr[4 * q + 1] = v[1]; ssr[4 * q + 2] = v[2]; ssr[4 * q + 3] = v[3];
            }
        }
    };
    // The transform of one chunk in 16 half pieces (hp = 2*element + phase) so that the step loop can pin ~4 VALU beside
    // every MFMA: phase 0 = convert, affine, exponent; phase 1 = reciprocal, product, convert back, zero padding.
    float xf_a = 0.f, xf_t = 0.f, xf_y = 0.f;           // element in flight between its two phases / its dword partner
    auto halo_xform_half = [&](int j, int hp) __attribute__((always_inline)) {
        if (W32_ABL(32)) return;
        const int e = hp >> 1;
        if constexpr (GN) {
            if ((hp & 1) == 0) {
                float sc, sh;
                if constexpr (HPT > 12) {                  // big halos (>= 60 parked registers): constants re-read per element pair
                    if ((e & 1) == 0) {
                        const f32x4 v = *(const f32x4*)(i2i_smem + SS0 + kc * 64 + (e >> 1) * 16);
                        ssr[0] = v[0]; ssr[1] = v[1]; ssr[2] = v[2]; ssr[3] = v[3];
                    }
                    sc = ssr[2 * (e & 1)]; sh = ssr[2 * (e & 1) + 1];
                } else {
                    sc = ssr[2 * e]; sh = ssr[2 * e + 1];
                }
                xf_a = __builtin_fmaf(to_f32<T>(rh[j][e]), sc, sh);
                xf_t = exp2_fast(xf_a * -1.44269504088896341f);
            } else {
                const float y = xf_a * __builtin_amdgcn_rcpf(1.0f + xf_t);
                if ((e & 1) == 0) xf_y = y;
                else {                                     // both halves of a dword: convert, pack, zero padding with ONE select
                    typedef T tx2_t __attribute__((ext_vector_type(2)));
                    tx2_t pk;
                    pk[0] = from_f32<T>(xf_y); pk[1] = from_f32<T>(y);
                    unsigned w = __builtin_bit_cast(unsigned, pk);
                    w = ((sg_pad >> j) & 1u) ? 0u : w;
                    pk = __builtin_bit_cast(tx2_t, w);
                    rh[j][e - 1] = pk[0]; rh[j][e] = pk[1];
                }
            }
        } else {
            if (hp == 15) rh[j] = ((sg_pad >> j) & 1u) ? zero_chunk<T>() : rh[j];
        }
    };
    auto halo_xform = [&](int j) __attribute__((always_inline)) {
#pragma unroll
        for (int hp = 0; hp < 16; ++hp) halo_xform_half(j, hp);
    };
    auto halo_store = [&](int j) __attribute__((always_inline)) {
        *(chunk_t*)(i2i_smem + (j == HPT - 1 ? st_last : st_off + j * PPJ * 32)) = rh[j];
    };

    f32x16 acc[FM][FN];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };

    // ---- tile coordinates
    const int tiles_per_img = sc.tiles_x * sc.tiles_y;
    auto decode = [&](int s, int& img, int& ty0, int& tx0) __attribute__((always_inline)) {
        img = s / tiles_per_img;
        const int t = s - img * tiles_per_img, ty = t / sc.tiles_x;
        ty0 = ty * TH;
        tx0 = (t - ty * sc.tiles_x) * TW;
    };
    int c_img, c_ty0, c_tx0;
    decode(s_cur, c_img, c_ty0, c_tx0);

    // ---- prologue (once per workgroup): weights of steps 0..2, bias, and the first tile's slab 0 synchronously
#pragma unroll
    for (int t = 0; t < RING; ++t)
#pragma unroll
        for (int q = 0; q < BPW; ++q) b_dma_q(0, t, t, q);
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
    set_stage(c_img, c_ty0, c_tx0, pad_of(c_ty0, c_tx0), 0);
    ss_dma();
#pragma unroll
    for (int j = 0; j < HPT; ++j) halo_load(j, false);
    wait_vmcnt<0>();
    lds_barrier();
    load_ssr();
#pragma unroll
    for (int j = 0; j < HPT; ++j) { halo_xform(j); halo_store(j); }
    lds_barrier();

    // ---- per-lane LDS read bases.  Pixel fragment row = u + c with u = wm*FM*34 + l31 (lane) and c = (i+dy)*34 + dx
    // (compile time); bit 3 of row u + c depends on (u + c) mod 16 only: 16 bases x_off[c & 15]; the row part of c and the
    // k16 plane enter as the immediate c*32 + kk*PLANE (no address arithmetic in the loop).
    int x_off[16];
    {
        const int xu = wm * FM * HW2 + l31;
#pragma unroll
        for (int m = 0; m < 16; ++m) x_off[m] = HS0 + xu * 32 + ((lh ^ (((xu + m) >> 3) & 1)) << 4);
    }
    int w_off = BS0 + (wn * WTN + l31) * 128 + ((lh ^ swz3(l31)) << 4);     // fragment j: + j*4096 (swizzle unchanged)

    chunk_t xf[2][FM], wf[2][FN];
    auto xread = [&](int tap, int i, int kk) __attribute__((always_inline)) -> chunk_t {
        const int c = (i + tap / 3) * HW2 + tap % 3;
        return *(const chunk_t*)(i2i_smem + x_off[c & 15] + (kk * PLANE + c * 32));
    };
    auto wread = [&](int buf, int j, int kk) __attribute__((always_inline)) -> chunk_t {
        return *(const chunk_t*)(i2i_smem + (w_off ^ (kk << 5)) + buf * BN * 128 + j * 4096);
    };

    // Window of step s (after P_s, beside k16 step 3): the DMA batch of B[s+3] FIRST, then the halo chunks t, t+5, ... of
    // the staged slab (taps 0..4) and, in window 0, its GroupNorm constants.  vmcnt retires in order: P_s waits for the DMA
    // batch of window s-2, i.e. it leaves that window's halo loads, the whole window s-1 in flight -- a halo load (HBM
    // latency under load: the ablation that served them from cache was 14 % faster) has three steps to land: covered by the
    // wait of P_{t+3}, transformed during step t+4 (q-th chunk of the window beside k16 step q), stored after P_8 -- when
    // every fragment read of the current halo has completed -- in the shadow of the slab's last 16 MFMAs.
    constexpr int LW = 5;
    auto nh = [](int t) constexpr {                    // VMEM operations of window t besides the DMA batch
        int c = 0;
        for (int j = t; j < HPT && t >= 0 && t < LW; j += LW) ++c;
        return c + ((GN && t == 0) ? 1 : 0);
    };
    static_assert(HPT <= 3 * LW, "halo chunks do not fit the windows of taps 0..4 / k16 steps 0..2");

    // One k16 step, pinned MFMA by MFMA (sched_barrier(0) after each: the group solver of sched_group_barrier clumped up to
    // 110 VALU in front of a single MFMA whenever a step carried a transform).  Beside MFMA m: the half pieces of the parked
    // chunk's GroupNorm+SiLU that fall to it (`pre`), fragment read m of the NEXT k16 step (weights first: the i-major MFMA
    // order needs every weight fragment and x[0] at once), then item m - NRD of `post` (DMA pieces / halo stores).
    auto kstep = [&](auto tapc, auto kkc, auto&& pre, auto has_pre_c, auto&& post, auto npost_c) __attribute__((always_inline)) {
        constexpr int tap = decltype(tapc)::value, kk = decltype(kkc)::value;
        constexpr int cur = kk & 1, nxt = cur ^ 1;
        constexpr bool xnext = kk < 3 || tap < NTAPS - 1;          // not across the slab hand-over
        constexpr int ntap = kk < 3 ? tap : tap + 1, nkk = (kk + 1) & 3;
        constexpr int NMM = FM * FN, NPO = decltype(npost_c)::value;
        constexpr bool HP = decltype(has_pre_c)::value;
        constexpr int NRD = FN + (xnext ? FM : 0);
        static_assert(NRD <= NMM, "");
        static_for_w<NMM>([&](auto mc) __attribute__((always_inline)) {
            constexpr int m = decltype(mc)::value;
            if constexpr (HP) {
#pragma unroll
                for (int hp = 0; hp < 16; ++hp)
                    if ((hp * NMM) / 16 == m) pre(hp);
            }
            if constexpr (m < FN) wf[nxt][m] = wread(ntap % RING, m, nkk);            // 9 % 3 == 0: the next slab's tap 0 too
            else if constexpr (m < NRD) xf[nxt][m - FN] = xread(ntap, m - FN, nkk);
            // post items spread over the MFMAs after the reads (all of them beside the last one if there are more)
            constexpr int room = NMM - NRD;
            if constexpr (NPO > 0 && m >= NRD) {
                constexpr int p0 = (NPO * (m - NRD)) / room, p1 = (NPO * (m - NRD + 1)) / room;
                post(icw<p0>{}, icw<p1>{});
            }
            constexpr int i = m / FN, j = m % FN;
            if (!W32_ABL(8)) acc[i][j] = mma32(wf[cur][j], xf[cur][i], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    auto none = [](auto, auto) __attribute__((always_inline)) {};

    constexpr int DMA_OPS = BPW;
    // `settled`: first slab of a tile -- everything issued before it was waited for at the tile border (vmcnt 0), and
    // the epilogue's stores sit in the queue: steps 0 and 1 need nothing new and must not wait behind those stores.
    auto step = [&](int slab, bool settled, auto tapc) __attribute__((always_inline)) {
        constexpr int tap = decltype(tapc)::value;
        // chunks loaded in window tap-4 landed before P_{tap-1}: transform the q-th beside k16 step q
        auto xf_q = [&](auto qc, int hp) __attribute__((always_inline)) {
            constexpr int q = decltype(qc)::value, j = tap - 4 + q * LW;
            if constexpr (tap >= 4 && j < HPT) {
                if (hp == 0) reg_fence(rh[j]);
                halo_xform_half(j, hp);
            }
        };
        auto has_q = [&](int q) constexpr { return tap >= 4 && tap - 4 + q * LW < HPT; };
        kstep(tapc, icw<0>{}, [&](int hp) __attribute__((always_inline)) { xf_q(icw<0>{}, hp); }, icw<has_q(0)>{}, none, icw<0>{});
        kstep(tapc, icw<1>{}, [&](int hp) __attribute__((always_inline)) { xf_q(icw<1>{}, hp); }, icw<has_q(1)>{}, none, icw<0>{});
        kstep(tapc, icw<2>{}, [&](int hp) __attribute__((always_inline)) { xf_q(icw<2>{}, hp); }, icw<has_q(2)>{}, none, icw<0>{});
        __builtin_amdgcn_sched_barrier(0);
        W32_TRS(1);
        // -- P_s: publishes B[s+1] (the DMA batch of window s-2).  Outstanding VMEM allowed: the halo loads (+ constants) of
        //    window s-2 and the whole window s-1.
        if constexpr (tap < 2) {
            if (!settled) wait_vmcnt<nh(tap - 2) + DMA_OPS + nh(tap - 1)>();
        } else {
            wait_vmcnt<nh(tap - 2) + DMA_OPS + nh(tap - 1)>();
        }
        W32_TRS(2);
        if (!W32_ABL(64)) lds_barrier();
        W32_TRS(3);
        // -- after P_s, all beside the MFMAs of k16 step 3 (nothing but the wait and the barrier is serial): the DMA pieces of
        //    B[s+3] into the ring slot step s just released beside its first MFMAs, then (at tap 8) the halo stores, the staged
        //    slab's halo chunks of this window (hidden loads) and, in window 0, its constants.
        constexpr int NST = (tap == NTAPS - 1) ? HPT : 0;
        constexpr int NLD = (tap < LW) ? (HPT - tap + LW - 1) / LW : 0;              // halo loads of this window
        constexpr int NXT = (GN && (tap == 0 || (tap == 3 && HPT <= 12))) ? 1 : 0;    // ss_dma (window 0) / load_ssr (after P_3)
        auto dma_early = [&](int hp) __attribute__((always_inline)) {
            if (hp < BPW) {
                if constexpr (tap + RING < NTAPS) b_dma_q(slab, tap + RING, tap % RING, hp);
                else b_dma_q(slab + 1, tap + RING - NTAPS, tap % RING, hp);
            }
        };
        static_assert(BPW <= 16 && !has_q(3), "");
        kstep(tapc, icw<3>{}, dma_early, icw<1>{},
              [&](auto p0c, auto p1c) __attribute__((always_inline)) {       // items [p0, p1) of (halo stores, halo loads, constants)
                  constexpr int p0 = decltype(p0c)::value, p1 = decltype(p1c)::value;
#pragma unroll
                  for (int it = p0; it < p1; ++it) {
                      if (it < NST) halo_store(it);
                      else if (it < NST + NLD) halo_load(tap + (it - NST) * LW, true);
                      else {
                          if constexpr (tap == 0) ss_dma();
                          if constexpr (tap == 3 && HPT <= 12) load_ssr();    // constants published by P_3; the previous ones died with tap 8
                      }
                  }
              }, icw<NST + NLD + NXT>{});
        __builtin_amdgcn_sched_barrier(0);
        W32_TRS(4);
    };

    // ---- epilogue of one tile.  acc[i][j][r]: pixel (tile row wm*FM+i, column l31), channels j*32 + 8*(r>>2) + 4*lh + (r&3):
    // per register quad a lane owns 4 consecutive channels = an 8-byte piece of the pixel's 256-byte row (the wave's 128
    // channels).  The pieces go straight into a wave-private LDS image of SPX pixel rows (16-byte chunk c of pixel px at
    // c ^ (px & 15): conflict-free for the column-of-pixels piece writes and for the row reads), the image is read back as
    // whole rows -- 16 lanes x 16 bytes per pixel -- for full-line global stores, and the GroupNorm partial sums of the
    // STORED values are taken from the same registers with v_dot2c (2 channels per instruction; the finest group is a
    // quad).  The residual is loaded in the row layout (full lines), staged through the same image and added in fp32
    // before the one rounding.
    typedef T tx4 __attribute__((ext_vector_type(4)));
    typedef T tx2 __attribute__((ext_vector_type(2)));
    const T* __restrict__ res = (const T*)p.res;
    const bool do_stats = p.gn_part != nullptr;
    auto epilogue = [&](int img, int ty0, int tx0) __attribute__((always_inline)) {
        constexpr bool FULL = false;
        // every lane constant of the epilogue is re-derived from an opaque copy of the lane id: otherwise LICM computes
        // the staging / store addresses once per kernel and they sit in (spilled) registers across the whole tile stream
        int elane = lane;
        opaque(elane);
        const int l31 = elane & 31, lh = elane >> 5;
        const int c16 = elane & 15, p4 = elane >> 4;          // row-layout role: chunk c16 of pixels p4, p4+4, ...
        char* const stg = i2i_smem + STG0 + wave * (SPX * 256);
        const int nrb = n0 + wn * WTN + c16 * 8;              // first channel of the row-layout chunk
        const bool nok = FULL || nrb < p.N;
        // byte offsets of this lane's row-layout chunk inside the staging image (pixel k*4 + p4) and of its pieces
        const int rl_off = p4 * 256 + ((c16 ^ p4) << 4);      // + k*1024 + ((k*4) & 12) folded below: (k*4+p4)&15 = (k*4 & 12) | p4
        const unsigned o_lane = (unsigned)(p4 * p.ldc + nrb), r_lane = RES ? (unsigned)(p4 * p.ldr + nrb) : 0u;
        tx2 ones;
        ones[0] = (T)1.0f; ones[1] = (T)1.0f;
        float gs0 = 0.f, gq0 = 0.f, gs1 = 0.f, gq1 = 0.f;     // (sum, sum of squares) of the chunk's two channel quads
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            __builtin_amdgcn_sched_barrier(0);                // one tile row at a time (register budget)
            const int oy = ty0 + wm * FM + i;
            const bool rowok = FULL || oy < p.ho;
            const int64_t rowpix = ((int64_t)img * p.ho + (rowok ? oy : 0)) * p.wo + tx0;     // uniform
            T* const obase = (T*)p.c + rowpix * p.ldc;
            const T* const rbase = RES ? res + rowpix * p.ldr : nullptr;
#pragma unroll
            for (int rd = 0; rd < NRND; ++rd) {
                if constexpr (RES) {                           // (a) residual of the round's pixels, row layout -> stage
                    chunk_t rr[SPX / 4];
#pragma unroll
                    for (int k = 0; k < SPX / 4; ++k) {
                        const int pxr = rd * SPX + k * 4;      // pixel column (without p4) inside the tile row
                        const bool ok = FULL || (rowok && nok && tx0 + pxr + p4 < p.wo);
                        rr[k] = *(const chunk_t*)(rbase + (ok ? (unsigned)(pxr * p.ldr) + r_lane : 0u));
                    }
#pragma unroll
                    for (int k = 0; k < SPX / 4; ++k) *(chunk_t*)(stg + ((rl_off + k * 1024) ^ (((k * 4) & 12) << 4))) = rr[k];
                    wave_sync();
                }
                if (NRND == 1 || (l31 / SPX) == rd) {          // (b) the lanes of this round's pixel columns finish their pieces
                    const int px = l31 % SPX;
                    char* const prow = stg + px * 256 + 8 * lh;
                    const int pswz = (px & 15) << 4;
#pragma unroll
                    for (int j = 0; j < FN; ++j) {
                        __builtin_amdgcn_sched_barrier(0);    // four pieces at a time: 16 accumulators + 16 bias values live
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            char* const a = prow + (((j * 4 + q) << 4) ^ pswz);
                            const f32x4 bq = *(const f32x4*)(i2i_smem + BI0 + (wn * WTN + j * 32 + 8 * q) * 4 + lh * 16);
                            float v[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(p.alpha, acc[i][j][4 * q + r], bq[r]);
                            if constexpr (RES) {
                                const tx4 r4 = *(const tx4*)a;
#pragma unroll
                                for (int r = 0; r < 4; ++r) v[r] += to_f32<T>(r4[r]);
                            }
                            tx4 o4;
#pragma unroll
                            for (int r = 0; r < 4; ++r) o4[r] = from_f32<T>(v[r]);
                            *(tx4*)a = o4;
                        }
                    }
                }
                wave_sync();
                __builtin_amdgcn_sched_barrier(0);
                // (c) whole rows back: full-line stores + statistics of what is stored
                T* const orow = obase + o_lane;
#pragma unroll
                for (int k = 0; k < SPX / 4; ++k) {
                    const int pxr = rd * SPX + k * 4;
                    chunk_t c = *(const chunk_t*)(stg + ((rl_off + k * 1024) ^ (((k * 4) & 12) << 4)));
                    const bool ok = FULL || (rowok && nok && tx0 + pxr + p4 < p.wo);
                    if (ok && !W32_ABL(1)) *(chunk_t*)(orow + (unsigned)(pxr * p.ldc)) = c;
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
        }
        // ---- GroupNorm partial sums: lane -> the 4 lanes sharing a chunk (2 shuffles) -> wave (its staging block) ->
        // workgroup -> one slot per (tile, group).  Fixed order: deterministic.
        if (do_stats) {
            gs0 += __shfl_xor(gs0, 16); gs0 += __shfl_xor(gs0, 32);
            gq0 += __shfl_xor(gq0, 16); gq0 += __shfl_xor(gq0, 32);
            gs1 += __shfl_xor(gs1, 16); gs1 += __shfl_xor(gs1, 32);
            gq1 += __shfl_xor(gq1, 16); gq1 += __shfl_xor(gq1, 32);
            float* st = (float*)stg;                         // [32 quads][2]
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
                    const float* sw = (const float*)(i2i_smem + STG0 + (wmm * WN + wnn) * (SPX * 256));
                    for (int q = q0; q < q0 + nq; ++q) { S += sw[q * 2]; Q += sw[q * 2 + 1]; }
                }
                const int tile_in_img = (ty0 / TH) * sc.tiles_x + tx0 / TW;
                float* out = p.gn_part + (((int64_t)img * tiles_per_img + tile_in_img) * groups + g) * 2;
                out[0] = S;
                out[1] = Q;
            }
            lds_barrier();                                   // the staging blocks are free again
        }
    };
    // first fragments of the first step
    W32_TR(0);
#pragma unroll
    for (int i = 0; i < FM; ++i) xf[0][i] = xread(0, i, 0);
#pragma unroll
    for (int j = 0; j < FN; ++j) wf[0][j] = wread(0, j, 0);

    // ---- the tile stream
    for (int k = 0;; ++k) {
        const bool has_next = lg + (k + 1) * sc.lgroups < cnt;
        opaque(hp0); opaque(hy0); opaque(hx0);
        int n_img = c_img, n_ty0 = c_ty0, n_tx0 = c_tx0;
        if (has_next) decode(s_cur + sc.lgroups, n_img, n_ty0, n_tx0);
        const unsigned c_pad = pad_of(c_ty0, c_tx0), n_pad = pad_of(n_ty0, n_tx0);
        zero_acc();              // at the TOP of the tile: the accumulators are not live around the loop's back edge
        W32_TR(7);
        for (int slab = 0; slab < nslab; ++slab) {
            // what is staged during this slab: the tile's next slab, or slab 0 of the next tile (without a next tile:
            // this tile's slab 0 again -- never read; the operation counts per window do not change)
            if (slab + 1 < nslab) set_stage(c_img, c_ty0, c_tx0, c_pad, slab + 1);
            else set_stage(n_img, n_ty0, n_tx0, n_pad, 0);
            opaque(hrel0); opaque(wrapm); opaque(st_off); opaque(st_last); opaque(w_off);
#pragma unroll
            for (int q = 0; q < BPW; ++q) opaque(b_voff[q]);       // (their 64-bit extensions would otherwise live in register pairs)
            __builtin_amdgcn_sched_barrier(0);
            const bool settled = slab == 0;
            static_for_w<NTAPS>([&](auto tc) __attribute__((always_inline)) { step(slab, settled, tc); });
            if (slab + 1 == nslab) wait_vmcnt<0>();           // tile border: the next tile's B[1], B[2] have landed
            lds_barrier();                                    // the staged halo (stored after P_8) is complete
#pragma unroll
            for (int i = 0; i < FM; ++i) xf[0][i] = xread(0, i, 0);
            W32_TRS(5);
        }
        W32_TR(1);                    // (I2I_TRACE=1: segment 1 = the whole slab loop of the tile)
        epilogue(c_img, c_ty0, c_tx0);
        W32_TR(6);
        if (!has_next) break;
        s_cur += sc.lgroups;
        c_img = n_img; c_ty0 = n_ty0; c_tx0 = n_tx0;
    }
    // the last tile's tail windows issued DMA / halo loads nobody reads: they must land before the workgroup's LDS and
    // registers are released
    wait_vmcnt<0>();
#pragma unroll
    for (int j = 0; j < HPT; ++j) reg_fence(rh[j]);
#ifdef I2I_TRACE
    W32_TR(7);
    if (p.ws && lane == 0) {
        unsigned* o = (unsigned*)p.ws + ((size_t)blockIdx.x * NW + wave) * 16;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = tr_acc[k];
    }
#endif
#undef W32_TR
#undef W32_TRS
#undef W32_ABL
}

// workgroups per (XCD, channel tile) of a launch: the resident ones (one per CU).  I2I_W32_LGROUPS overrides: a test hook
// (small tensors would otherwise never put two tiles on one workgroup), read per launch.
int w32_lgroups(int ntn, int nsp) {
    static int ncu = 0;
    if (!ncu) {
#ifdef I2I_EMU
        ncu = 256;
#else
        hipDeviceProp_t pr;
        int dev = 0;
        ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
#endif
    }
    const bool by_xcd = 8 % ntn == 0;                   // channel tile = XCD % ntn: all workgroups of an XCD share one
    int l = by_xcd ? ncu / 8 : (ncu / 8) / ntn;
    const char* e = getenv("I2I_W32_LGROUPS");
    if (e && atoi(e) > 0) l = atoi(e);
    const int per_stream = by_xcd ? (nsp * ntn + 7) / 8 : (nsp + 7) / 8;      // spatial tiles of one (XCD[, channel tile]) stream
    if (l > per_stream) l = per_stream;
    return l < 1 ? 1 : l;
}

template <typename T, int TH, int BN, int WM, int WN>
int launch_w32(const i2i_igemm_params& p, hipStream_t s) {
    w32_sched sc;
    sc.tiles_x = (p.wo + W32_TW - 1) / W32_TW;
    sc.tiles_y = (p.ho + TH - 1) / TH;
    sc.nsp = sc.tiles_x * sc.tiles_y * p.nimg;
    sc.ntn = (p.N + BN - 1) / BN;
    sc.lgroups = w32_lgroups(sc.ntn, sc.nsp);
    const unsigned wgs = 8u * (unsigned)(8 % sc.ntn == 0 ? 1 : sc.ntn) * (unsigned)sc.lgroups;
    const size_t smem = w32_lds_bytes(TH, BN, WM * WN);
    const dim3 g(wgs), b(WM * WN * 64);
    if (p.gn_ss && p.res) hipLaunchKernelGGL((conv3x3_w32_kernel<T, TH, BN, WM, WN, true, true>), g, b, smem, s, p, sc);
    else if (p.gn_ss) hipLaunchKernelGGL((conv3x3_w32_kernel<T, TH, BN, WM, WN, true, false>), g, b, smem, s, p, sc);
    else if (p.res) hipLaunchKernelGGL((conv3x3_w32_kernel<T, TH, BN, WM, WN, false, true>), g, b, smem, s, p, sc);
    else hipLaunchKernelGGL((conv3x3_w32_kernel<T, TH, BN, WM, WN, false, false>), g, b, smem, s, p, sc);
    return i2i::check_launch("conv3x3_w32");
}

// tile ids 40..49 (i2i_igemm_params.tile): 40 = auto among the w32 configurations
//   41: 8 x 32 px x 256 ch, 4 waves (128 px x 128 ch each, one per SIMD)      42: 12 x 32 px x 128 ch, 4 waves (96 x 128: with
//   16 rows a thread parks 20 halo chunks = 80 registers and the slab loop spills)
// (8-wave forms of the same workgroup tiles -- 128 x 64 / 64 x 128 wave tiles at two waves per SIMD -- were built and
// measured in round 3: within +-3 % of these without the GroupNorm prologue, out of registers with it;
// profiles/r3_w32_ab_nogn.log.  Removed.)
int w32_cfg(const i2i_igemm_params& p) {
    int cfg = p.tile;
    if (cfg == 0 || cfg == 40) cfg = (p.N % 256 == 0) ? 41 : 42;
    return cfg;
}
void w32_cfg_geometry(int cfg, int* th, int* bn, int* wtn) {
    if (cfg == 41) { *th = 8; *bn = 256; *wtn = 128; }
    else { *th = 12; *bn = 128; *wtn = 128; }
}

template <typename T>
int launch_w32_t(const i2i_igemm_params& p, hipStream_t s) {
    switch (w32_cfg(p)) {
        case 41: return launch_w32<T, 8, 256, 2, 2>(p, s);
        case 42: return launch_w32<T, 12, 128, 4, 1>(p, s);
    }
    return i2i::fail(I2I_ERR_BAD_ARG, "conv3x3_w32: unknown tile config %d", p.tile);
}

}  // namespace

namespace i2i {
// Eligibility: 16-bit dtype, 3x3 stride 1 pad 1 without an upsampling gather, 64-aligned channel counts, plane at least
// one 8 x 32 tile, at least 128 output channels, 16-byte epilogue vectors, GroupNorm only together with SiLU.
bool conv3x3_w32_eligible(const i2i_igemm_params& p, int dtype) {
    if (dtype != I2I_BF16 && dtype != I2I_F16) return false;
    if (p.ks != 3 || p.stride != 1 || p.pad != 1 || p.geglu || p.zcount > 1 || p.bias_mode == 2 || p.subpix || p.out_f32 || p.act_out) return false;
    if (p.ups || p.up_h || p.up_w || p.ho != p.hin || p.wo != p.win) return false;
    if (p.c0 % W32_CK || p.c1 % W32_CK || (p.c0 + p.c1) < W32_CK) return false;
    if (p.wo < W32_TW || p.ho < 8 || p.N < 128 || p.N % 8) return false;
    if (p.ldc % 8 || (p.res && p.ldr % 8)) return false;
    if (p.gn_ss && p.act != 1) return false;
    if (!p.gn_ss && p.act) return false;
    return true;
}
// tile == 0 routing: the wide tiles win on every VAE shape that fills the chip (profiles/r3_w32_ab_*.log); channel counts
// that are not multiples of 128 would waste a third of a tile and grids below ~7/8 of the CUs (UNet planes, batch 1) are
// better served by the 8x16 tiles of the halo conv.
bool conv3x3_w32_auto(const i2i_igemm_params& p, int dtype) {
    if (!conv3x3_w32_eligible(p, dtype) || p.N % 128) return false;
    int th, bn, wtn;
    w32_cfg_geometry(w32_cfg(p), &th, &bn, &wtn);
    const long tiles = (long)p.nimg * ((p.ho + th - 1) / th) * ((p.wo + W32_TW - 1) / W32_TW) * ((p.N + bn - 1) / bn);
    return tiles >= 224;
}
int conv3x3_w32_gn_parts(const i2i_igemm_params& p, int dtype, int groups) {
    if (!conv3x3_w32_eligible(p, dtype) || groups < 1 || p.N % groups) return 0;
    const int cpg = p.N / groups;
    int th, bn, wtn;
    w32_cfg_geometry(w32_cfg(p), &th, &bn, &wtn);
    if (cpg % 4 || wtn % cpg || bn % cpg) return 0;        // the epilogue sums channel quads
    return ((p.wo + W32_TW - 1) / W32_TW) * ((p.ho + th - 1) / th);
}
int conv3x3_w32(const i2i_igemm_params& p, int dtype, hipStream_t s) {
    switch (dtype) {
        case I2I_BF16: return launch_w32_t<__bf16>(p, s);
        case I2I_F16: return launch_w32_t<_Float16>(p, s);
    }
    return fail(I2I_ERR_BAD_ARG, "conv3x3_w32: bad dtype");
}
}  // namespace i2i
