// Plain GEMM (nn.Linear / 1x1 convolution) on the "wide" operating point of gfx950: 32x32x16 MFMA, one wave per SIMD,
// wave tiles of 64 (or 32) rows x 160 (or 128) columns.  Replaces, for the 16-bit dtypes, the cuBLAS calls under
// diffusers' BasicTransformerBlock (attn to_q|k / to_out, GEGLU ff.net.0, ff.net.2), Transformer2DModel.proj_in / proj_out
// and the 1x1 conv_shortcut of ResnetBlock2D reached from the UNet call at src/pix2pix_turbo.py:199, with bias, residual
// add and GEGLU folded into the epilogue.  Same contract as igemm.hip for ks == 1:
//     C[m][n] = epilogue(alpha * sum_k A[m][k] * B[n][k]),   A = [c0 | c1] channel-concatenated token rows, B = weights [N][K].
//
// Why a second GEMM engine next to gemm_dma.hip (profiles/r4_pmc_igemm_*): the 16x16x32 / two-waves-
// per-SIMD engine needs one ds_read_b128 per two MFMAs of 16 matrix cycles and tops out at 500-600 TFLOP/s on these
// shapes; and its 256x128 / 128x128 tiles do not divide this model's widths (N = 320 * 2^k: a third of the last column
// tile is padding, 384 tiles on 256 CUs run as 1.5 rounds).  Here
//   * a k16 step of a 64 x 160 wave tile is 10 MFMAs of 32 matrix cycles against 7 ds_read_b128;
//   * the workgroup tile is 256 x 160 (or 128 x 160 / 256 x 128 / 128 x 128): 320, 640, 1280, 2560, 5120 and 10240 are
//     multiples of 160, and at batch 8 the row counts 32768 / 8192 / 2048 make the tile counts multiples of the 256 CUs;
//   * both operands reach LDS by global_load_lds_dwordx4 in full 128-byte lines (8 rows per 1-KiB piece, XOR swizzle carried
//     by the per-lane SOURCE address) into a 3-deep ring of 64-wide K stages, waited for with a counted vmcnt; ONE raw
//     s_barrier per stage, placed before the last k16 step; the DMA pieces of the stage after next are spread over the
//     following four k16 steps, one per MFMA slot, order pinned with sched_group_barrier;
//   * MFMA operands swapped (A-operand = weight rows): a lane owns 4 consecutive columns of one row per register quad, quads
//     are half-exchanged with v_permlane32_swap so that every lane stores (and fetches the residual as) 16 bytes.
// 16-bit dtypes only; the exact-f32 parity mode, split-K shapes and everything with a gather stay on gemm_dma.hip.
#include <stdlib.h>

#include "i2i_dev.h"
#include "launch.h"

namespace i2i {
int splitk_reduce(const i2i_igemm_params& p, int dtype, hipStream_t s);      // gemm_dma.hip: sums the fp32 slices, applies the epilogue
}

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mma32(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mma32(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

template <int V> struct icg { static constexpr int value = V; };
// 0.0f written into an accumulator register here and now (volatile: stays where it is written)
#ifdef I2I_EMU
__device__ __forceinline__ float g32_acc_zero() { return 0.f; }
#else
__device__ __forceinline__ float g32_acc_zero() {
    float a;
    asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(a));
    return a;
}
#endif
template <int N, class F> __device__ __forceinline__ void static_for_g(F&& f) {
    if constexpr (N > 0) {
        static_for_g<N - 1>(f);
        f(icg<N - 1>{});
    }
}

__device__ __forceinline__ int gswz3(int row) { return (row >> 1) & 7; }

// scalar value the optimiser must not look through (emits no instruction)
#ifdef I2I_EMU
__device__ __forceinline__ void sopaque(unsigned&) {}
#else
__device__ __forceinline__ void sopaque(unsigned& x) { asm volatile("" : "+v"(x)); }
#endif

// wave-level ordering point between LDS writes and reads of OTHER lanes of the same wave (the hardware's LDS queue is in order per wave:
// no instruction; the emulator's lanes are fibers and need the rendezvous)
#ifdef I2I_EMU
__device__ __forceinline__ void g32_wave_sync() { int z = 0; (void)__shfl_xor(z, 1); }
#else
__device__ __forceinline__ void g32_wave_sync() { __builtin_amdgcn_wave_barrier(); }
#endif

__device__ __attribute__((aligned(16))) uint32_t g32_zero16[4] = {0u, 0u, 0u, 0u};   // DMA source of the dummy pieces

// c + a.x*b.x + a.y*b.y in fp32 (v_dot2c_f32_bf16 / v_dot2c_f32_f16): two stored channels per instruction for the statistics
typedef __bf16 g32_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 g32_f16x2 __attribute__((ext_vector_type(2)));
#ifdef I2I_EMU
__device__ __forceinline__ float g32_dot2(g32_bf16x2 a, g32_bf16x2 b, float c) { return fmaf((float)a[1], (float)b[1], fmaf((float)a[0], (float)b[0], c)); }
__device__ __forceinline__ float g32_dot2(g32_f16x2 a, g32_f16x2 b, float c) { return fmaf((float)a[1], (float)b[1], fmaf((float)a[0], (float)b[0], c)); }
#else
__device__ __forceinline__ float g32_dot2(g32_bf16x2 a, g32_bf16x2 b, float c) { return __builtin_amdgcn_fdot2_f32_bf16(a, b, c, false); }
__device__ __forceinline__ float g32_dot2(g32_f16x2 a, g32_f16x2 b, float c) { return __builtin_amdgcn_fdot2(a, b, c, false); }
#endif

constexpr int G32_NW = 4, G32_BK = 64, G32_KQ = 4;       // waves per workgroup (stacked along M), K per stage, k16 steps per stage

// FMW / FNW: 32-row / 32-column fragments per wave.  Workgroup tile = (4 * 32 * FMW) rows x (32 * FNW) columns.
// GATHER: the A operand is the im2col view of a 3x3 convolution (stride 1 or 2, zero padding 0 or 1 incl. the VAE downsamplers'
// F.pad(0,1,0,1)) over ONE NHWC source with cin % 64 == 0: stage s = 64 channels of tap s*64 / cin, the tap's pixel offset is
// a scalar added to the per-lane source offset of every A piece, taps outside the plane read a 16-byte zero block.
// STATS: the epilogue also emits the GroupNorm partial sums of the STORED output (p.gn_part), one slot per (row tile, group).
// SPLITK: grid y = p.splitk slices of the K stages; every slice writes its raw fp32 accumulators to p.ws [slice][M][N] and
// splitk_reduce (gemm_dma.hip) sums them in a fixed order and applies the epilogue -- the 3x3 convolutions of the UNet's 8 x 8 /
// 16 x 16 planes at batch 8 (M = 512 .. 2048 rows against K = 11520 .. 23040: 16 - 32 tiles would leave the chip empty).
// LNF: LayerNorm folded into the GEMM (i2i_igemm_params.ln_cs): the A rows are the UN-normalised tokens, the weights carry the LayerNorm
// weight (W' = W * gamma, folded by the device-side merge), and the epilogue computes rstd * (acc - mu * colsum[n]) + bias'[n].  mu / rstd of
// a row come from the row fragments the MFMAs consume anyway: 8 v_dot2c per fragment and k16 step (sum and sum of squares of the lane's 8
// elements), in the MFMA shadow.  With n_trans the column tiles at and beyond it issue their MFMAs with the operands in the OTHER order
// (A-operand = token rows): an accumulator lane then owns 8 consecutive TOKENS of one output column after the half exchange and stores
// them as 16 bytes of the transposed output c2 -- the self-attention V^T [C][B*T] -- so to_q | to_k | to_v is one launch.
template <typename T, int FMW, int FNW, int RING, bool GEGLU, bool GATHER = false, bool STATS = false, bool SPLITK = false, bool LNF = false>
__global__ __launch_bounds__(G32_NW * 64, 1) void gemm_w32_kernel(const i2i_igemm_params p) {
    static_assert(!(GEGLU && (GATHER || STATS)) && !(SPLITK && (GEGLU || STATS)), "");
    static_assert(!(LNF && (GATHER || STATS || SPLITK)), "");
    constexpr int NW = G32_NW, BK = G32_BK, KQ = G32_KQ;
    constexpr int WTM = 32 * FMW, BM = NW * WTM, BN = 32 * FNW;
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int PA = BM / 8, PB = BN / 8;               // 1-KiB DMA pieces (8 rows of 128 bytes) per stage
    static_assert(PA % NW == 0 && PB % NW == 0, "every wave issues the same pieces");
    constexpr int QA = PA / NW, QB = PB / NW, OPB = QA + QB;
    constexpr int NMM = FMW * FNW, NRD = FMW + FNW;
    static_assert(RING >= 2 && RING <= 4, "");
    typedef typename Elem<T>::chunk_t chunk_t;
    static_assert(Elem<T>::EPC == 8, "16-bit dtypes only");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;

    // XCD-aware tile order (workgroup b -> XCD b % 8): every XCD gets a contiguous run of tiles.  The SMALLER operand is the
    // one re-read along a run: column tiles fastest when the weight matrix is the smaller one (N <= M: an XCD's run walks
    // the weights while its A row blocks stay put), row tiles fastest otherwise (M = 2048 token rows against 10240 weight
    // rows: the LDS-DMA igemm's column-fastest order re-fetched 853 MB per launch through the fabric for 31 MB of operands,
    // profiles/r4_pmc_igemm_summary.txt).
    int bid;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
    const bool m_fast = p.N > p.M;
    const int m0 = (m_fast ? bid % ntm : bid / ntn) * BM, n0 = (m_fast ? bid / ntm : bid % ntn) * BN;
    // LNF with a transposed column range: the tiles at and beyond n_trans run a second copy of the WHOLE body with the MFMA operands in the
    // other order (uniform over the workgroup; two straight-line bodies instead of one body with 160 accumulator phis at every join)
    const bool trans = LNF && !GEGLU && p.n_trans > 0 && n0 >= p.n_trans;
    auto body = [&](auto trc) __attribute__((always_inline)) {
    constexpr bool TR = decltype(trc)::value != 0;
    // K % 64 == 0 (host check).  SPLITK: this workgroup's stages are s0 .. s0 + nk - 1 (a slice past the end has nk = 0 and
    // writes zeros)
    int s0 = 0, nk = p.K / BK;
    if constexpr (SPLITK) {
        const int per = (nk + p.splitk - 1) / p.splitk;
        s0 = (int)blockIdx.y * per;
        const int left = nk - s0;
        nk = left < 0 ? 0 : (left < per ? left : per);
    }

    const char* a0 = (const char*)p.a0;
    const char* a1 = (const char*)p.a1;
    const char* bw = (const char*)p.b;

    // ---- DMA pieces of this wave: A pieces pc = wave + q*NW (rows pc*8 .. +7 of the tile), B pieces likewise behind them.
    // lane -> row pc*8 + (lane>>3), physical chunk lane&7 = source chunk (lane&7) ^ swz(row); with NW == 4 the swizzle term
    // ((row>>1)&7 = (pc&1)*4 + (lane>>4)) is the same for every piece of a wave.
    const int r8 = lane >> 3;
    const unsigned sch16 = (unsigned)(((lane & 7) ^ (((wave & 1) << 2) | (r8 >> 1))) << 4);      // byte offset of the source chunk in its row
    unsigned a_row[QA], a_voff[QA], b_voff[QB];
    unsigned a_ok[GATHER ? QA : 1];                       // GATHER: bit t = tap t of this piece's row lies inside the plane
#pragma unroll
    for (int q = 0; q < QA; ++q) {
        int m = m0 + (wave + q * NW) * 8 + r8;
        m = m < p.M ? m : p.M - 1;                        // clamped rows feed accumulator rows that are never stored
        a_row[q] = (unsigned)m;
        if constexpr (!GATHER) {
            a_voff[q] = a_row[q] * ((unsigned)p.lda0 * 2u) + sch16;
        } else {
            const int hw = p.ho * p.wo;
            const int img = m / hw, rem = m - img * hw;
            const int oy = rem / p.wo, ox = rem - oy * p.wo;
            const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
            // byte offset of tap (0,0): may be "negative" (wraps; such a tap is masked, the valid ones wrap back)
            a_voff[q] = (unsigned)((img * p.hin + iy0) * p.win + ix0) * ((unsigned)p.lda0 * 2u) + sch16;
            unsigned ok = 0u;
#pragma unroll
            for (int t = 0; t < 9; ++t)
                if ((unsigned)(iy0 + t / 3) < (unsigned)p.hin && (unsigned)(ix0 + t % 3) < (unsigned)p.win) ok |= 1u << t;
            a_ok[q] = ok;
        }
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        int n = n0 + (wave + q * NW) * 8 + r8;
        n = n < p.N ? n : p.N - 1;
        b_voff[q] = (unsigned)n * ((unsigned)p.ldb * 2u) + sch16;
    }
    const int s_sw = p.c1 ? p.c0 / BK : nk;               // first stage that reads the second source (c0 % 64 == 0: host check)
    // One piece of the open window: A pieces 0 .. QA-1, then B pieces.  EVERY window issues its pieces, branch-free (a branch
    // would split the basic block and un-pin the MFMA / DMA interleave, and the per-wave VMEM counts stay the same in every
    // window so the counted waits are exact).  The window state -- source bases of the stage being fetched, destination
    // slot, and an all-ones / all-zeros lane-offset mask -- is set once per stage (open_window); past the last stage the
    // bases point at a 16-byte zero block and the mask is 0: the pieces then re-fetch those 16 bytes into the slot, which is
    // dead by then (tail) or never read (prologue), instead of operand rows.
    const char* zero = (const char*)g32_zero16;
    const char* w_abase = zero;
    const char* w_bbase = zero;
    unsigned w_msk = 0u;
    char* w_dst = i2i_smem;
    // GATHER: (tap, channel offset) of the NEXT window to open -- windows open in stage order, so the decode is a running
    // counter in scalar registers -- and of the open one: tap index (bit of a_ok) and byte offset of (tap pixel, channels)
    // Stage order of the gather (round 5): the NINE TAPS of a 64-channel slab run back to back (stage s = tap s % 9 of slab s / 9;
    // the contraction index of stage s is tap * cin + slab * 64 in the [tap][cin] weight layout).  With the taps outermost -- all
    // cin / 64 slabs of tap 0, then tap 1, ... -- a tile came back to the same pixels only after cin / 64 stages: 15 MB of A
    // operand per XCD between two visits at 960 channels, against 4 MiB of L2, so every tap re-fetched its pixels through the
    // fabric (PMC: 599 MB fetched per launch for 63 MB of input at 960 -> 320 @ 64 x 64, 3.8 TB/s: the launch was fabric-bound).
    int g_tap = 0, g_ky = 0, g_kx = 0, g_ci = 0;
    if constexpr (GATHER && SPLITK) {                     // the slice's first stage
        const int slab0 = s0 / 9;
        g_tap = s0 - slab0 * 9;
        g_ci = slab0 * BK;
        g_ky = g_tap / 3;
        g_kx = g_tap - g_ky * 3;
    }
    unsigned w_koff = 0u;                                 // GATHER: byte offset of the open window's stage in a weight row
    unsigned w_tap = 0u, w_aoff = 0u;
    auto open_window = [&](int s, int slot) __attribute__((always_inline)) {      // s < 0: dummy window
        const bool on = s >= 0;
        if constexpr (!GATHER) {
            const int sa = s >= s_sw ? s - s_sw : s;
            const char* ab = (s >= s_sw ? a1 : a0) + (size_t)(sa < 0 ? 0 : sa + s0) * (BK * 2);      // (SPLITK: one source, s_sw = nk)
            w_abase = on ? ab : zero;
        } else {
            w_tap = (unsigned)g_tap;
            w_aoff = (unsigned)((g_ky * p.win + g_kx) * p.lda0 + g_ci) * 2u;
            w_koff = (unsigned)(g_tap * p.c0 + g_ci) * 2u;
            g_tap += 1; g_kx += 1;                         // (selects, not branches: the K loop stays one basic block per half)
            const int wrap3 = g_kx == 3 ? 1 : 0;
            g_kx = wrap3 ? 0 : g_kx;
            g_ky += wrap3;
            const int wrap9 = g_tap == 9 ? 1 : 0;
            g_tap = wrap9 ? 0 : g_tap;
            g_ky = wrap9 ? 0 : g_ky;
            g_ci += wrap9 ? BK : 0;
        }
        if constexpr (GATHER) w_bbase = on ? bw + w_koff : zero;
        else w_bbase = on ? bw + (size_t)(s < 0 ? 0 : s + s0) * (BK * 2) : zero;
        w_msk = on ? 0xffffffffu : 0u;
        w_dst = i2i_smem + slot * STAGE;
        sopaque(w_msk);                                    // (the ksteps must not be specialised on the window kind)
    };
    auto dma_piece = [&](auto qc) __attribute__((always_inline)) {
        constexpr int q = decltype(qc)::value;
        if constexpr (q < QA) {
            if constexpr (!GATHER) {
                glds16_sv(w_abase, a_voff[q] & w_msk, w_dst + (wave + q * NW) * 1024);
            } else {
                const bool ok = ((a_ok[q] >> w_tap) & w_msk & 1u) != 0u;      // dummy windows (w_msk == 0) and taps past 8: zeros
                glds16(ok ? a0 + (a_voff[q] + w_aoff) : zero, w_dst + (wave + q * NW) * 1024);
            }
        } else glds16_sv(w_bbase, b_voff[q - QA] & w_msk, w_dst + (PA + wave + (q - QA) * NW) * 1024);
    };
    auto a_voff_for = [&](bool second) __attribute__((always_inline)) {
        const unsigned ldb2 = (unsigned)(second ? p.lda1 : p.lda0) * 2u;
#pragma unroll
        for (int q = 0; q < QA; ++q) a_voff[q] = a_row[q] * ldb2 + sch16;
    };

    f32x16 acc[FMW][FNW];                                 // (zeroed in the shadow of the prologue's DMAs, below)

    // ---- per-lane fragment read offsets: row l31 of a 32-row fragment, chunk (2*kk + lh) ^ swz(l31); fragment i / j and the
    // ring slot are added, the k16 step enters as ^ (kk << 5)
    const int rd_a = (wave * WTM + l31) * 128 + ((lh ^ gswz3(l31)) << 4);
    const int rd_b = BM * 128 + l31 * 128 + ((lh ^ gswz3(l31)) << 4);

    chunk_t xf[2][FMW], wf[2][FNW];
    // LNF: (sum, sum of squares) over the K elements this lane has seen of fragment row i (its k half: lanes l and l + 32 are combined
    // after the loop); trans: this column tile belongs to the transposed range (uniform over the workgroup)
    typedef T tx2 __attribute__((ext_vector_type(2)));
    float ls[LNF ? FMW : 1], lq[LNF ? FMW : 1];
#pragma unroll
    for (int i = 0; i < (LNF ? FMW : 1); ++i) { ls[i] = 0.f; lq[i] = 0.f; }
    tx2 ones2;
    ones2[0] = (T)1.0f; ones2[1] = (T)1.0f;

    // The batch of stage s + RING (OPB pieces per wave) is issued in the window that opens after barrier P_s: PPS + PREM
    // pieces in the last k16 step of stage s (window slot 0), PPS in each of the k16 steps 0 .. KQ-2 of stage s + 1.
    constexpr int PPS = OPB / KQ, PREM = OPB - PPS * KQ;
    auto win_lo = [](int w) constexpr { return w == 0 ? 0 : PPS + PREM + (w - 1) * PPS; };
    auto win_n = [](int w) constexpr { return w == 0 ? PPS + PREM : PPS; };

    // One k16 step kk of the stage whose read bases are (rbase_a, rbase_b): the fragment reads of the NEXT k16 step (step nkk
    // of the bases given) go out beside its first MFMAs, then the DMA pieces [plo, plo+pn) of the open window, one memory
    // operation per MFMA slot (pinned).
    auto kstep = [&](auto kkc, int rbase_a, int rbase_b, auto nkkc, auto ploc, auto pnc) __attribute__((always_inline)) {
        constexpr int kk = decltype(kkc)::value, nkk = decltype(nkkc)::value, plo = decltype(ploc)::value, pn = decltype(pnc)::value;
        constexpr int cur = kk & 1, nxt = cur ^ 1;
        const int ra = rbase_a ^ (nkk << 5), rb = rbase_b ^ (nkk << 5);
#pragma unroll
        for (int j = 0; j < FNW; ++j) wf[nxt][j] = *(const chunk_t*)(i2i_smem + (rb + j * 4096));
#pragma unroll
        for (int i = 0; i < FMW; ++i) xf[nxt][i] = *(const chunk_t*)(i2i_smem + (ra + i * 4096));
        static_for_g<pn>([&](auto qc) __attribute__((always_inline)) { dma_piece(icg<plo + decltype(qc)::value>{}); });
#pragma unroll
        for (int i = 0; i < FMW; ++i)
#pragma unroll
            for (int j = 0; j < FNW; ++j) {
                if constexpr (TR) acc[i][j] = mma32(xf[cur][i], wf[cur][j], acc[i][j]);
                else acc[i][j] = mma32(wf[cur][j], xf[cur][i], acc[i][j]);
            }
        if constexpr (LNF) {
#pragma unroll
            for (int i = 0; i < FMW; ++i)
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    tx2 d;
                    d[0] = xf[cur][i][2 * h]; d[1] = xf[cur][i][2 * h + 1];
                    ls[i] = g32_dot2(d, ones2, ls[i]);
                    lq[i] = g32_dot2(d, d, lq[i]);
                }
        }
        constexpr int TOT = NRD + pn, NVAL = LNF ? 8 * FMW : 0;
        static_for_g<NMM>([&](auto mc) __attribute__((always_inline)) {
            constexpr int m = decltype(mc)::value;
            constexpr int lo = (TOT * m) / NMM, hi = (TOT * (m + 1)) / NMM;
            constexpr int nr = (hi < NRD ? hi : NRD) - (lo < NRD ? lo : NRD), nd = (hi - lo) - nr;
            constexpr int nv = (NVAL * (m + 1)) / NMM - (NVAL * m) / NMM;
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if constexpr (nr > 0) __builtin_amdgcn_sched_group_barrier(0x100, nr, 0);
            if constexpr (nd > 0) __builtin_amdgcn_sched_group_barrier(0x010, nd, 0);
            if constexpr (nv > 0) __builtin_amdgcn_sched_group_barrier(0x002, nv, 0);
        });
    };

    // ---- prologue: stages 0 .. RING-2 in flight and the window of stage RING-1 opened exactly as the loop would have left
    // it behind a barrier "P_-1" (its first PPS + PREM pieces issued here, the rest beside k16 steps 0 .. 2 of stage 0), so
    // that the VMEM issue stream is the steady-state one from the first barrier on.  Stages past the end are dummy windows
    // into their own (never read) slots.  Wait for stage 0, publish, first fragments.
#pragma unroll
    for (int t = 0; t < RING; ++t) {
        if (t == s_sw && t < nk) a_voff_for(true);
        open_window(t < nk ? t : -1, t);
        if (t < RING - 1) static_for_g<OPB>([&](auto qc) __attribute__((always_inline)) { dma_piece(qc); });
        else static_for_g<win_n(0)>([&](auto qc) __attribute__((always_inline)) { dma_piece(qc); });
    }
    // the accumulator writes, pinned in front of the wait (as plain constants hipcc re-materialises them behind the barrier, in front of
    // the first MFMA: 160 instructions of a tile whose K loop may be 200 MFMAs)
#pragma unroll
    for (int i = 0; i < FMW; ++i)
#pragma unroll
        for (int j = 0; j < FNW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = g32_acc_zero();
    wait_vmcnt<(RING - 2) * OPB + win_n(0)>();
    lds_barrier();
#pragma unroll
    for (int j = 0; j < FNW; ++j) wf[0][j] = *(const chunk_t*)(i2i_smem + (rd_b + j * 4096));
#pragma unroll
    for (int i = 0; i < FMW; ++i) xf[0][i] = *(const chunk_t*)(i2i_smem + (rd_a + i * 4096));

    // ---- K loop.  Stage s sits in slot s % RING.  Barrier P_s stands before the last k16 step of stage s: every wave has
    // then issued AND completed its last fragment reads of the slot (they run one k16 step ahead), so the slot is free for
    // stage s + RING; P_s also publishes stage s + 1 (each wave waits for its own pieces with a counted vmcnt just before).
    // VMEM issue order per wave: [W_0] [W_1] ... (OPB pieces each); at P_s the batch W_{s+RING-1} is complete and everything
    // up to W_{s+1} must have landed: exactly the RING-2 batches behind it may still fly.
    {
        int cur = 0;
        for (int s = 0; s < nk; ++s) {
            const int nxt = cur + 1 == RING ? 0 : cur + 1;
            const int ca = rd_a + cur * STAGE, cb = rd_b + cur * STAGE;
            kstep(icg<0>{}, ca, cb, icg<1>{}, icg<win_lo(1)>{}, icg<win_n(1)>{});
            kstep(icg<1>{}, ca, cb, icg<2>{}, icg<win_lo(2)>{}, icg<win_n(2)>{});
            kstep(icg<2>{}, ca, cb, icg<3>{}, icg<win_lo(3)>{}, icg<win_n(3)>{});
            __builtin_amdgcn_sched_barrier(0);
            wait_vmcnt<(RING - 2) * OPB>();
            lds_barrier();
            {                                              // the window of stage s + RING opens: slot `cur` is free now
                const int pend = s + RING < nk ? s + RING : -1;
                if (pend == s_sw && pend >= 0) a_voff_for(true);
                open_window(pend, cur);
            }
            const int na = rd_a + nxt * STAGE, nb = rd_b + nxt * STAGE;
            __builtin_amdgcn_sched_barrier(0);
            kstep(icg<3>{}, na, nb, icg<0>{}, icg<win_lo(0)>{}, icg<win_n(0)>{});
            __builtin_amdgcn_sched_barrier(0);
            cur = nxt;
        }
    }
    wait_vmcnt<0>();                                       // (tail windows: dummy pieces still in flight)

    // ---- LNF: mean / rstd of the rows this lane's fragments cover (row l31 of fragment i: the two k halves live in lanes l and l + 32)
    float ra[LNF ? FMW : 1], rb[LNF ? FMW : 1];           // rstd, -rstd * mean
    if constexpr (LNF) {
        const float invk = 1.0f / (float)p.K;
#pragma unroll
        for (int i = 0; i < FMW; ++i) {
            float s0 = ls[i], s1 = ls[i], q0 = lq[i], q1 = lq[i];
            half_swap(s0, s1);                             // lanes 0-31: (own, partner's); lanes 32-63: (partner's, own)
            half_swap(q0, q1);
            const float mu = (s0 + s1) * invk;
            const float var = fmaxf((q0 + q1) * invk - mu * mu, 0.f);
            ra[i] = rsqrtf(var + p.ln_eps);
            rb[i] = -ra[i] * mu;
        }
        if constexpr (TR) {
            // ---- the transposed range: acc[i][j][r] = token row m0 + wave*WTM + i*32 + 8*(r>>2) + 4*lh + (r&3), column n0 + j*32 + l31.
            // Every lane needs (rstd, -rstd*mu) of the 8 token rows of a register-quad pair: the table goes through LDS behind the
            // operand ring (wave-private rows; the LDS queue of a wave is in order).
            float* st = (float*)(i2i_smem + RING * STAGE) + (wave * WTM) * 2;
            if (lh == 0) {
#pragma unroll
                for (int i = 0; i < FMW; ++i) { st[(i * 32 + l31) * 2] = ra[i]; st[(i * 32 + l31) * 2 + 1] = rb[i]; }
            }
            g32_wave_sync();
            T* __restrict__ out2 = (T*)p.c2;
            typedef T tx8 __attribute__((ext_vector_type(8)));
#pragma unroll
            for (int j = 0; j < FNW; ++j) {
                const int n = n0 + j * 32 + l31;
                const bool nok = n < p.N;
                const float csn = p.ln_cs[nok ? n : 0], bn = p.bias[nok ? n : 0];
#pragma unroll
                for (int i = 0; i < FMW; ++i)
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        f32x4 qa, qb;
#pragma unroll
                        for (int r = 0; r < 4; ++r) { qa[r] = acc[i][j][8 * pr + r]; qb[r] = acc[i][j][8 * pr + 4 + r]; }
                        float v[8];
                        widen_pair(qa, qb, v);             // the lane's 8 consecutive tokens 16*pr + 8*lh .. +7 of fragment i
                        const int mr = i * 32 + 16 * pr + 8 * lh;
                        const int m = m0 + wave * WTM + mr;
                        const f32x4 t0 = *(const f32x4*)(st + mr * 2), t1 = *(const f32x4*)(st + mr * 2 + 4),
                                    t2 = *(const f32x4*)(st + mr * 2 + 8), t3 = *(const f32x4*)(st + mr * 2 + 12);
                        const float a8[8] = {t0[0], t0[2], t1[0], t1[2], t2[0], t2[2], t3[0], t3[2]};
                        const float b8[8] = {t0[1], t0[3], t1[1], t1[3], t2[1], t2[3], t3[1], t3[3]};
                        tx8 o;
#pragma unroll
                        for (int r = 0; r < 8; ++r) o[r] = from_f32<T>(__builtin_fmaf(a8[r], v[r], __builtin_fmaf(b8[r], csn, bn)));
                        if (nok && m < p.M) *(tx8*)(out2 + (int64_t)(n - p.n_trans) * p.ldc2 + m) = o;      // M % 8 == 0 (host check)
                    }
            }
            return;
        }
    }

    // ---- epilogue.  acc[i][j][r]: row m0 + wave*WTM + i*32 + l31, column n0 + j*32 + 8*(r>>2) + 4*lh + (r&3).  Register quads
    // (2*pr, 2*pr+1) are half-exchanged between lanes l and l+32 (widen_pair): the lane then owns the 8 consecutive columns
    // j*32 + 16*pr + 8*lh .. +7 of its row: 16-byte residual loads and stores.
    if constexpr (SPLITK) {
        // raw fp32 partial sums: a register quad is 4 consecutive columns of one row (16-byte stores); N % 8 == 0 (host check)
        float* __restrict__ ws = (float*)p.ws + (int64_t)blockIdx.y * p.M * p.N;
#pragma unroll
        for (int i = 0; i < FMW; ++i) {
            const int m = m0 + wave * WTM + i * 32 + l31;
#pragma unroll
            for (int j = 0; j < FNW; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = n0 + j * 32 + 8 * q + 4 * lh;
                    if (m < p.M && n < p.N) *(f32x4*)(ws + (int64_t)m * p.N + n) = f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                }
        }
        return;
    }
    typedef T tx8 __attribute__((ext_vector_type(8)));
    const T* __restrict__ res = (const T*)p.res;
    T* __restrict__ out = (T*)p.c;
    constexpr int NV = STATS ? FNW * 8 : 1;              // STATS: (sum, sum of squares) of the lane's stored values per 4-channel quad
    auto stats_out = [&](float (&gacc)[NV]) __attribute__((always_inline)) {
        // ---- GroupNorm partial sums of the stored tile (the next layer's norm): lane (l31, lh) holds NV sums over ITS row(s);
        // rows -> wave through an LDS transpose (lane t adds value t % 32 of the 32 lanes of half t / 32: fixed order), waves ->
        // workgroup -> one slot per (image, row tile, group).  The operand ring is dead: every wave's DMA has landed (the
        // wait above) and the barrier below says every wave is done reading it.
        if constexpr (STATS) {
            static_assert(!STATS || NV == 32, "one value per lane and half: FNW == 4");
            if (p.gn_part) {
                constexpr int LST = NV + 4;                   // lane stride in floats: 16-byte writes of 16 lanes hit 16 distinct bank quads
                float* st = (float*)i2i_smem;                 // [NW][64][LST]
                float* st2 = st + NW * 64 * LST;              // [NW][32 quads][2]
                lds_barrier();
#pragma unroll
                for (int v = 0; v < NV; v += 4)
                    *(f32x4*)(st + (wave * 64 + lane) * LST + v) = f32x4{gacc[v], gacc[v + 1], gacc[v + 2], gacc[v + 3]};
                lds_barrier();
                {
                    const int half = lane >> 5, v = lane & 31;
                    float tot = 0.f;
                    for (int l = 0; l < 32; ++l) tot += st[(wave * 64 + half * 32 + l) * LST + v];
                    const int jpr = v >> 2, h = (v >> 1) & 1, sq = v & 1;        // v = ((j*2 + pr)*2 + h)*2 + sq
                    const int quad = jpr * 4 + half * 2 + h;                      // column quad of the tile: j*8 + pr*4 + lh*2 + h
                    st2[(wave * 32 + quad) * 2 + sq] = tot;
                }
                lds_barrier();
                const int groups = p.gn_part_groups, cpg = p.N / groups, ng_tile = BN / cpg;
                const int g = n0 / cpg + tid;
                if (tid < ng_tile && g < groups) {
                    const int q0 = (tid * cpg) >> 2, nq = cpg >> 2;
                    float S = 0.f, Q = 0.f;
                    for (int w = 0; w < NW; ++w)
                        for (int q = q0; q < q0 + nq; ++q) { S += st2[(w * 32 + q) * 2]; Q += st2[(w * 32 + q) * 2 + 1]; }
                    const int hw = p.ho * p.wo, parts = hw / BM;
                    const int img = m0 / hw, part = (m0 - img * hw) / BM;
                    float* o2 = p.gn_part + (((int64_t)img * parts + part) * groups + g) * 2;
                    o2[0] = S;
                    o2[1] = Q;
                }
            }
        }
    };
#ifndef G32_EPI_PIPE
#define G32_EPI_PIPE 1
#endif
    // ---- The pipelined epilogue (G32_EPI_PIPE=0: the first form below, the A/B baseline).  The first form fetched each column block's
    // bias and residual at its use, behind runtime branches: load, vmcnt(0) -- which also waits for the previous block's STORES --, FMAs,
    // store, 2 x FNW times per tile: ten serial HBM round trips on a 256 x 160 tile whose K loop (K = 320) is 200 MFMAs.  Here the bias
    // and the residual of block t + 1 are requested before block t's stores (branch-free: hipcc's own vmcnt counts are then exact and do not
    // wait for a store), and tiles inside M x N (all of them on this model's shapes) take an instantiation without predicates.
    constexpr int NBLK = GEGLU ? FNW : 2 * FNW;          // column blocks of 8 OUTPUT columns per lane: (j, pr), or j for GEGLU
    auto pipe = [&](auto fullc, auto resc) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(fullc)::value != 0, RES = decltype(resc)::value != 0;
        const int mrow = m0 + wave * WTM + l31;           // + i*32
        auto col_in = [&](auto tc) __attribute__((always_inline)) {      // first packed (input) column of block t's bias
            constexpr int t = decltype(tc)::value;
            if constexpr (GEGLU) return n0 + t * 32; else return n0 + (t >> 1) * 32 + 16 * (t & 1) + 8 * lh;
        };
        auto col_out = [&](auto tc) __attribute__((always_inline)) {     // first output column of the lane's 8 results of block t
            constexpr int t = decltype(tc)::value;
            if constexpr (GEGLU) return (n0 + t * 32) / 2 + 8 * lh; else return n0 + (t >> 1) * 32 + 16 * (t & 1) + 8 * lh;
        };
        auto col_ok = [&](auto tc) __attribute__((always_inline)) {
            if constexpr (FULL) return true;
            else if constexpr (GEGLU) return col_in(tc) + 32 <= p.N;      // N % 32 == 0 (host check)
            else return col_in(tc) < p.N;                                  // N % 8 == 0: a chunk is inside or outside as a whole
        };
        // bias of block t (4 x 16 bytes for GEGLU: value and gate quads; 2 x 16 bytes otherwise), double-buffered like the residual and
        // branch-free: without a bias the loads read the first bytes of the weights and the values are selected away
        constexpr int NBQ = GEGLU ? 4 : 2;
        f32x4 bb[2][NBQ];
        f32x4 cb[LNF ? 2 : 1][LNF ? NBQ : 1];              // LNF: the column sums of the folded weights, like the bias
        const bool has_bias = GEGLU ? p.bias != nullptr : p.bias_mode == 1;
        const float* const bsrc = has_bias ? p.bias : (const float*)p.b;
        auto bload = [&](auto tc) __attribute__((always_inline)) {
            constexpr int t = decltype(tc)::value;
            if constexpr (t < NBLK) {
                const int nb = (has_bias && col_ok(tc)) ? col_in(tc) : 0;      // (clamped: lanes outside N never store)
#pragma unroll
                for (int q = 0; q < NBQ; ++q) {
                    const f32x4 b = *(const f32x4*)(bsrc + nb + (GEGLU ? 8 * q + 4 * lh : 4 * q));
#pragma unroll
                    for (int r = 0; r < 4; ++r) bb[t & 1][q][r] = has_bias ? b[r] : 0.f;
                    if constexpr (LNF) cb[t & 1][q] = *(const f32x4*)(p.ln_cs + (col_ok(tc) ? col_in(tc) : 0) + (GEGLU ? 8 * q + 4 * lh : 4 * q));
                }
            }
        };
        chunk_t rr[2][FMW];
        auto rload = [&](auto tc) __attribute__((always_inline)) {
            constexpr int t = decltype(tc)::value;
            if constexpr (RES && t < NBLK) {
                const bool nok = col_ok(tc);
                const int no = col_out(tc);
#pragma unroll
                for (int i = 0; i < FMW; ++i) {
                    const int m = mrow + i * 32;
                    const bool ok = FULL || (nok && m < p.M);
                    rr[t & 1][i] = *(const chunk_t*)(res + (ok ? (int64_t)m * p.ldr + no : 0));
                }
            }
        };
        float gacc[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) gacc[v] = 0.f;
        const tx2 ones = ones2;
        bload(icg<0>{});
        rload(icg<0>{});
        static_for_g<NBLK>([&](auto tc) __attribute__((always_inline)) {
            constexpr int t = decltype(tc)::value;
            bload(icg<t + 1>{});                              // the next block's bias and residual: issued before this block's stores
            rload(icg<t + 1>{});
            const bool nok = col_ok(tc);
            const int no = col_out(tc);
#pragma unroll
            for (int i = 0; i < FMW; ++i) {
                f32x4 qa, qb;
                if constexpr (GEGLU) {
                    // weight rows (and bias) are interleaved per 32 as [16 value rows | 16 gate rows] (packer.geglu_linear): a fragment's
                    // registers 0..7 are values and 8..15 the gates of the SAME 16 output columns: out = value * gelu(gate), lane-local
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float a, g;
                            if constexpr (LNF) {           // rstd * acc - rstd * mu * colsum + bias'   (alpha = 1: host check)
                                a = __builtin_fmaf(ra[i], acc[i][t][4 * q + r], __builtin_fmaf(rb[i], cb[t & 1][q][r], bb[t & 1][q][r]));
                                g = __builtin_fmaf(ra[i], acc[i][t][8 + 4 * q + r], __builtin_fmaf(rb[i], cb[t & 1][2 + q][r], bb[t & 1][2 + q][r]));
                            } else {
                                a = __builtin_fmaf(p.alpha, acc[i][t][4 * q + r], bb[t & 1][q][r]);
                                g = __builtin_fmaf(p.alpha, acc[i][t][8 + 4 * q + r], bb[t & 1][2 + q][r]);
                            }
                            const float o = a * gelu_erf_f(g);
                            if (q == 0) qa[r] = o; else qb[r] = o;
                        }
                } else {
                    constexpr int j = t >> 1, pr = t & 1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { qa[r] = acc[i][j][8 * pr + r]; qb[r] = acc[i][j][8 * pr + 4 + r]; }
                }
                float v[8];
                widen_pair(qa, qb, v);                     // wave-wide: before any lane drops out
                const int m = mrow + i * 32;
                if (!FULL && (!nok || m >= p.M)) continue;
                if constexpr (!GEGLU) {
                    if constexpr (LNF) {                   // (lanes l and l + 32 hold the same row: ra / rb survive the exchange)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            v[r] = __builtin_fmaf(ra[i], v[r], __builtin_fmaf(rb[i], cb[t & 1][0][r], bb[t & 1][0][r]));
                            v[4 + r] = __builtin_fmaf(ra[i], v[4 + r], __builtin_fmaf(rb[i], cb[t & 1][1][r], bb[t & 1][1][r]));
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { v[r] = __builtin_fmaf(p.alpha, v[r], bb[t & 1][0][r]); v[4 + r] = __builtin_fmaf(p.alpha, v[4 + r], bb[t & 1][1][r]); }
                    }
                }
                if constexpr (RES) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] += to_f32<T>(rr[t & 1][i][r]);
                }
                tx8 o;
#pragma unroll
                for (int r = 0; r < 8; ++r) o[r] = from_f32<T>(v[r]);
                *(tx8*)(out + (int64_t)m * p.ldc + no) = o;
                if constexpr (STATS) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        tx2 d0, d1;
                        d0[0] = o[4 * h]; d0[1] = o[4 * h + 1]; d1[0] = o[4 * h + 2]; d1[1] = o[4 * h + 3];
                        const int vi = (t * 2 + h) * 2;      // ((j*2 + pr)*2 + h)*2
                        gacc[vi] = g32_dot2(d0, ones, gacc[vi]);         gacc[vi] = g32_dot2(d1, ones, gacc[vi]);
                        gacc[vi + 1] = g32_dot2(d0, d0, gacc[vi + 1]);   gacc[vi + 1] = g32_dot2(d1, d1, gacc[vi + 1]);
                    }
                }
            }
        });
        if constexpr (STATS) stats_out(gacc);
    };
    static_assert(G32_EPI_PIPE || !LNF, "the LayerNorm fold lives in the pipelined epilogue only");
    if (G32_EPI_PIPE) {
        const bool full = m0 + BM <= p.M && n0 + BN <= p.N;      // (uniform)
        if (full) { if (res) pipe(icg<1>{}, icg<1>{}); else pipe(icg<1>{}, icg<0>{}); }
        else      { if (res) pipe(icg<0>{}, icg<1>{}); else pipe(icg<0>{}, icg<0>{}); }
        return;
    }
    if constexpr (!GEGLU) {
        // STATS: (sum, sum of squares) of the lane's stored values per 4-channel quad: index ((j*2 + pr)*2 + h)*2 + {0, 1}
        float gacc[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) gacc[v] = 0.f;
        const tx2 ones = ones2;
#pragma unroll
        for (int j = 0; j < FNW; ++j) {
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const int n = n0 + j * 32 + 16 * pr + 8 * lh;
                const bool nok = n < p.N;                 // N % 8 == 0 (host check): a chunk is inside or outside as a whole
                f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
                if (p.bias_mode == 1 && nok) {
                    b0 = *(const f32x4*)(p.bias + n);
                    b1 = *(const f32x4*)(p.bias + n + 4);
                }
                chunk_t rr[FMW];
                if (res) {
#pragma unroll
                    for (int i = 0; i < FMW; ++i) {
                        const int m = m0 + wave * WTM + i * 32 + l31;
                        const bool ok = nok && m < p.M;
                        rr[i] = *(const chunk_t*)(res + (ok ? (int64_t)m * p.ldr + n : 0));
                    }
                }
#pragma unroll
                for (int i = 0; i < FMW; ++i) {
                    f32x4 qa, qb;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { qa[r] = acc[i][j][8 * pr + r]; qb[r] = acc[i][j][8 * pr + 4 + r]; }
                    float v[8];
                    widen_pair(qa, qb, v);                 // wave-wide: before any lane drops out
                    const int m = m0 + wave * WTM + i * 32 + l31;
                    if (!nok || m >= p.M) continue;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { v[r] = __builtin_fmaf(p.alpha, v[r], b0[r]); v[4 + r] = __builtin_fmaf(p.alpha, v[4 + r], b1[r]); }
                    if (res) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] += to_f32<T>(rr[i][r]);
                    }
                    tx8 o;
#pragma unroll
                    for (int r = 0; r < 8; ++r) o[r] = from_f32<T>(v[r]);
                    *(tx8*)(out + (int64_t)m * p.ldc + n) = o;
                    if constexpr (STATS) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            tx2 d0, d1;
                            d0[0] = o[4 * h]; d0[1] = o[4 * h + 1]; d1[0] = o[4 * h + 2]; d1[1] = o[4 * h + 3];
                            const int vi = ((j * 2 + pr) * 2 + h) * 2;
                            gacc[vi] = g32_dot2(d0, ones, gacc[vi]);         gacc[vi] = g32_dot2(d1, ones, gacc[vi]);
                            gacc[vi + 1] = g32_dot2(d0, d0, gacc[vi + 1]);   gacc[vi + 1] = g32_dot2(d1, d1, gacc[vi + 1]);
                        }
                    }
                }
            }
        }
        if constexpr (STATS) stats_out(gacc);
    } else {
        // GEGLU: weight rows (and bias) are interleaved per 32 as [16 value rows | 16 gate rows] (packer.geglu_linear), so a
        // fragment's registers 0..7 are values and 8..15 the gates of the SAME 16 output columns: out = value * gelu(gate),
        // lane-local; output column block = (n0 + j*32) / 2.
#pragma unroll
        for (int j = 0; j < FNW; ++j) {
            const int nv = n0 + j * 32;                   // packed column of this fragment's first value
            const bool nok = nv + 32 <= p.N;              // N % 32 == 0 (host check)
            f32x4 bv[2], bg[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                bv[q] = f32x4{0.f, 0.f, 0.f, 0.f}; bg[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (p.bias && nok) {
                    bv[q] = *(const f32x4*)(p.bias + nv + 8 * q + 4 * lh);
                    bg[q] = *(const f32x4*)(p.bias + nv + 16 + 8 * q + 4 * lh);
                }
            }
            const int no = nv / 2 + 8 * lh;               // output column of the lane's 8 results after the exchange
            chunk_t rr[FMW];
            if (res) {
#pragma unroll
                for (int i = 0; i < FMW; ++i) {
                    const int m = m0 + wave * WTM + i * 32 + l31;
                    const bool ok = nok && m < p.M;
                    rr[i] = *(const chunk_t*)(res + (ok ? (int64_t)m * p.ldr + no : 0));
                }
            }
#pragma unroll
            for (int i = 0; i < FMW; ++i) {
                f32x4 qa, qb;
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float a = __builtin_fmaf(p.alpha, acc[i][j][4 * q + r], bv[q][r]);
                        const float g = __builtin_fmaf(p.alpha, acc[i][j][8 + 4 * q + r], bg[q][r]);
                        const float o = a * gelu_erf_f(g);
                        if (q == 0) qa[r] = o; else qb[r] = o;
                    }
                float v[8];
                widen_pair(qa, qb, v);
                const int m = m0 + wave * WTM + i * 32 + l31;
                if (!nok || m >= p.M) continue;
                if (res) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] += to_f32<T>(rr[i][r]);
                }
                tx8 o;
#pragma unroll
                for (int r = 0; r < 8; ++r) o[r] = from_f32<T>(v[r]);
                *(tx8*)(out + (int64_t)m * p.ldc + no) = o;
            }
        }
    }
    };      // body
    if constexpr (LNF && !GEGLU) {
        if (trans) body(icg<1>{}); else body(icg<0>{});
    } else body(icg<0>{});
}

// tile ids 50..59 (i2i_igemm_params.tile): 50 = auto among the configurations
//   51: 256 x 160 (64 x 160 per wave)   52: 128 x 160 (32 x 160)   53: 256 x 128 (64 x 128)   54: 128 x 128 (32 x 128)
//   55 / 56: 128 x 160 / 128 x 128 with a 2-deep ring (72 / 64 KiB of LDS, under 256 registers): TWO workgroups per CU, so that
//            one's epilogue (the GEGLU gate activation is ~1.4x the MFMA time of a K = 320 tile) runs beside the other's K loop
struct G32Cfg { int bm, bn; };
G32Cfg g32_geometry(int cfg) {
    switch (cfg) {
        case 51: return {256, 160};
        case 52: case 55: return {128, 160};
        case 53: return {256, 128};
        default: return {128, 128};
    }
}
long g32_tiles(const i2i_igemm_params& p, int cfg) {
    const G32Cfg g = g32_geometry(cfg);
    return (long)((p.M + g.bm - 1) / g.bm) * ((p.N + g.bn - 1) / g.bn);
}
// auto (measured per shape, same box: profiles/r4b_bench_ops_gemm_w32_vs_dma.log, r4e_bench_ops_*.log):
//   * GEGLU with K < 1280: the gate activation costs more issue time than the tile's MFMAs -> the two-workgroups-per-CU
//     form (55), whose epilogues run beside the other workgroup's K loop (320 -> 2560 @ 32768 rows: 0.113 -> 0.097 ms);
//   * the tall tile (256 rows) when it still gives a full round of workgroups;
//   * otherwise the 128-row tile whose column width (160 / 128, whichever divide N) fills more of its last round
//     (5120 -> 1280 @ 2048 rows: 160 tiles of 128 x 128 beat 128 tiles of 128 x 160, 0.045 vs 0.050 ms).
int g32_cfg(const i2i_igemm_params& p) {
    if (p.tile >= 51 && p.tile <= 56) return p.tile;
    const bool w160 = p.N % 160 == 0, w128 = p.N % 128 == 0;
    if (p.geglu && w160 && p.K < 1280) return 55;
    // (round 6, measured: the plain short-K projections -- to_out, proj_in / proj_out, to_q: K = 320 / 640 -- on the same two-workgroups-per-CU
    // form: -0.06 ... -0.13 +- 0.07 ms per step, per-op times unchanged (20.4 vs 20.7 us for 32768 x 320 x 320).  They are not tile-bound: 63 MB
    // of operand + residual + output per 6.7 GFLOP is 3.1 TB/s, i.e. these launches sit on the HBM roofline, not the MFMA one:
    // profiles/r6j_ab_bs8_g32_shortk.log)
    const int tall = w160 ? 51 : 53;
    if (g32_tiles(p, tall) >= 256) return tall;
    auto fill = [&](int cfg) { const long t = g32_tiles(p, cfg); return (double)t / (double)(((t + 255) / 256) * 256); };
    if (w160 && w128) return fill(54) > fill(52) ? 54 : 52;
    return w160 ? 52 : 54;
}

template <typename T, int FMW, int FNW, int RING = 3>
int launch_g32(const i2i_igemm_params& p, hipStream_t s) {
    constexpr int BM = G32_NW * 32 * FMW, BN = 32 * FNW;
    const unsigned tiles = (unsigned)(((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN));
    const size_t smem = (size_t)RING * (BM + BN) * 128 + (p.ln_cs && p.n_trans > 0 ? (size_t)BM * 8 : 0);      // (+ the row-statistics table of the transposed tiles)
    const dim3 g(tiles), b(G32_NW * 64);
    const bool gather = p.ks == 3, stats = p.gn_part != nullptr;
    if (p.ln_cs) {
        if (gather || stats || p.splitk > 1) return i2i::fail(I2I_ERR_BAD_ARG, "gemm_w32: the LayerNorm fold is a plain GEMM epilogue (no gather / split-K / statistics)");
        if (p.geglu) hipLaunchKernelGGL((gemm_w32_kernel<T, FMW, FNW, RING, true, false, false, false, true>), g, b, smem, s, p);
        else hipLaunchKernelGGL((gemm_w32_kernel<T, FMW, FNW, RING, false, false, false, false, true>), g, b, smem, s, p);
        return i2i::check_launch("gemm_w32<ln>");
    }
    // (the gather, split-K and the statistics epilogue exist for the 3-deep ring only, the statistics for 128-column tiles only:
    // gemm_w32_eligible / gemm_w32_gn_parts admit nothing else)
    if constexpr (RING == 3) {
        if (p.splitk > 1) {
            const dim3 gs(tiles, (unsigned)p.splitk);
            if (gather) hipLaunchKernelGGL((gemm_w32_kernel<T, FMW, FNW, RING, false, true, false, true>), gs, b, smem, s, p);
            else hipLaunchKernelGGL((gemm_w32_kernel<T, FMW, FNW, RING, false, false, false, true>), gs, b, smem, s, p);
            const int rc = i2i::check_launch("gemm_w32<splitk>");
            return rc != I2I_OK ? rc : i2i::splitk_reduce(p, Elem<T>::DT, s);
        }
        if constexpr (FNW == 4) {
            if (gather && stats) { hipLaunchKernelGGL((gemm_w32_kernel<T, FMW, FNW, RING, false, true, true>), g, b, smem, s, p); return i2i::check_launch("gemm_w32<conv,stats>"); }
            if (stats) { hipLaunchKernelGGL((gemm_w32_kernel<T, FMW, FNW, RING, false, false, true>), g, b, smem, s, p); return i2i::check_launch("gemm_w32<stats>"); }
        }
        if (gather) { hipLaunchKernelGGL((gemm_w32_kernel<T, FMW, FNW, RING, false, true, false>), g, b, smem, s, p); return i2i::check_launch("gemm_w32<conv>"); }
    }
    if (gather || stats || p.splitk > 1) return i2i::fail(I2I_ERR_BAD_ARG, "gemm_w32: this tile configuration has no 3x3 gather / split-K / statistics epilogue");
    if (p.geglu) hipLaunchKernelGGL((gemm_w32_kernel<T, FMW, FNW, RING, true>), g, b, smem, s, p);
    else hipLaunchKernelGGL((gemm_w32_kernel<T, FMW, FNW, RING, false>), g, b, smem, s, p);
    return i2i::check_launch("gemm_w32");
}

template <typename T>
int launch_g32_t(const i2i_igemm_params& p, hipStream_t s) {
    switch (g32_cfg(p)) {
        case 51: return launch_g32<T, 2, 5>(p, s);
        case 52: return launch_g32<T, 1, 5>(p, s);
        case 53: return launch_g32<T, 2, 4>(p, s);
        case 54: return launch_g32<T, 1, 4>(p, s);
        case 55: return launch_g32<T, 1, 5, 2>(p, s);
        case 56: return launch_g32<T, 1, 4, 2>(p, s);
    }
    return i2i::fail(I2I_ERR_BAD_ARG, "gemm_w32: unknown tile config %d", p.tile);
}

}  // namespace

// ---- Two translation units (W32_PART = 0 / 1, csrc/build.py and tests/emu/build_emu.py; without it: everything): one per 16-bit type, so
// that the ~80 instantiations compile side by side.  Part 0 also holds the eligibility rules and the dispatcher.
#if !defined(W32_PART) || W32_PART == 1
namespace i2i {
int gemm_w32_f16(const i2i_igemm_params& p, hipStream_t s) { return launch_g32_t<_Float16>(p, s); }
}  // namespace i2i
#endif
#if !defined(W32_PART) || W32_PART == 0
namespace i2i {
int gemm_w32_f16(const i2i_igemm_params& p, hipStream_t s);
// GroupNorm partial-sum slots per image the wide GEMM writes for this op (0 = it cannot): the 128-column tiles (53 / 54), row
// tiles that do not straddle images, channels-per-group a multiple of 4 that divides the 128 columns of a tile.
static int g32_gn_parts(const i2i_igemm_params& p, int groups) {
    if (groups < 1 || p.N % groups || p.geglu || p.splitk > 1) return 0;      // (split-K: the reduce launch has no statistics)
    const int cfg = g32_cfg(p);
    if (cfg != 53 && cfg != 54) return 0;
    const int bm = g32_geometry(cfg).bm, cpg = p.N / groups, hw = p.ho * p.wo;
    if (cpg % 4 || 128 % cpg || hw % bm) return 0;
    return hw / bm;
}
// What the wide GEMM takes: a 16-bit contraction with one z -- a plain one (1x1, stride 1, up to two channel-concatenated
// sources) or the im2col view of a 3x3 convolution (stride 1 / 2, padding 0 / 1, one source, no upsample) -- K and the first
// source in whole 64-wide stages, 16-byte rows everywhere, per-column bias or none, optional residual / GEGLU (plain only) /
// GroupNorm partial sums of the output (g32_gn_parts); offsets fit 32 bits.
bool gemm_w32_eligible(const i2i_igemm_params& p, int dtype) {
    if (dtype != I2I_BF16 && dtype != I2I_F16) return false;
    if (p.ks != 1 && p.ks != 3) return false;
    const bool gather = p.ks == 3;
    if (!gather && (p.stride != 1 || p.pad != 0)) return false;
    if (gather && (p.stride < 1 || p.stride > 2 || p.pad < 0 || p.pad > 1 || p.c1 || p.a1 || p.geglu || p.tile == 55 || p.tile == 56)) return false;
    if (p.ups != 0 || p.up_h || p.up_w || p.subpix || p.k2_a) return false;
    if (p.zcount > 1 || p.gn_ss || p.act || p.act_out || p.out_f32 || p.bias_mode == 2) return false;
    // split-K: only when a caller names one of the 3-deep-ring tiles (tile 0 keeps split-K ops on the LDS-DMA igemm)
    if (p.splitk > 1 && (!p.ws || (((uintptr_t)p.ws) & 15) || p.c1 || p.geglu || p.gn_part || p.tile < 51 || p.tile > 54 || p.splitk > p.K / G32_BK)) return false;
    const int cin = p.c0 + p.c1;
    if (p.K != p.ks * p.ks * cin || cin % G32_BK || cin < G32_BK || (p.c1 && p.c0 % G32_BK)) return false;
    if (p.lda0 % 8 || (p.a1 && p.lda1 % 8) || p.ldb % 8 || p.ldc % 8 || (p.res && p.ldr % 8)) return false;
    if (((uintptr_t)p.a0 | (uintptr_t)p.a1 | (uintptr_t)p.b | (uintptr_t)p.c | (uintptr_t)p.res) & 15) return false;
    if (p.bias && (p.geglu || p.bias_mode == 1) && ((uintptr_t)p.bias & 15)) return false;      // (both epilogues read the bias as 16-byte vectors)
    if (p.M < 1 || p.N < 32 || p.N % 8) return false;
    if (p.geglu && (p.N % 32 || p.bias_mode == 2)) return false;
    if (p.gn_part && g32_gn_parts(p, p.gn_part_groups) == 0) return false;
    if (p.ln_cs) {      // LayerNorm fold: plain one-source GEMM, per-column bias' present, alpha 1, no slices / statistics
        if (gather || p.c1 || p.a1 || p.splitk > 1 || p.gn_part || p.alpha != 1.0f || !p.bias || p.bias_mode != 1 || (((uintptr_t)p.ln_cs) & 15)) return false;
        if (p.n_trans) {
            const int bn = g32_geometry(g32_cfg(p)).bn;
            if (p.geglu || p.res || p.n_trans < 0 || p.n_trans >= p.N || p.n_trans % bn || !p.c2 || (((uintptr_t)p.c2) & 15) || p.ldc2 % 8 || p.M % 8) return false;
        }
    } else if (p.n_trans || p.c2) return false;
    const uint64_t a_rows = gather ? (uint64_t)p.nimg * p.hin * p.win : (uint64_t)p.M;
    const uint64_t a_bytes = a_rows * (uint64_t)(p.lda0 > p.lda1 ? p.lda0 : p.lda1) * 2u, b_bytes = (uint64_t)p.N * p.ldb * 2u;
    if (a_bytes >= (1ull << 32) || b_bytes >= (1ull << 32)) return false;
    return true;
}
int gemm_w32_gn_parts(const i2i_igemm_params& p, int dtype, int groups) {
    i2i_igemm_params q = p;
    q.gn_part = nullptr;                                   // (the query comes before the planner attaches the slab)
    return gemm_w32_eligible(q, dtype) ? g32_gn_parts(q, groups) : 0;
}
// tile == 0 routing (measured per shape against the LDS-DMA igemm, same box: profiles/r4b_bench_ops_gemm.log): the UNet's
// projections -- widths that are multiples of 160, K of at least five stages -- with at least half a round of workgroups
// win by 1.2 - 2x (1280 -> 10240 @ 2048 rows: 0.132 -> 0.065 ms); the VAE's 1x1 convolutions (K = 128 .. 512 over 0.5 - 2 M
// rows: HBM-bound streams of short tiles) are 10 - 40 % FASTER on the LDS-DMA igemm's persistent tile stream and stay there.
// 3x3 convolutions: the stride-2 downsamplers that fill the chip (the VAE encoder's three: K = 1152 .. 4608 over 32 K - 512 K
// rows); stride-1 3x3 convolutions belong to the halo kernels, the small UNet planes to the split-K LDS-DMA igemm.
// I2I_GEMM_W32=0 sends everything back to the LDS-DMA igemm, =2 takes every eligible op; I2I_GEMM_W32_CONV=0 keeps the 3x3
// convolutions off it (A/B and test hooks, read per launch).
bool gemm_w32_auto(const i2i_igemm_params& p, int dtype) {
    const char* e = getenv("I2I_GEMM_W32");
    const int mode = e ? atoi(e) : 1;
    if (mode == 0 || !gemm_w32_eligible(p, dtype)) return false;
    if (mode == 2) return true;
    if (p.ks == 3) {
        const char* ec = getenv("I2I_GEMM_W32_CONV");
        if ((ec && atoi(ec) == 0) || p.stride != 2 || (p.N % 128 && p.N % 160)) return false;
        return g32_tiles(p, g32_cfg(p)) >= 256;
    }
    if (p.N % 160 || p.K < 5 * G32_BK) return false;
    return g32_tiles(p, g32_cfg(p)) >= 128;
}
int gemm_w32(const i2i_igemm_params& p, int dtype, hipStream_t s) {
    switch (dtype) {
        case I2I_BF16: return launch_g32_t<__bf16>(p, s);
        case I2I_F16: return gemm_w32_f16(p, s);
    }
    return fail(I2I_ERR_BAD_ARG, "gemm_w32: bad dtype");
}
}  // namespace i2i
#endif  // W32_PART == 0
