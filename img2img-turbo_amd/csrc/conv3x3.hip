// Halo-tiled 3x3 stride-1 pad-1 implicit-GEMM convolution for gfx950 -- the kernel that carries ~87 % of
// the forward's FLOPs (every resnet conv and upsampler conv of the VAE and UNet; F.conv2d in
// diffusers ResnetBlock2D / Upsample2D, with the preceding F.group_norm + F.silu, F.interpolate(nearest 2x)
// and torch.cat folded into the operand staging, bias / residual into the epilogue).
//
// Workgroup = TH x 16 output pixels of ONE image x BN output channels; WM x WN waves, each owning
// TH/WM tile rows x BN/WN channels as 16x16 fp32 MFMA fragments.  K loop = (channel slab of 64 halves /
// 32 floats) x (9 taps):
//   * A operand (pixels): the (TH+2) x 18 input halo of the slab is staged ONCE per slab
//     (global_load_dwordx4 -> GroupNorm affine + SiLU in registers -> ds_write_b128) and the nine taps
//     read their fragments from that one LDS image at shifted rows; an m-fragment is one 16-pixel tile
//     row, so tap (dy,dx) reads 16 consecutive 128-byte rows starting at (ty+dy)*18 + dx.  The XOR
//     swizzle lds_chunk_off2 keeps those reads conflict-free for every start row.
//   * B operand (weights [N][9*Cin], LoRA merged): streamed by LDS-DMA (global_load_lds_dwordx4) into a
//     double buffer, one [BN][64] slab per tap, issued one tap ahead; the per-lane SOURCE address carries
//     the swizzle, the LDS destination is lane-linear.  No VGPRs, no ds_write, no VALU.
//   * one raw s_barrier per tap (plus one per slab for the halo hand-over); vmcnt is only drained where
//     the DMA result is needed.  LDS = halo (single buffer: the next slab's halo waits in registers) +
//     2 weight buffers = 72.5 KiB for the 16x16x128 tile, so two workgroups share a CU and cover each
//     other's prologue / slab hand-over / epilogue.
//   * MFMA operands are issued swapped (A = weights, B = pixels) so an accumulator lane holds 4
//     CONSECUTIVE channels of one pixel: bias / residual / store move as 8-byte (16-byte fp32) vectors.
// Nearest-2x upsampling is an index map while staging the halo (Upsample2D never materialises).
#include "i2i_dev.h"
#include "launch.h"

namespace {

constexpr int TW = 16;
template <int V> struct ic { static constexpr int value = V; };   // compile-time int passed through generic lambdas
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) {   // f(ic<0>{}) ... f(ic<N-1>{})
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(ic<N - 1>{});
    }
}

// PD = pixel-fragment prefetch distance in row groups (2 or 3); MINW = min waves per SIMD for the register allocator.
// SUBPIX: sub-pixel form of "nearest-2x upsample then 3x3 conv" (diffusers Upsample2D): output pixel (2y+a, 2x+b)
// only ever sees a 2x2 neighbourhood of the SOURCE plane, so each output parity (a,b) is a 2x2 convolution of
// the source with the tap weights pre-summed (packer.subpixel_weights) -- 4/9 of the MFMA work, no index map.
// A workgroup then owns TH x 16 SOURCE positions of one parity: halo (TH+1) x 17, 4 taps per slab, 2-deep weight
// ring (4 % 3 != 0), outputs scattered to (2y+a, 2x+b).
// DBH: double-buffered halo.  The next slab's halo is transformed and stored into the OTHER halo image one chunk
// per tap, interleaved with the MFMAs of the current slab (chunk loaded in tap t's window, stored in tap t+1's), so
// the serial hand-over (two barriers + the whole GroupNorm/SiLU pass with the matrix pipe idle) disappears and
// only ~2 chunks are live in registers.  The LDS this costs is paid with a 2-deep weight ring (buffer = step
// parity, toggled at run time because 9 taps is odd): (2 x halo + 2 x weights) of the 8x16x128 tile = 79 KiB,
// still two workgroups per CU.
template <typename T, int TH, int BN, int WM, int WN, int PD, int MINW, bool SUBPIX, bool DBH>
__global__ __launch_bounds__(WM* WN * 64, MINW) void conv3x3_halo_kernel(const i2i_igemm_params p) {
    constexpr int NW = WM * WN, NT = NW * 64;
    constexpr int EPC = Elem<T>::EPC;
    constexpr int CK = 8 * EPC;                      // channels per slab: 64 (16-bit) / 32 (f32)
    constexpr int KS = SUBPIX ? 2 : 3, NTAPS = KS * KS, RING = (SUBPIX || DBH) ? 2 : 3;
    constexpr int HW2 = TW + KS - 1, HALO = (TH + KS - 1) * HW2;
    static_assert(TH % WM == 0 && BN % (16 * WN) == 0, "");
    constexpr int FM = TH / WM;                      // m-fragments per wave = tile rows per wave
    constexpr int WTN = BN / WN, FN = WTN / 16;
    constexpr int HPT = (HALO * 8 + NT - 1) / NT;    // halo chunks per thread per slab
    constexpr int NPIECE = BN / 8;                   // 1-KiB LDS-DMA pieces per weight slab (8 rows each)
    constexpr int BPW = (NPIECE + NW - 1) / NW;      // pieces per wave
    constexpr int MPC = (EPC == 4) ? 4 : 1;          // MFMA instructions per chunk pair (f32: 4 x 16x16x4)
    typedef typename Elem<T>::chunk_t chunk_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 15, lq = lane >> 4;
    const int kc = tid & 7;
    // Measurement hook (csrc/build.py --tag trace --defs=-DI2I_TRACE=1; never in the product build): every wave sums the
    // shader cycles (s_memtime) it spends in each segment of the step pipeline and writes the eight totals to p.ws
    // [workgroup][wave][8] at the end (benchmarks/bench_ops.py --trace prints the averages).  The s_memtime reads drain
    // lgkmcnt, so the traced kernel is ~10 % slower than the product one; the split between segments is what it is for.
#ifdef I2I_TRACE
    unsigned tr_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned tr_t = (unsigned)__builtin_amdgcn_s_memtime();
#define I2I_TR(k) do { const unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime(); tr_acc[k] += t_ - tr_t; tr_t = t_; } while (0)
    // ablation switches of the trace build (p.splitk is otherwise unused by this kernel; results are WRONG with any bit set):
    // 1 no output stores, 2 no bias loads, 4 no weight DMA inside the step loop, 8 no MFMAs, 16 no halo loads inside the step loop
#define I2I_ABL(bit) ((p.splitk & (bit)) != 0)
#define I2I_SETPRIO(v) do { if (I2I_ABL(1024)) __builtin_amdgcn_s_setprio(v); } while (0)
#else
#define I2I_SETPRIO(v) do { } while (0)
#define I2I_TR(k) do { } while (0)
#define I2I_ABL(bit) false
#endif

    // ---- XCD-aware tile id: workgroup b runs on XCD b % 8; give every XCD one contiguous run of tiles so
    // neighbouring tiles (shared halo rows, same weights) meet in the same L2 (bijective for any grid).
    int bid;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
#ifdef I2I_TRACE
    // phase-stagger experiment: the two workgroups of a CU start together and run identical tiles, so their serial phases
    // (prologue, slab hand-over, epilogue) coincide; delay one of each pair once at the start of the launch
    if (blockIdx.x < 512) {
        const int idx = blockIdx.x >> 3;
        if ((I2I_ABL(64) && (idx & 1)) || (I2I_ABL(128) && (idx & 32))) { __builtin_amdgcn_s_sleep(96); }
        if ((I2I_ABL(256) && (idx & 1)) || (I2I_ABL(512) && (idx & 32))) { __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); }
    }
#endif
    // plane the tiles walk over: the output plane, or the SOURCE plane for the sub-pixel form
    const int pl_h = SUBPIX ? p.hin : p.ho, pl_w = SUBPIX ? p.win : p.wo;
    const int tiles_x = (pl_w + TW - 1) / TW, tiles_y = (pl_h + TH - 1) / TH;
    const int ntn = (p.N + BN - 1) / BN;
    const int tn = bid % ntn; bid /= ntn;
    int pa = 0, pb = 0;                              // output parity (row, column) of this workgroup
    if (SUBPIX) { pa = (bid >> 1) & 1; pb = bid & 1; bid >>= 2; }
    const int tx0 = (bid % tiles_x) * TW; bid /= tiles_x;
    const int ty0 = (bid % tiles_y) * TH;
    const int img = bid / tiles_y;
    const int n0 = tn * BN;

    const T* __restrict__ a0 = (const T*)p.a0;
    const T* __restrict__ a1 = (const T*)p.a1;
    const T* __restrict__ bw = (const T*)p.b + (SUBPIX ? (int64_t)(pa * 2 + pb) * p.N * p.ldb : 0);
    const int cin = p.c0 + p.c1;
    const int hin_up = p.up_h ? p.up_h : (p.hin << p.ups), win_up = p.up_w ? p.up_w : (p.win << p.ups);
    const bool has_gn = p.gn_ss != nullptr;

    // LDS map.  DBH: weights first (so the buffer toggle is an XOR of the offset with BN*128), two halo images, two
    // GroupNorm constant blocks.  Otherwise: one halo image, the weight ring, one constant block.
    constexpr int BS0 = DBH ? 0 : HALO * 128;                        // [RING][BN] rows of 128 B
    constexpr int HS0 = DBH ? RING * BN * 128 : 0;                   // [1 or 2][HALO] rows of 128 B (one slab each)
    constexpr int SS0 = DBH ? HS0 + 2 * HALO * 128 : BS0 + RING * BN * 128;   // [1 or 2][CK][2] fp32 (scale, shift)
    constexpr int BI0 = SS0 + (DBH ? 1024 : 512);                    // [BN] fp32 bias of this channel tile
    char* Hs = i2i_smem + HS0;
    char* Bs = i2i_smem + BS0;
    char* Ss = i2i_smem + SS0;
    // The per-channel bias of the tile rides into LDS with the prologue's DMA batch: the epilogue then has no global load to
    // wait for (a bias load issued there sits a full L2 round trip -- ~7 % of a 2-slab tile -- on the critical path).
    const bool bias_lds = p.bias_mode == 1 && (p.N & 3) == 0;
    int hcur = 0;                                    // DBH: halo image / constant block of the slab being multiplied

    // ---- this thread's halo chunks: chunk id v = tid + j*NT -> halo pixel v>>3, chunk kc = tid&7 (constant).
    // One 32-bit pixel index per chunk (inside image `img`; ~0 = zero padding); the byte offset
    // pixel * ld + kc*16 is formed at the load against a wave-uniform base (SGPR base + 32-bit VGPR offset).
    unsigned hpix[HPT];
#pragma unroll
    for (int j = 0; j < HPT; ++j) {
        const int hp = (tid >> 3) + j * (NT / 8);
        unsigned pix = ~0u;
        if (hp < HALO) {
            const int hy = hp / HW2, hx = hp - hy * HW2;
            if (SUBPIX) {                                              // source coordinates of halo pixel (hy, hx)
                const int iy = ty0 + hy + pa - 1, ix = tx0 + hx + pb - 1;
                if ((unsigned)iy < (unsigned)p.hin && (unsigned)ix < (unsigned)p.win) pix = (unsigned)(iy * p.win + ix);
            } else {
                const int iy = ty0 + hy - 1, ix = tx0 + hx - 1;       // coordinates in the (upsampled) input plane
                if ((unsigned)iy < (unsigned)hin_up && (unsigned)ix < (unsigned)win_up)
                    pix = (unsigned)(up_src(iy, p.hin, hin_up, p.ups) * p.win + up_src(ix, p.win, win_up, p.ups));
            }
        }
        hpix[j] = pix;
    }
    const char* img0 = (const char*)a0 + (int64_t)img * p.hin * p.win * p.lda0 * (int)sizeof(T);
    const char* img1 = (const char*)a1 + (int64_t)img * p.hin * p.win * p.lda1 * (int)sizeof(T);

    // ---- weight DMA: piece pc = wave + q*NW covers LDS rows pc*8 .. +7; lane -> row pc*8 + (lane>>3),
    // physical chunk lane&7, i.e. source chunk (lane&7) ^ swz(row).  Rows past N are clamped (their
    // accumulator columns are never stored).
    unsigned b_voff[BPW];     // per-lane BYTE offset inside the weight matrix (N*ldb*sizeof(T) < 2^32)
#pragma unroll
    for (int q = 0; q < BPW; ++q) {
        const int pc = wave + q * NW;
        const int row = pc * 8 + (lane >> 3);
        int n = n0 + row;
        n = n < p.N ? n : p.N - 1;
        b_voff[q] = (unsigned)(n * p.ldb + (((lane & 7) ^ lds_swz2(row)) * EPC)) * (unsigned)sizeof(T);
    }
    auto b_dma = [&](int slab, int tap, int buf) __attribute__((always_inline)) {
        const char* src = (const char*)(bw + (tap * cin + slab * CK));      // wave-uniform part of the address
#pragma unroll
        for (int q = 0; q < BPW; ++q) {
            const int pc = wave + q * NW;
            if (NPIECE % NW == 0 || pc < NPIECE) glds16(src + b_voff[q], Bs + buf * BN * 128 + pc * 1024);
        }
    };

    chunk_t rh[HPT];

    // GroupNorm (scale, shift) of the slab's CK channels: CK*8 bytes by LDS-DMA into Ss (one partial piece)
    auto ss_dma = [&](int slab, int sbuf) __attribute__((always_inline)) {
        if (wave == 0 && lane < CK / 2)
            glds16(p.gn_ss + ((int64_t)img * cin + slab * CK) * 2 + lane * 4, Ss + sbuf * 512);
    };
    // One 16-byte load per call, ALWAYS and branch-free (padding lanes read pixel 0 of the image and are zeroed when
    // the halo is stored): the counted vmcnt waits rely on every wave issuing the same number of VMEM operations
    // per window.  `hidden`: the load is invisible to hipcc's waitcnt bookkeeping (see gload16_uncounted).
    auto halo_load = [&](int slab, int j, bool hidden) __attribute__((always_inline)) {
        const int ci = slab * CK;                                       // wave-uniform source select
        const char* base = ci < p.c0 ? img0 + ci * (int)sizeof(T) : img1 + (ci - p.c0) * (int)sizeof(T);
        const unsigned ldb = (unsigned)(ci < p.c0 ? p.lda0 : p.lda1) * (unsigned)sizeof(T);
        const unsigned pix = (hpix[j] == ~0u) ? 0u : hpix[j];
        const unsigned voff = pix * ldb + (unsigned)kc * 16u;
        if (hidden) gload16_uncounted(rh[j], base, voff);
        else rh[j] = *(const chunk_t*)(base + voff);
    };
    // GroupNorm affine + SiLU of one parked chunk, then its ds_write_b128 into halo image `hbuf`
    auto halo_store_one = [&](int j, int hbuf, const float* ssr) __attribute__((always_inline)) {
        const int hp = (tid >> 3) + j * (NT / 8);
        if (hp < HALO) {
            chunk_t c = (hpix[j] != ~0u) ? rh[j] : zero_chunk<T>();
            if (has_gn && hpix[j] != ~0u) {    // zero padding stays exactly zero (conv pads the ACTIVATED tensor)
                float v[EPC];
#pragma unroll
                for (int e = 0; e < EPC; ++e) v[e] = to_f32<T>(c[e]) * ssr[2 * e] + ssr[2 * e + 1];
                if (p.act == 1) {
#pragma unroll
                    for (int e = 0; e < EPC; ++e) v[e] = silu_f(v[e]);
                }
#pragma unroll
                for (int e = 0; e < EPC; ++e) c[e] = from_f32<T>(v[e]);
            }
            *(chunk_t*)(Hs + hbuf * HALO * 128 + lds_chunk_off2(hp, kc)) = c;
        }
    };
    auto load_ssr = [&](float* ssr, int sbuf) __attribute__((always_inline)) {   // this thread's 8 channels (kc) of the block
        if (has_gn) {
#pragma unroll
            for (int q = 0; q < EPC / 2; ++q) {
                const f32x4 v = *(const f32x4*)(Ss + sbuf * 512 + kc * EPC * 8 + q * 16);
                ssr[4 * q + 0] = v[0]; ssr[4 * q + 1] = v[1]; ssr[4 * q + 2] = v[2]; ssr[4 * q + 3] = v[3];
            }
        }
    };
    auto halo_store_all = [&]() __attribute__((always_inline)) {
        float ssr[2 * EPC];
        load_ssr(ssr, 0);
#pragma unroll
        for (int j = 0; j < HPT; ++j) halo_store_one(j, 0, ssr);
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nslab = cin / CK;

    // ---- prologue: weights of steps 0..2 and the slab's GN constants by DMA, halo of slab 0 through registers ----
#pragma unroll
    for (int t = 0; t < RING; ++t) b_dma(0, t, t);
    if (has_gn) ss_dma(0, 0);
    if (bias_lds && wave == NW - 1 && lane < BN / 4) {
        int n = n0 + lane * 4;
        n = n + 4 <= p.N ? n : p.N - 4;                  // lanes past a ragged tile's end re-read its last quad (never stored)
        glds16(p.bias + n, i2i_smem + BI0);
    }
#pragma unroll
    for (int j = 0; j < HPT; ++j) halo_load(0, j, false);
    I2I_TR(8);
    wait_vmcnt<0>();
    I2I_TR(9);
    lds_barrier();
    I2I_TR(10);
    halo_store_all();
    I2I_TR(11);
    lds_barrier();
    I2I_TR(12);

    // ---- per-lane LDS read offsets, all precomputed so that every fragment read is "register + immediate":
    // pixel fragment row = W + c + lr with W = wm*FM*18 (wave) and c = (i+dy)*18 + dx (compile time).  The
    // swizzle depends on (W + lr + c) mod 8 only, i.e. on c mod 8: eight per-lane bases x_off[c & 7] (and the
    // same with chunk bit 2 flipped, one v_xor, for the second k-group); the row part of c enters as the
    // immediate c * 128 since x_off[m] is built for row u = W + lr alone.
    int x_off[8];
    {
        const int u = wm * FM * HW2 + lr;
#pragma unroll
        for (int m = 0; m < 8; ++m) x_off[m] = HS0 + u * 128 + ((lq ^ lds_swz2(u + m)) << 4);
    }
    // weight fragment row = wn*WTN + j*16 + lr (swizzle independent of j): base of buffer 0, k-group 0
    constexpr bool PERM = (EPC == 8) && (FN % 2 == 0);    // 16-bit outputs: weight rows permuted for 16-byte stores
    int w_off = lds_chunk_off2(wn * WTN + (PERM ? frag_row_perm(lr) : lr), lq) + BS0;   // DBH: toggled per step (^ BN*128)
    int bcur = 0;                                    // DBH: weight buffer of the current step (uniform)

    // ---- the step pipeline.  A step = one (slab, tap) = 2 k-groups x FM tile rows = 2*FM "row groups" of FN
    // MFMAs each.  Fragment reads run AHEAD of the MFMAs that use them, across k-groups and across the
    // step barrier:
    //   xq[4]  rotating pixel fragments, slot g % 4 for row group g, read PD groups ahead;
    //   w0[FN] weights of k-group 0, re-read for the NEXT step during this step's k-group 1 (legal: the
    //          weight slab of step s+1 landed and was barrier-published at the end of step s-1, see below);
    //   w1[FN] weights of k-group 1, read during k-group 0.
    // The schedule is pinned per row group with sched_group_barrier ({reads} then {FN MFMAs}) so the list
    // scheduler neither hoists the whole step's 24 reads (VGPR blow-up) nor sinks them next to their use.
    // Weight DMA runs two steps ahead into a 3-deep ring: step s issues slab s+2 into buffer (tap+2) % 3
    // (last read in step s-1), waits for it with vmcnt(0) before the barrier that ends step s.
    chunk_t xq[4], w0[FN], w1[FN];
    constexpr int NG = 2 * FM;                       // row groups per step
    static_assert(NG % 4 == 0 && PD >= 1 && PD <= 3, "xq rotation (4 slots) must close over a step");
    auto xf_read = [&](int tapv, int g) __attribute__((always_inline)) -> chunk_t {   // pixel fragment of row group g
        const int dy = tapv / KS, dx = tapv % KS, kg = g / FM, i = g % FM;
        const int c = (i + dy) * HW2 + dx;
        return *(const chunk_t*)(i2i_smem + (x_off[c & 7] ^ (kg * 64)) + c * 128);
    };
    auto wf_read = [&](int tapv, int kg, int j) __attribute__((always_inline)) -> chunk_t {
        return *(const chunk_t*)(i2i_smem + ((w_off ^ (kg * 64)) + (tapv % RING) * BN * 128 + j * 2048));
    };
    // DBH: weight buffer = step parity; w_off already points at this step's buffer, `rel` = 1 reads the next step's
    auto wf_read_t = [&](int rel, int kg, int j) __attribute__((always_inline)) -> chunk_t {
        return *(const chunk_t*)(i2i_smem + ((w_off ^ (kg * 64) ^ (rel * BN * 128)) + j * 2048));
    };

    // One row group g of step `tap`: its prefetch reads, then its FN MFMAs (schedule pinned in that order).
    auto row_group = [&](auto tapc, auto gc, bool more) __attribute__((always_inline)) {
        constexpr int tap = decltype(tapc)::value, g = decltype(gc)::value;
        constexpr int kg = g / FM, i = g % FM;
        constexpr int WPG = (FN + FM - 1) / FM;       // weight fragments fetched per row group
        constexpr int j0 = i * WPG, nw = (j0 >= FN) ? 0 : ((j0 + WPG <= FN) ? WPG : FN - j0);
        constexpr bool xpre = (g + PD < NG) || (tap < NTAPS - 1);
        // pixel fragment PD row groups ahead (next step's first PD at the tail; not across a slab hand-over)
        if constexpr (g + PD < NG) xq[(g + PD) % 4] = xf_read(tap, g + PD);
        else if constexpr (tap < NTAPS - 1) xq[(g + PD) % 4] = xf_read(tap + 1, g + PD - NG);
        // weight fragments: k-group 1 of this step during k-group 0, k-group 0 of the next step during k-group 1
#pragma unroll
        for (int t = 0; t < nw; ++t) {
            if constexpr (DBH) {
                if constexpr (kg == 0) w1[j0 + t] = wf_read_t(0, 1, j0 + t);
                else if (more) w0[j0 + t] = wf_read_t(1, 0, j0 + t);
            } else {
                if constexpr (kg == 0) w1[j0 + t] = wf_read(tap, 1, j0 + t);
                else if (more) w0[j0 + t] = wf_read(tap + 1, 0, j0 + t);
            }
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) if (!I2I_ABL(8)) acc[i][j] = mma_chunk(kg == 0 ? w0[j] : w1[j], xq[g % 4], acc[i][j]);
        if constexpr (nw + (xpre ? 1 : 0) > 0) __builtin_amdgcn_sched_group_barrier(0x100, nw + (xpre ? 1 : 0), 0);
        __builtin_amdgcn_sched_group_barrier(0x008, FN * MPC, 0);
    };

    // Step s = (slab, tap).  Its ONE barrier P_s sits between the two k-groups:
    //   weight slab B[s] (ring buffer tap % 3) is read between P_{s-1} and P_s  (w0 during k-group 1 of step
    //   s-1, w1 during k-group 0 of step s), so it must be published by P_{s-1} and its buffer is free after
    //   P_s: the DMA of B[s+3] is issued right after P_s and waited for at P_{s+2} with a COUNTED vmcnt that
    //   leaves the batch issued after P_{s+1} in flight -- two full steps of latency cover, never vmcnt(0)
    //   in steady state.
    constexpr int DMA_OPS = (NPIECE % NW == 0) ? BPW : 0;     // DMA instructions every wave issues per batch
    auto nh = [](int t) constexpr { int c = 0; for (int j = t; j < HPT && t >= 0; j += NTAPS) ++c; return c; };   // halo loads issued in tap t's window
    auto step = [&](int slab, bool next_slab, auto tapc) __attribute__((always_inline)) {
        constexpr int tap = decltype(tapc)::value;
        const bool more = tap < NTAPS - 1 || next_slab;   // another step follows
        if (!DBH && tap == NTAPS - 2 && next_slab && has_gn) ss_dma(slab + 1, 0);   // Ss was last read at the previous hand-over
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (tap == 0) {                         // first step of a slab: the halo image is new
#pragma unroll
            for (int g = 0; g < PD; ++g) xq[g] = xf_read(0, g);
            __builtin_amdgcn_sched_group_barrier(0x100, PD, 0);
        }
        static_for<FM>([&](auto gc) __attribute__((always_inline)) { row_group(tapc, gc, more); });
        __builtin_amdgcn_sched_barrier(0);
        I2I_TR(0);
        // -- P_s: publish B[s+1].  Outstanding VMEM ops allowed = the window issued after P_{s-1}: the halo loads of
        //    tap-1 (always issued) and, if it exists, the DMA batch of B[s+2].  At tap 0 the previous window is
        //    tap 8 of the previous slab, whose halo loads the hand-over already waited for.
        if constexpr (RING == 3) {
            constexpr int nhp = (tap == 0) ? 0 : nh(tap - 1);
            if (tap + 2 < NTAPS || next_slab) wait_vmcnt<DMA_OPS + nhp>();
            else wait_vmcnt<nhp>();
        } else {
            wait_vmcnt<0>();                              // 2-deep ring: B[s+1] is the most recent batch
        }
        I2I_TR(1);
        lds_barrier();
        I2I_TR(2);
        if constexpr (DBH) {
            // -- window after P_s (every P waited vmcnt(0): the loads of the previous window have landed).
            //    Chunks j = t, t+L, t+2L are loaded in tap t's window (t < L = NTAPS-2), transformed and stored
            //    into the other halo image in tap t+1's window: the last store sits in tap L = NTAPS-2, so
            //    P_{NTAPS-1} publishes the complete image before the next slab reads it.
            constexpr int L = NTAPS - 2;
            static_assert(3 * L >= HPT, "halo chunks do not fit the taps of a slab");
            if constexpr (tap >= 1 && tap <= L) {
                if (next_slab) {
                    float ssr[2 * EPC];
                    load_ssr(ssr, hcur ^ 1);
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        constexpr int t0 = tap - 1;
                        const int j = t0 + k * L;
                        if (j < HPT) { reg_fence(rh[j]); halo_store_one(j, hcur ^ 1, ssr); }
                    }
                }
            }
            if constexpr (tap < L) {
                const int hs = next_slab ? slab + 1 : slab;       // always issued: nothing downstream may count on a branch
#pragma unroll
                for (int k = 0; k < 3; ++k) if (tap + k * L < HPT) halo_load(hs, tap + k * L, true);
            }
            if (tap == 0 && next_slab && has_gn) ss_dma(slab + 1, hcur ^ 1);     // lands by P_1, first read in tap 1's window
            // weights of step s+2 into the buffer this step just released
            if (tap + 2 < NTAPS) b_dma(slab, tap + 2, bcur);
            else if (next_slab) b_dma(slab + 1, tap + 2 - NTAPS, bcur);
        } else {
        // -- window after P_s: next slab's halo into registers (uncounted loads; from the current slab again when
        //    there is no next one, so the count per window never changes), then the DMA of B[s+3] into the buffer
        //    just released.  Halo first: the hand-over then waits with vmcnt(DMA_OPS) and leaves the DMA in flight.
        {
            const int hs = next_slab ? slab + 1 : slab;
            if (!I2I_ABL(16)) {
            if constexpr (tap < HPT) halo_load(hs, tap, true);
            if constexpr (tap + NTAPS < HPT) halo_load(hs, tap + NTAPS, true);
            if constexpr (tap + 2 * NTAPS < HPT) halo_load(hs, tap + 2 * NTAPS, true);
            }
            static_assert(DBH || 3 * NTAPS >= HPT, "halo loads do not fit the taps of a slab");
        }
        if (!I2I_ABL(4)) {
        if (tap + RING < NTAPS) b_dma(slab, tap + RING, tap % RING);
        else if (next_slab) b_dma(slab + 1, tap + RING - NTAPS, tap % RING);
        }
        }
        __builtin_amdgcn_sched_barrier(0);
        I2I_TR(3);
        static_for<FM>([&](auto gc) __attribute__((always_inline)) { row_group(tapc, ic<decltype(gc)::value + FM>{}, more); });
        __builtin_amdgcn_sched_barrier(0);
        I2I_TR(4);
        if constexpr (DBH) {
            w_off ^= BN * 128;                            // next step multiplies the other weight buffer
            bcur ^= 1;
            if (tap == NTAPS - 1 && next_slab) {          // next slab multiplies the other halo image
                const int d = hcur ? -(HALO * 128) : HALO * 128;
#pragma unroll
                for (int m = 0; m < 8; ++m) x_off[m] += d;
                hcur ^= 1;
            }
        } else if (tap == NTAPS - 1 && next_slab) {       // halo hand-over: everyone is done reading Hs
            wait_vmcnt<(RING == 3) ? DMA_OPS : 0>();                        // my halo loads have landed (the newer DMA batch stays in flight)
#pragma unroll
            for (int j = 0; j < HPT; ++j) reg_fence(rh[j]);
            I2I_SETPRIO(0);
            lds_barrier();
            halo_store_all();
            lds_barrier();
            I2I_SETPRIO(1);
            I2I_TR(5);
        }
    };

    // first weights of the first step (every later step finds w0 preloaded by its predecessor)
#pragma unroll
    for (int j = 0; j < FN; ++j) w0[j] = wf_read(0, 0, j);     // buffer 0 in both layouts
    I2I_TR(6);
    I2I_SETPRIO(1);      // MFMA stream first: the co-resident workgroup's VALU phases (prologue / hand-over / epilogue) take the leftover slots
    for (int slab = 0; slab < nslab; ++slab) {
        const bool next_slab = slab + 1 < nslab;
        static_for<NTAPS>([&](auto tc) __attribute__((always_inline)) { step(slab, next_slab, tc); });
    }
    // The last slab issued its (unused) halo loads too: they must have landed before the epilogue may reuse their
    // destination registers -- hipcc does not know those registers have a write in flight.
    I2I_SETPRIO(0);
    wait_vmcnt<0>();
#pragma unroll
    for (int j = 0; j < HPT; ++j) reg_fence(rh[j]);
    I2I_TR(13);

    // ---- epilogue: alpha, bias, residual, store, GroupNorm partial sums of what was stored.
    // Lane (lr, lq) holds pixel (oy, tx0 + lr).  16-bit outputs with an even fragment count take the wide path:
    // fragment pairs are half-exchanged (widen_pair) so every lane stores 16 bytes = 8 consecutive channels;
    // otherwise 4 consecutive channels (8 bytes, fp32 16) of channel quad `lqc` per fragment.
    const T* __restrict__ res = (const T*)p.res;
    const int sx = tx0 + lr;                         // tile-plane column of this lane; output column below
    const int ox = SUBPIX ? 2 * sx + pb : sx;
    const bool do_stats = p.gn_part != nullptr;      // GroupNorm partial sums of the stored output (next layer's norm)
    float gs[FN], gq[FN];                            // per 4-channel quad this lane accumulates (wide: 2 per fragment pair)
    int gquad[FN];                                   // its quad index inside the wave's channel span (channel / 4)
#pragma unroll
    for (int j = 0; j < FN; ++j) { gs[j] = 0.f; gq[j] = 0.f; gquad[j] = 0; }
    typedef T tx4 __attribute__((ext_vector_type(4)));
    const bool wide = PERM && !p.out_f32 && (p.N % 8 == 0) && (p.ldc % 8 == 0) && (!res || p.ldr % 8 == 0);
    if (wide) {
        if constexpr (PERM) {
            // residual chunks first, all of them in flight together (the stores below may alias them as far as
            // the compiler knows, so it would otherwise serialise load -> wait -> store per fragment)
            chunk_t rres[FM][FN / 2];
            if (res) {
#pragma unroll
                for (int jp = 0; jp < FN / 2; ++jp) {
                    const int n = n0 + wn * WTN + (2 * jp + (lq >> 1)) * 16 + (lq & 1) * 8;
#pragma unroll
                    for (int i = 0; i < FM; ++i) {
                        const int oy = SUBPIX ? 2 * (ty0 + wm * FM + i) + pa : ty0 + wm * FM + i;
                        const bool ok = ox < p.wo && oy < p.ho && n < p.N;
                        const int64_t m = ((int64_t)img * p.ho + (ok ? oy : (SUBPIX ? 2 * ty0 + pa : ty0))) * p.wo + (ok ? ox : (SUBPIX ? 2 * tx0 + pb : tx0));
                        rres[i][jp] = *(const chunk_t*)(res + m * p.ldr + (ok ? n : 0));
                    }
                }
            }
#pragma unroll
            for (int jp = 0; jp < FN / 2; ++jp) {
                const int cw = (2 * jp + (lq >> 1)) * 16 + (lq & 1) * 8;     // channel inside the wave's span
                const int n = n0 + wn * WTN + cw;
                gquad[2 * jp] = cw >> 2; gquad[2 * jp + 1] = (cw >> 2) + 1;
                float bv[8];
                if (bias_lds) {
                    const f32x4 b0 = *(const f32x4*)(i2i_smem + BI0 + (wn * WTN + cw) * 4), b1 = *(const f32x4*)(i2i_smem + BI0 + (wn * WTN + cw + 4) * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { bv[r] = b0[r]; bv[4 + r] = b1[r]; }
                } else {
#pragma unroll
                    for (int r = 0; r < 8; ++r) bv[r] = (p.bias_mode == 1 && n < p.N && !I2I_ABL(2)) ? p.bias[n + r] : 0.f;
                }
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    float v[8];
                    widen_pair(acc[i][2 * jp], acc[i][2 * jp + 1], v);        // wave-wide: before any lane drops out
                    const int oy = SUBPIX ? 2 * (ty0 + wm * FM + i) + pa : ty0 + wm * FM + i;
                    if (ox >= p.wo || oy >= p.ho || n >= p.N) continue;
                    const int64_t m = ((int64_t)img * p.ho + oy) * p.wo + ox;
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] = p.alpha * v[r] + bv[r];
                    if (res) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] += to_f32<T>(rres[i][jp][r]);
                    }
                    chunk_t o;
#pragma unroll
                    for (int r = 0; r < 8; ++r) o[r] = from_f32<T>(v[r]);
                    if (!I2I_ABL(1)) *(chunk_t*)((T*)p.c + m * p.ldc + n) = o;
                    if (do_stats) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            const float f = to_f32<T>(o[r]);
                            gs[2 * jp + (r >> 2)] += f; gq[2 * jp + (r >> 2)] += f * f;
                        }
                    }
                }
            }
        }
    } else {
        const int lqc = PERM ? frag_quad_of_lane(lq) : lq;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = n0 + wn * WTN + j * 16 + lqc * 4;
            gquad[j] = j * 4 + lqc;
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (bias_lds) {
                const f32x4 b0 = *(const f32x4*)(i2i_smem + BI0 + (wn * WTN + j * 16 + lqc * 4) * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) bv[r] = b0[r];
            } else if (p.bias_mode == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) bv[r] = (n + r < p.N) ? p.bias[n + r] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int oy = SUBPIX ? 2 * (ty0 + wm * FM + i) + pa : ty0 + wm * FM + i;
                if (ox >= p.wo || oy >= p.ho || n >= p.N) continue;
                const int64_t m = ((int64_t)img * p.ho + oy) * p.wo + ox;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = p.alpha * acc[i][j][r] + bv[r];
                if (n + 3 < p.N) {
                    if (res) {
                        const tx4 rv = *(const tx4*)(res + m * p.ldr + n);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += to_f32<T>(rv[r]);
                    }
                    if (p.out_f32) {
                        *(f32x4*)((float*)p.c + m * p.ldc + n) = f32x4{v[0], v[1], v[2], v[3]};
                    } else {
                        tx4 o;
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = from_f32<T>(v[r]);
                        *(tx4*)((T*)p.c + m * p.ldc + n) = o;
                        if (do_stats) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) { const float f = to_f32<T>(o[r]); gs[j] += f; gq[j] += f * f; }
                        }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (n + r < p.N) {
                            float t = v[r];
                            if (res) t += to_f32<T>(res[m * p.ldr + n + r]);
                            if (p.out_f32) ((float*)p.c)[m * p.ldc + n + r] = t;
                            else ((T*)p.c)[m * p.ldc + n + r] = from_f32<T>(t);
                        }
                    }
                }
            }
        }
    }
    I2I_TR(14);
    // ---- GroupNorm partial sums: lane -> 16 pixels (shuffles) -> wave (LDS) -> workgroup -> one slot per group.
    // Fixed reduction order: deterministic.  Only full 4-channel vectors were accumulated (host checks N % 4 == 0,
    // dtype output, channels-per-group a multiple of 4 that divides the wave's channel span).
    if (do_stats) {
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) { gs[j] += __shfl_xor(gs[j], m); gq[j] += __shfl_xor(gq[j], m); }
        lds_barrier();                                   // every wave is done with its fragments
        float* st = (float*)i2i_smem;                    // [NW][FN*4 quads][2]
        if (lr == 0) {
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                st[(wave * FN * 4 + gquad[j]) * 2 + 0] = gs[j];
                st[(wave * FN * 4 + gquad[j]) * 2 + 1] = gq[j];
            }
        }
        lds_barrier();
        const int groups = p.gn_part_groups, cpg = p.N / groups;
        const int ng_tile = BN / cpg;
        const int g = n0 / cpg + tid;
        if (tid < ng_tile && g < groups) {
            const int c0w = tid * cpg;                   // first channel of the group inside the n-tile
            const int wnn = c0w / WTN, q0 = (c0w - wnn * WTN) >> 2, nq = cpg >> 2;
            float S = 0.f, Q = 0.f;
            for (int wmm = 0; wmm < WM; ++wmm)
                for (int q = q0; q < q0 + nq; ++q) {
                    S += st[((wmm * WN + wnn) * FN * 4 + q) * 2 + 0];
                    Q += st[((wmm * WN + wnn) * FN * 4 + q) * 2 + 1];
                }
            // one slot per (image, spatial tile[, parity of the sub-pixel form]): every workgroup owns its own
            constexpr int NPAR = SUBPIX ? 4 : 1;
            const int tile_in_img = ((ty0 / TH) * tiles_x + tx0 / TW) * NPAR + (SUBPIX ? pa * 2 + pb : 0);
            float* out = p.gn_part + (((int64_t)img * (tiles_x * tiles_y * NPAR) + tile_in_img) * groups + g) * 2;
            out[0] = S;
            out[1] = Q;
        }
    }
#ifdef I2I_TRACE
    I2I_TR(7);
    if (p.ws && lane == 0) {
        unsigned* o = (unsigned*)p.ws + ((size_t)blockIdx.x * NW + wave) * 16;
#pragma unroll
        for (int k = 0; k < 16; ++k) o[k] = tr_acc[k];
    }
#endif
#undef I2I_TR
#undef I2I_ABL
#undef I2I_SETPRIO
}

template <typename T, int TH, int BN, int WM, int WN, int PD, int MINW, bool SUBPIX = false, bool DBH = false>
int launch_halo(const i2i_igemm_params& p, hipStream_t s) {
    constexpr int KS = SUBPIX ? 2 : 3, RING = (SUBPIX || DBH) ? 2 : 3;
    const int pl_h = SUBPIX ? p.hin : p.ho, pl_w = SUBPIX ? p.win : p.wo;
    const unsigned tiles = (unsigned)(((pl_w + TW - 1) / TW) * ((pl_h + TH - 1) / TH) * p.nimg * ((p.N + BN - 1) / BN)) * (SUBPIX ? 4u : 1u);
    const size_t smem = ((DBH ? 2 : 1) * (TH + KS - 1) * (TW + KS - 1) + RING * BN) * 128 + (DBH ? 1024 : 512) + BN * 4;
    hipLaunchKernelGGL((conv3x3_halo_kernel<T, TH, BN, WM, WN, PD, MINW, SUBPIX, DBH>), dim3(tiles), dim3(WM * WN * 64), smem, s, p);
    return i2i::check_launch("conv3x3_halo");
}

// tile ids (i2i_igemm_params.tile): 10 = auto; 11..19 force one configuration (tests / tuning)
int halo_cfg(const i2i_igemm_params& p) {
    int cfg = p.tile;
    if (cfg == 0 || cfg == 10) {
        const bool tall = p.ho >= 16;
        if (p.N <= 16) cfg = 16;
        else if (p.N <= 64 || (p.N % 128 != 0 && p.N % 128 <= 64)) cfg = tall ? 14 : 15;
        else if (p.ups && tall && p.c0 + p.c1 >= 512) cfg = 18;   // measured (profiles/r1_conv_tiles_bench_ops.log)
        else if (p.N == 256 && p.c0 + p.c1 == 256 && p.ho * p.wo >= 128 * 128) cfg = 34;   // +6 % (profiles/r1_conv_bn256_bench_ops.log)
        else if (p.c0 + p.c1 <= 128 && tall && p.ho * p.wo >= 256 * 256) cfg = 12;   // 2-slab tiles: +5..15 % over 13/17 (profiles/r2_conv_tiles_128ch_bench_ops.log)
        else cfg = (p.c0 + p.c1 <= 256) ? 17 : 13;
    }
    return cfg;
}
// (tile rows, channel tile, channels per wave) of a configuration
void halo_cfg_geometry(int cfg, int* th, int* bn, int* wtn) {
    switch (cfg) {
        case 11: case 19: *th = 16; *bn = 128; *wtn = 64; break;
        case 12: case 18: case 32: *th = 16; *bn = 128; *wtn = 64; break;
        case 13: case 17: case 31: case 33: *th = 8; *bn = 128; *wtn = 64; break;
        case 14: *th = 16; *bn = 64; *wtn = 32; break;
        case 34: *th = 8; *bn = 256; *wtn = 64; break;
        case 15: *th = 8; *bn = 64; *wtn = 32; break;
        default: *th = 8; *bn = 16; *wtn = 16; break;
    }
}

// tile rows of the sub-pixel upsampler launch (16 with 8 waves when that still fills the chip, else 8 with 4 waves)
int subpix_th(const i2i_igemm_params& p) {
    return (p.hin >= 16 && p.nimg * ((p.hin + 15) / 16) * ((p.win + 15) / 16) * ((p.N + 127) / 128) >= 128) ? 16 : 8;
}

template <typename T>
int launch_halo_t(const i2i_igemm_params& p, hipStream_t s) {
    if (p.subpix) {      // sub-pixel upsampler: weights are [4 parities][N][4*cin], tiles walk the source plane
        if (subpix_th(p) == 16) return launch_halo<T, 16, 128, 4, 2, 2, 2, true>(p, s);
        return launch_halo<T, 8, 128, 2, 2, 2, 2, true>(p, s);
    }
    const int cfg = halo_cfg(p);
    switch (cfg) {
        case 11: return launch_halo<T, 16, 128, 2, 2, 2, 1>(p, s);   // 4 waves x (8 rows x 64 ch), 1 workgroup / CU
        case 12: return launch_halo<T, 16, 128, 4, 2, 2, 2>(p, s);   // 8 waves x (4 rows x 64 ch), 1 workgroup / CU
        case 13: return launch_halo<T, 8, 128, 2, 2, 2, 2>(p, s);    // 4 waves x (4 rows x 64 ch), 2 workgroups / CU
        case 14: return launch_halo<T, 16, 64, 2, 2, 2, 2>(p, s);
        case 15: return launch_halo<T, 8, 64, 2, 2, 2, 2>(p, s);
        case 16: return launch_halo<T, 8, 16, 4, 1, 2, 2>(p, s);
        case 17: return launch_halo<T, 8, 128, 2, 2, 3, 2>(p, s);    // as 13, prefetch distance 3
        case 18: return launch_halo<T, 16, 128, 4, 2, 3, 2>(p, s);   // as 12, prefetch distance 3
        case 19: return launch_halo<T, 16, 128, 2, 2, 3, 1>(p, s);   // as 11, prefetch distance 3
        case 34: return launch_halo<T, 8, 256, 2, 4, 2, 2>(p, s);    // 8 waves x (4 rows x 64 ch): halo staged once per 256 channels
        case 31: return launch_halo<T, 8, 128, 2, 2, 2, 2, false, true>(p, s);    // as 13, double-buffered halo + 2-deep ring
        case 32: return launch_halo<T, 16, 128, 4, 2, 2, 2, false, true>(p, s);   // as 12, double-buffered halo
        case 33: return launch_halo<T, 8, 128, 2, 2, 3, 2, false, true>(p, s);    // as 31, prefetch distance 3
    }
    return i2i::fail(I2I_ERR_BAD_ARG, "conv3x3: unknown tile config %d", cfg);
}

}  // namespace

namespace i2i {
// Eligibility: 3x3, stride 1, pad 1, slab-aligned channel counts, a plane at least one tile wide.
bool conv3x3_halo_eligible(const i2i_igemm_params& p, int dtype) {
    const int ck = (dtype == I2I_F32) ? 32 : 64;
    if (p.ks != 3 || p.stride != 1 || p.pad != 1 || p.geglu || p.zcount > 1 || p.bias_mode == 2) return false;
    if (p.c0 % ck || p.c1 % ck || (p.c0 + p.c1) < ck) return false;
    if (p.wo < TW || p.ho < 8) return false;
    if (p.ho != (p.up_h ? p.up_h : (p.hin << p.ups)) || p.wo != (p.up_w ? p.up_w : (p.win << p.ups))) return false;
    if ((p.up_h || p.up_w) && (p.ups != 1 || p.subpix)) return false;
    if (p.subpix && (p.ups != 1 || p.win < TW || p.hin < 8 || p.ldb != 4 * (p.c0 + p.c1))) return false;
    if (p.ldc % 4 || (p.res && p.ldr % 4)) return false;
    return true;
}
// GroupNorm partial-sum slots per image (= spatial tiles) the halo kernel writes for this op, 0 if it cannot:
// needs dtype output, N % 4 == 0 and channels-per-group a multiple of 4 dividing the channels one wave owns.
int conv3x3_halo_gn_parts(const i2i_igemm_params& p, int dtype, int groups) {
    if (!conv3x3_halo_eligible(p, dtype) || p.out_f32 || groups < 1 || p.N % groups || p.N % 4) return 0;
    const int cpg = p.N / groups;
    if (p.subpix) {      // tiles walk the SOURCE plane, four parity workgroups (= four slots) per tile; 64 channels per wave
        if (cpg % 4 || 64 % cpg) return 0;
        const int th = subpix_th(p);
        return 4 * ((p.win + TW - 1) / TW) * ((p.hin + th - 1) / th);
    }
    int th, bn, wtn;
    halo_cfg_geometry(halo_cfg(p), &th, &bn, &wtn);
    if (cpg % 4 || wtn % cpg) return 0;
    return ((p.wo + TW - 1) / TW) * ((p.ho + th - 1) / th);
}
int conv3x3_halo(const i2i_igemm_params& p, int dtype, hipStream_t s) {
    switch (dtype) {
        case I2I_F32: return launch_halo_t<float>(p, s);
        case I2I_BF16: return launch_halo_t<__bf16>(p, s);
        case I2I_F16: return launch_halo_t<_Float16>(p, s);
    }
    return fail(I2I_ERR_BAD_ARG, "conv3x3: bad dtype");
}
}  // namespace i2i
