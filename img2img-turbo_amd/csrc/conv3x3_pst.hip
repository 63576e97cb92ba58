// Persistent-stream variant of the halo-tiled 3x3 convolution (conv3x3.hip): a workgroup runs its tiles as ONE continuous
// sequence of (slab, tap) steps.  The weight DMA ring and the halo prefetch of the "following slab" cross tile borders, so a
// tile's first slab is staged under the previous tile's last taps and its epilogue runs with the next tile's weights already
// in flight -- the prologue / epilogue share of a tile (about one slab time: DESIGN.md section 9) overlaps instead of adding.
// EXPERIMENT BUILD ONLY (-DI2I_PST_CONV=1; also compiled into the CPU emulator build, which checks its logic): not part of the
// product library until it has been measured on hardware.  3x3 stride 1 pad 1, no sub-pixel form, single halo buffer.
#ifdef I2I_PST_CONV
#include <cstdlib>
#include "i2i_dev.h"
#include "launch.h"

namespace {

constexpr int TW = 16;
template <int V> struct ic { static constexpr int value = V; };
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(ic<N - 1>{});
    }
}

template <typename T, int TH, int BN, int WM, int WN, int PD, int MINW>
__global__ __launch_bounds__(WM* WN * 64, MINW) void conv3x3_pst_kernel(const i2i_igemm_params p) {
    constexpr bool SUBPIX = false, DBH = false;
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

    // ---- XCD-aware tile id: workgroup b runs on XCD b % 8; give every XCD one contiguous run of tiles so
    // neighbouring tiles (shared halo rows, same weights) meet in the same L2 (bijective for any grid).
    int bid;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int pl_h = p.ho, pl_w = p.wo;
    const int tiles_x = (pl_w + TW - 1) / TW, tiles_y = (pl_h + TH - 1) / TH;
    const int ntn = (p.N + BN - 1) / BN;
    const int ntiles = tiles_x * tiles_y * p.nimg * ntn;
    const int G = gridDim.x;                         // workgroup `bid` runs tiles bid, bid + G, bid + 2G, ...
    constexpr int pa = 0, pb = 0;
    // current tile (the one being multiplied / stored) -- changes at every tile border
    int tx0, ty0, img, n0;
    auto decode = [&](int t, int& tx, int& ty, int& im, int& nn) __attribute__((always_inline)) {
        const int tn = t % ntn; t /= ntn;
        tx = (t % tiles_x) * TW; t /= tiles_x;
        ty = (t % tiles_y) * TH;
        im = t / tiles_y;
        nn = tn * BN;
    };
    decode(bid, tx0, ty0, img, n0);

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
    char* Hs = i2i_smem + HS0;
    char* Bs = i2i_smem + BS0;
    char* Ss = i2i_smem + SS0;
    int hcur = 0;                                    // DBH: halo image / constant block of the slab being multiplied

    // ---- this thread's halo chunks: chunk id v = tid + j*NT -> halo pixel v>>3, chunk kc = tid&7 (constant).
    // One 32-bit pixel index per chunk (inside its image; ~0 = zero padding).  They describe the FOLLOWING slab (f_*): the
    // same tile while it has slabs left, else the workgroup's next tile -- all halo work after the prologue (loads, GN
    // constants, hand-over store) is for the following slab, the current one is read from LDS only.
    unsigned f_hpix[HPT];
    auto fill_hpix = [&](unsigned* hp_out, int tx, int ty) __attribute__((always_inline)) {
        int t_ = tid;            // opaque: nothing lane-derived in here may be hoisted out of the tile loop into registers
#ifndef I2I_EMU
        asm volatile("" : "+v"(t_));
#endif
#pragma unroll
        for (int j = 0; j < HPT; ++j) {
            const int hp = (t_ >> 3) + j * (NT / 8);
            unsigned pix = ~0u;
            if (hp < HALO) {
                const int hy = hp / HW2, hx = hp - hy * HW2;
                const int iy = ty + hy - 1, ix = tx + hx - 1;       // coordinates in the (upsampled) input plane
                if ((unsigned)iy < (unsigned)hin_up && (unsigned)ix < (unsigned)win_up)
                    pix = (unsigned)(up_src(iy, p.hin, hin_up, p.ups) * p.win + up_src(ix, p.win, win_up, p.ups));
            }
            hp_out[j] = pix;
            __builtin_amdgcn_sched_barrier(0);     // one chunk at a time: this also runs mid-stream with every accumulator live
        }
    };
    fill_hpix(f_hpix, tx0, ty0);
    auto img_base = [&](const T* a, int lda, int im) __attribute__((always_inline)) -> const char* {
        return (const char*)a + (int64_t)im * p.hin * p.win * lda * (int)sizeof(T);
    };
    const char* f_img0 = img_base(a0, p.lda0, img);
    const char* f_img1 = img_base(a1, p.lda1, img);
    int f_img = img;
#ifdef I2I_GLDS_ASM
    unsigned ld0_b = (unsigned)p.lda0 * (unsigned)sizeof(T), ld1_b = (unsigned)p.lda1 * (unsigned)sizeof(T);
#ifndef I2I_EMU
    asm volatile("" : "+v"(ld0_b), "+v"(ld1_b));
#endif
#endif

    // ---- weight DMA: piece pc = wave + q*NW covers LDS rows pc*8 .. +7; lane -> row pc*8 + (lane>>3),
    // physical chunk lane&7, i.e. source chunk (lane&7) ^ swz(row).  Rows past N are clamped (their
    // accumulator columns are never stored).
    unsigned b_voff[BPW], f_bvoff[BPW];     // per-lane BYTE offset inside the weight matrix (N*ldb*sizeof(T) < 2^32)
    auto fill_bvoff = [&](unsigned* out, int nn) __attribute__((always_inline)) {
        int l_ = lane;
#ifndef I2I_EMU
        asm volatile("" : "+v"(l_));
#endif
#pragma unroll
        for (int q = 0; q < BPW; ++q) {
            const int pc = wave + q * NW;
            const int row = pc * 8 + (l_ >> 3);
            int n = nn + row;
            n = n < p.N ? n : p.N - 1;
            out[q] = (unsigned)(n * p.ldb + (((l_ & 7) ^ lds_swz2(row)) * EPC)) * (unsigned)sizeof(T);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    fill_bvoff(b_voff, n0);
#pragma unroll
    for (int q = 0; q < BPW; ++q) f_bvoff[q] = b_voff[q];
    // weights of (slab, tap): the current tile's rows, or (fol) the following slab's tile's rows
    auto b_dma_any = [&](const unsigned* voff, int slab, int tap, int buf) __attribute__((always_inline)) {
        const char* src = (const char*)(bw + (tap * cin + slab * CK));      // wave-uniform part of the address
#pragma unroll
        for (int q = 0; q < BPW; ++q) {
            const int pc = wave + q * NW;
            if (NPIECE % NW == 0 || pc < NPIECE) glds16(src + voff[q], Bs + buf * BN * 128 + pc * 1024);
        }
    };
    auto b_dma = [&](int slab, int tap, int buf) __attribute__((always_inline)) { b_dma_any(b_voff, slab, tap, buf); };
    auto b_dma_f = [&](int slab, int tap, int buf) __attribute__((always_inline)) { b_dma_any(f_bvoff, slab, tap, buf); };

    chunk_t rh[HPT];

    // GroupNorm (scale, shift) of the slab's CK channels: CK*8 bytes by LDS-DMA into Ss (one partial piece)
    auto ss_dma = [&](int slab, int sbuf) __attribute__((always_inline)) {
        if (wave == 0 && lane < CK / 2)
            glds16(p.gn_ss + ((int64_t)f_img * cin + slab * CK) * 2 + lane * 4, Ss + sbuf * 512);
    };
    // One 16-byte load per call, ALWAYS and branch-free (padding lanes read pixel 0 of the image and are zeroed when
    // the halo is stored): the counted vmcnt waits rely on every wave issuing the same number of VMEM operations
    // per window.  `hidden`: the load is invisible to hipcc's waitcnt bookkeeping (see gload16_uncounted).
    auto halo_load = [&](int slab, int j, bool hidden) __attribute__((always_inline)) {
        const int ci = slab * CK;                                       // wave-uniform source select
        const char* base = ci < p.c0 ? f_img0 + ci * (int)sizeof(T) : f_img1 + (ci - p.c0) * (int)sizeof(T);
#ifdef I2I_GLDS_ASM
        const unsigned ldb = ci < p.c0 ? ld0_b : ld1_b;
#else
        const unsigned ldb = (unsigned)(ci < p.c0 ? p.lda0 : p.lda1) * (unsigned)sizeof(T);
#endif
        const unsigned pix = (f_hpix[j] == ~0u) ? 0u : f_hpix[j];
        const unsigned voff = pix * ldb + (unsigned)kc * 16u;
        if (hidden) gload16_uncounted(rh[j], base, voff);
        else rh[j] = *(const chunk_t*)(base + voff);
    };
    // GroupNorm affine + SiLU of one parked chunk, then its ds_write_b128 into halo image `hbuf`
    auto halo_store_one = [&](int j, int hbuf, const float* ssr) __attribute__((always_inline)) {
        const int hp = (tid >> 3) + j * (NT / 8);
        if (hp < HALO) {
            chunk_t c = (f_hpix[j] != ~0u) ? rh[j] : zero_chunk<T>();
            if (has_gn && f_hpix[j] != ~0u) {    // zero padding stays exactly zero (conv pads the ACTIVATED tensor)
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
#pragma unroll
    for (int j = 0; j < HPT; ++j) halo_load(0, j, false);
    wait_vmcnt<0>();
    // I2I_GLDS_ASM build: retire these loads in hipcc's own bookkeeping on EVERY path, here: their consumers below sit in
    // per-lane conditionals, and a load the compiler still considers pending on the skipped path gets a vmcnt wait at the
    // first reuse of its register -- inside the slab loop, draining the (then invisible) DMA ring there
#ifdef I2I_GLDS_ASM
#pragma unroll
    for (int j = 0; j < HPT; ++j) reg_fence(rh[j]);
#endif
    lds_barrier();
    halo_store_all();
    lds_barrier();

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
        for (int j = 0; j < FN; ++j) acc[i][j] = mma_chunk(kg == 0 ? w0[j] : w1[j], xq[g % 4], acc[i][j]);
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
    constexpr int NST = FM * (FN / 2);                        // 16-byte stores per wave of an exact tile's epilogue
    auto nh = [](int t) constexpr { int c = 0; for (int j = t; j < HPT && t >= 0; j += NTAPS) ++c; return c; };   // halo loads issued in tap t's window
    // fslab: index of the FOLLOWING slab inside ITS tile (slab + 1, or 0 of the next tile; == slab when nothing follows)
    // est: NST if the epilogue that ran just before this slab left NST counted stores behind (first slab of a tile), else 0
    auto step = [&](int slab, bool next_slab, int fslab, int est, auto tapc) __attribute__((always_inline)) {
        constexpr int tap = decltype(tapc)::value;
        const bool more = tap < NTAPS - 1 || next_slab;   // another step follows
        if (!DBH && tap == NTAPS - 2 && next_slab && has_gn) ss_dma(fslab, 0);   // Ss was last read at the previous hand-over
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (tap == 0) {                         // first step of a slab: the halo image is new
#pragma unroll
            for (int g = 0; g < PD; ++g) xq[g] = xf_read(0, g);
            __builtin_amdgcn_sched_group_barrier(0x100, PD, 0);
        }
        static_for<FM>([&](auto gc) __attribute__((always_inline)) { row_group(tapc, gc, more); });
        __builtin_amdgcn_sched_barrier(0);
        // -- P_s: publish B[s+1].  Outstanding VMEM ops allowed = the window issued after P_{s-1}: the halo loads of
        //    tap-1 (always issued) and, if it exists, the DMA batch of B[s+2].  At tap 0 the previous window is
        //    tap 8 of the previous slab, whose halo loads the hand-over already waited for.
        if constexpr (RING == 3) {
            constexpr int nhp = (tap == 0) ? 0 : nh(tap - 1);
            // tile border: the previous tile's stores sit between the awaited batch and this window (tap 0: [B, stores],
            // tap 1: [B, stores, window 0]); in-order vmcnt => they may stay in flight if they are counted
            if (tap <= 1 && est && (tap + 2 < NTAPS || next_slab)) wait_vmcnt<DMA_OPS + nhp + (tap <= 1 ? NST : 0)>();
            else if (tap + 2 < NTAPS || next_slab) wait_vmcnt<DMA_OPS + nhp>();
            else wait_vmcnt<nhp>();
        } else {
            wait_vmcnt<0>();                              // 2-deep ring: B[s+1] is the most recent batch
        }
        lds_barrier();
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
            const int hs = fslab;
            if constexpr (tap < HPT) halo_load(hs, tap, true);
            if constexpr (tap + NTAPS < HPT) halo_load(hs, tap + NTAPS, true);
            if constexpr (tap + 2 * NTAPS < HPT) halo_load(hs, tap + 2 * NTAPS, true);
            static_assert(DBH || 3 * NTAPS >= HPT, "halo loads do not fit the taps of a slab");
        }
        if (tap + RING < NTAPS) b_dma(slab, tap + RING, tap % RING);
        else if (next_slab) b_dma_f(fslab, tap + RING - NTAPS, tap % RING);
        }
        __builtin_amdgcn_sched_barrier(0);
        static_for<FM>([&](auto gc) __attribute__((always_inline)) { row_group(tapc, ic<decltype(gc)::value + FM>{}, more); });
        __builtin_amdgcn_sched_barrier(0);
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
            lds_barrier();
            halo_store_all();
            lds_barrier();
        }
    };

    // ---- epilogue of the CURRENT tile (tx0, ty0, img, n0); runs between two tiles of the stream with the next tile's halo
    // already in Hs and its first weight slabs in flight, so its GroupNorm scratch lives behind the GN constant block
    // (lane ids enter through an opaque copy: hipcc otherwise hoists every lane-derived piece of the output addressing out of
    // the tile loop and the step pipeline spills)
    // Returns the number of store instructions every wave is GUARANTEED to have issued (exact tile on the 16-byte path, else 0):
    // the first two steps of the next tile add it to their counted vmcnt instead of waiting for the stores to be acknowledged.
    auto epilogue_body = [&](int lr, int lq, int tid) __attribute__((always_inline)) -> int {
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
                // retire the loads in hipcc's bookkeeping on every path (their consumers sit behind per-lane `continue`s): a load it
                // still believes pending costs a vmcnt(0) at the first reuse of its register -- inside the next tile's steps
#pragma unroll
                for (int jp = 0; jp < FN / 2; ++jp)
#pragma unroll
                    for (int i = 0; i < FM; ++i) reg_fence(rres[i][jp]);
            }
#pragma unroll
            for (int jp = 0; jp < FN / 2; ++jp) {
                const int cw = (2 * jp + (lq >> 1)) * 16 + (lq & 1) * 8;     // channel inside the wave's span
                const int n = n0 + wn * WTN + cw;
                gquad[2 * jp] = cw >> 2; gquad[2 * jp + 1] = (cw >> 2) + 1;
                float bv[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) bv[r] = (p.bias_mode == 1 && n < p.N) ? p.bias[n + r] : 0.f;
#pragma unroll
                for (int r = 0; r < 8; ++r) reg_fence(bv[r]);
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
                    *(chunk_t*)((T*)p.c + m * p.ldc + n) = o;
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
            if (p.bias_mode == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) bv[r] = (n + r < p.N) ? p.bias[n + r] : 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) reg_fence(bv[r]);
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
    // ---- GroupNorm partial sums: lane -> 16 pixels (shuffles) -> wave (LDS) -> workgroup -> one slot per group.
    // Fixed reduction order: deterministic.  Only full 4-channel vectors were accumulated (host checks N % 4 == 0,
    // dtype output, channels-per-group a multiple of 4 that divides the wave's channel span).
    if (do_stats) {
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) { gs[j] += __shfl_xor(gs[j], m); gq[j] += __shfl_xor(gq[j], m); }
        lds_barrier();                                   // every wave is done with its fragments
        float* st = (float*)(Ss + 512);                  // [NW][FN*4 quads][2], behind the GN constants (Hs holds the next tile)
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
    const bool exact = ty0 + TH <= p.ho && tx0 + TW <= p.wo && n0 + BN <= p.N;
    const int nst = (wide && exact) ? NST : 0;
    note_vmem(nst);                                  // (emulator's asynchronous model: the stores the next waits count)
    return nst;
    };
    auto epilogue = [&]() __attribute__((always_inline)) -> int {
        int lr_o = lr, lq_o = lq, tid_o = tid;
#ifndef I2I_EMU
        asm volatile("" : "+v"(lr_o), "+v"(lq_o), "+v"(tid_o));
#endif
        return epilogue_body(lr_o, lq_o, tid_o);
    };

    // ---- the stream ----
    const int my_tiles = (ntiles - bid + G - 1) / G;
    int est = 0;                                     // counted stores of the previous tile's epilogue (0 before the first tile)
    // first weights of the first step (every later step finds w0 preloaded by its predecessor, across tiles too)
#pragma unroll
    for (int j = 0; j < FN; ++j) w0[j] = wf_read(0, 0, j);
    for (int ti = 0; ti < my_tiles; ++ti) {
        const bool last_tile = ti + 1 == my_tiles;
        for (int slab = 0; slab < nslab; ++slab) {
            const bool in_tile_next = slab + 1 < nslab;
            const bool next_slab = in_tile_next || !last_tile;
            const int fslab = in_tile_next ? slab + 1 : (last_tile ? slab : 0);
            if (!in_tile_next && !last_tile) {        // the following slab opens my next tile: its pixels, image and weight rows
                int ftx, fty, fimg, fn0;
                decode(bid + (ti + 1) * G, ftx, fty, fimg, fn0);
                fill_hpix(f_hpix, ftx, fty);
                f_img0 = img_base(a0, p.lda0, fimg);
                f_img1 = img_base(a1, p.lda1, fimg);
                f_img = fimg;
                fill_bvoff(f_bvoff, fn0);
            }
            static_for<NTAPS>([&](auto tc) __attribute__((always_inline)) { step(slab, next_slab, fslab, slab == 0 ? est : 0, tc); });
        }
        if (last_tile) {
            // The last slab issued its (unused) halo loads too: they must have landed before the epilogue may reuse their
            // destination registers -- hipcc does not know those registers have a write in flight.
            wait_vmcnt<0>();
#pragma unroll
            for (int j = 0; j < HPT; ++j) reg_fence(rh[j]);
        }
        est = epilogue();
        if (!last_tile) {                             // the following slab's tile becomes the current one
            decode(bid + (ti + 1) * G, tx0, ty0, img, n0);
#pragma unroll
            for (int q = 0; q < BPW; ++q) b_voff[q] = f_bvoff[q];
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
}

// LDS: one halo image + 3 weight buffers + GN constants (512 B) + GN partial scratch (1 KiB)
template <int TH, int BN> constexpr size_t pst_smem() { return (size_t)((TH + 2) * (TW + 2) + 3 * BN) * 128 + 512 + 1024; }

unsigned pst_wgs(unsigned resident) {      // same test / A-B hook as the persistent igemm
    const char* e = getenv("I2I_PERSIST_WGS");
    return e ? (unsigned)atoi(e) : resident;
}

template <typename T, int TH, int BN, int WM, int WN, int PD, int MINW>
int launch_pst(const i2i_igemm_params& p, hipStream_t s) {
    const unsigned tiles = (unsigned)(((p.wo + TW - 1) / TW) * ((p.ho + TH - 1) / TH) * p.nimg * ((p.N + BN - 1) / BN));
    constexpr size_t smem = pst_smem<TH, BN>();
    constexpr unsigned OCC = (unsigned)((160 * 1024) / smem);
    unsigned wgs = pst_wgs(256u * (OCC < 1 ? 1u : OCC));
    if (wgs == 0 || wgs > tiles) wgs = tiles;
    hipLaunchKernelGGL((conv3x3_pst_kernel<T, TH, BN, WM, WN, PD, MINW>), dim3(wgs), dim3(WM * WN * 64), smem, s, p);
    return i2i::check_launch("conv3x3_pst");
}

template <typename T>
int launch_pst_t(const i2i_igemm_params& p, int cfg, hipStream_t s) {
    if (cfg == 44) return launch_pst<T, 8, 256, 2, 4, 2, 2>(p, s);      // persistent form of tile 34 (8 waves, one workgroup per CU)
    // 43 = persistent form of tile 13; 47 (prefetch distance 3, tile 17) spills in the step pipeline: same kernel for now
    return launch_pst<T, 8, 128, 2, 2, 2, 2>(p, s);
}

}  // namespace

namespace i2i {
// tile ids 43 / 47: the 8x16x128 halo tile (prefetch distance 2 / 3) as a persistent stream
int conv3x3_pst(const i2i_igemm_params& p, int dtype, int cfg, hipStream_t s) {
    switch (dtype) {
        case I2I_F32: return launch_pst_t<float>(p, cfg, s);
        case I2I_BF16: return launch_pst_t<__bf16>(p, cfg, s);
        case I2I_F16: return launch_pst_t<_Float16>(p, cfg, s);
    }
    return fail(I2I_ERR_BAD_ARG, "conv3x3_pst: bad dtype");
}
// workgroups of a persistent launch (0 = not worth it: fewer than two tiles per resident workgroup)
bool conv3x3_pst_worthwhile(const i2i_igemm_params& p, int bn) {
    const unsigned tiles = (unsigned)(((p.wo + TW - 1) / TW) * ((p.ho + 7) / 8) * p.nimg * ((p.N + bn - 1) / bn));
    const size_t smem = bn == 256 ? pst_smem<8, 256>() : pst_smem<8, 128>();
    const unsigned wgs = pst_wgs(256u * (unsigned)((160 * 1024) / smem));
    return wgs > 0 && tiles >= 2 * wgs;
}
}  // namespace i2i
#endif  // I2I_PST_CONV
