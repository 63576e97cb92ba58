// LDS-DMA implicit GEMM for gfx950: the contraction kernel behind every nn.Linear / 1x1 conv / batched matmul
// of the forward, and behind the 3x3 convolutions the halo kernel does not take (stride 2, planes smaller
// than a halo tile, i.e. the 8x8 and 16x16 levels of the UNet).  Same contract as igemm.hip
// (C[m][n] = epilogue(alpha * sum_k A[m][k] B[n][k]), A gathered on the fly from NHWC), different engine:
//
//   * BOTH operands reach LDS by global_load_lds_dwordx4 (no VGPR staging, no ds_write, no VALU on the data):
//     a K step (64 halves / 32 floats) of the A tile [BM rows] and the B tile [BN rows] is 128 B per row;
//     each wave owns (BM+BN)/8/NW one-KiB pieces, lane -> (row, physical chunk), the per-lane SOURCE address
//     carries the XOR swizzle (and the im2col tap / padding: out-of-image taps read a 16-byte zero block).
//   * 3-deep stage ring, DMA issued three steps ahead right after the step barrier that frees the stage, and
//     waited for two steps later with a counted vmcnt (never 0 in steady state).
//   * one raw s_barrier per K step, placed between the two k-groups; fragment reads run ahead of the MFMAs
//     that use them (across k-groups and across the barrier), schedule pinned with sched_group_barrier.
//   * MFMA operands swapped (A-operand = weight rows) so a lane owns 4 consecutive n of one m: bias,
//     residual, GEGLU and the store are 8-byte (fp32 16-byte) vectors.
//   * split-K (grid z) for the weight-streaming shapes (M = B*64 pixels x K = 9*2560): fp32 partial slabs +
//     a reduce kernel that applies the epilogue; deterministic (fixed summation order).
// GroupNorm prologues are NOT handled here (the planner materialises act(GN(x)) for the small planes that
// take this path); igemm.hip remains the register-staged fallback for everything this kernel declines.
#include <cstdlib>
#include "i2i_dev.h"
#include "launch.h"

namespace {

__device__ __attribute__((aligned(16))) uint32_t g_zero16[4] = {0u, 0u, 0u, 0u};   // DMA source of padding chunks

template <int V> struct ic { static constexpr int value = V; };
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(ic<N - 1>{});
    }
}

template <typename T, int BM, int BN, int WM, int WN, int MINW, bool PERSIST>
__global__ __launch_bounds__(WM* WN * 64, MINW) void igemm_dma_kernel(const i2i_igemm_params p) {
    constexpr int NW = WM * WN;
    constexpr int EPC = Elem<T>::EPC;
    constexpr int BK = 8 * EPC;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int FM = WTM / 16, FN = WTN / 16;
    constexpr int MPC = (EPC == 4) ? 4 : 1;
    constexpr int PA = BM / 8, PB = BN / 8, PPW = (PA + PB) / NW;     // one-KiB DMA pieces per stage / per wave
    constexpr int STAGE = (BM + BN) * 128;
    static_assert(WTM % 16 == 0 && WTN % 16 == 0 && (PA + PB) % NW == 0, "");
    typedef typename Elem<T>::chunk_t chunk_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 15, lq = lane >> 4;

    // XCD-aware tile order (workgroup b -> XCD b % 8): each XCD gets a contiguous run of tiles, n fastest, so
    // the tiles that share an A row block meet in one L2.
    int bid;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int ntn = (p.N + BN - 1) / BN;
    const int ntiles = ((p.M + BM - 1) / BM) * ntn;
    // PERSIST: gridDim.x < ntiles workgroups, workgroup `bid` takes tiles bid, bid + G, bid + 2G, ... and runs them
    // as ONE continuous K-step stream (the DMA cursor runs up to three steps ahead of the MFMAs, across tile borders,
    // so a tile's loads fly under the previous tile's MFMAs and epilogue).  Otherwise one tile per workgroup.
    const int m0 = (bid / ntn) * BM, n0 = (bid % ntn) * BN;
    const int z = blockIdx.y, split = blockIdx.z;
    const int zb = z / p.zh_count, zh = z % p.zh_count;

    const int64_t a_off = (int64_t)zb * p.a_bs_b + (int64_t)zh * p.a_bs_h;
    const char* a0 = (const char*)((const T*)p.a0 + a_off);
    const char* a1 = p.a1 ? (const char*)((const T*)p.a1 + a_off) : nullptr;
    const char* bw = (const char*)((const T*)p.b + (int64_t)zb * p.b_bs_b + (int64_t)zh * p.b_bs_h);
    const char* zero = (const char*)g_zero16;
    const int cin = p.c0 + p.c1;
    const int hin_up = p.up_h ? p.up_h : (p.hin << p.ups), win_up = p.up_w ? p.up_w : (p.win << p.ups);
    const bool gather = p.ks != 1 || p.stride != 1 || p.ups != 0 || p.pad != 0;
    const int cin_shift = 31 - __builtin_clz((unsigned)(cin > 0 ? cin : 1));     // log2(cin) when cin is a power of two

    // ---- K range of this split ----
    const int nk = (p.K + BK - 1) / BK;
    const int nsp = p.splitk > 1 ? p.splitk : 1;
    const int per = (nk + nsp - 1) / nsp;
    const int s_begin = split * per;
    const int s_end = (s_begin + per < nk) ? s_begin + per : nk;

    // ---- per-n bias through LDS: one extra DMA op fetches the tile's BN floats into a 1 KiB slot behind the stage ring
    // (PERSIST: ring of 4 slots, one op in EVERY batch so the per-batch op count stays uniform), so the epilogue has
    // no global load that would have to wait (vmcnt is in-order) for the DMA batches in flight.
    const bool lb = p.bias && (p.geglu || p.bias_mode == 1) && (p.N & 3) == 0 && (((uintptr_t)p.bias & 15) == 0);
    const int bias_base = (PERSIST ? 3 : (per < 3 ? (per < 1 ? 1 : per) : 3)) * STAGE;
    auto bias_dma = [&](int n0, int slot) __attribute__((always_inline)) {
        const int c = lane * 4;
        const bool ok = lb && c < BN && n0 + c < p.N;
        glds16(ok ? (const char*)(p.bias + n0 + c) : zero, i2i_smem + bias_base + slot * 1024);
    };
    auto bias4 = [&](int n, int n0, int slot) __attribute__((always_inline)) -> f32x4 {    // bias[n .. n+3] of the tile at n0
        if (lb) return *(const f32x4*)(i2i_smem + bias_base + slot * 1024 + (n - n0) * 4);
        f32x4 b = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && (p.geglu || p.bias_mode == 1)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (n + r < p.N) b[r] = p.bias[n + r];
        }
        return b;
    };

    // ---- this wave's DMA pieces: pc = wave + q*NW; pc < PA -> A rows pc*8.., else B rows (pc-PA)*8.. ----
    // per lane: row = 8*piece + (lane>>3), physical chunk lane&7 holds source chunk (lane&7) ^ swz(row)
    unsigned pc_off[PPW];          // A: pixel index (GEMM row / image origin), B: byte offset of (row, source chunk)
    int pc_y[PPW], pc_x[PPW];      // gather only: input coordinates of tap (0,0) for an A row; y = INT_MIN/2: row >= M
    unsigned pc_chunk[PPW];        // source chunk index (for the K tail test)
    auto setup_pieces = [&](int m0, int n0) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        const int pc = wave + q * NW;
        const int r8 = lane >> 3;
        if (pc < PA) {
            const int row = pc * 8 + r8;
            const int sc = (lane & 7) ^ lds_swz2(row);
            const int m = m0 + row;
            pc_chunk[q] = sc;
            if (!gather) {
                pc_off[q] = (unsigned)(m < p.M ? m : 0);                  // row (= pixel) index; the two sources may differ in lda
                pc_y[q] = m < p.M ? 0 : -(1 << 28);
                pc_x[q] = 0;
            } else {
                const int hw = p.ho * p.wo;
                const int mm = m < p.M ? m : 0;
                const int img = mm / hw, rem = mm - img * hw;
                const int oy = rem / p.wo, ox = rem - oy * p.wo;
                pc_off[q] = (unsigned)(img * p.hin * p.win);              // pixel index of the image origin
                pc_y[q] = m < p.M ? oy * p.stride - p.pad : -(1 << 28);
                pc_x[q] = ox * p.stride - p.pad;
            }
        } else {
            const int row = (pc - PA) * 8 + r8;
            const int sc = (lane & 7) ^ lds_swz2(row);
            int n = n0 + row;
            n = n < p.N ? n : p.N - 1;                                   // clamped rows feed columns that are never stored
            pc_chunk[q] = sc;
            pc_off[q] = (unsigned)(n * p.ldb + sc * EPC) * (unsigned)sizeof(T);
            pc_y[q] = 0; pc_x[q] = 0;
        }
    }
    };
    setup_pieces(m0, n0);

    auto dma_stage = [&](int s, int stage) __attribute__((always_inline)) {     // K step s -> ring slot `stage`
        const int k0 = s * BK;
        // wave-uniform decode of the step: tap (ky,kx), channel offset and source (cin % BK == 0 when ks == 3)
        int tap = 0, ci0 = k0;
        if (p.ks != 1) { tap = k0 / cin; ci0 = k0 - tap * cin; }
        const int ky = tap / p.ks, kx = tap - ky * p.ks;
        const bool src1 = ci0 >= p.c0;
        const char* abase = src1 ? a1 + (size_t)(ci0 - p.c0) * sizeof(T) : a0 + (size_t)ci0 * sizeof(T);
        const unsigned lda_b = (unsigned)(src1 ? p.lda1 : p.lda0) * (unsigned)sizeof(T);
        char* dst = i2i_smem + stage * STAGE;
#pragma unroll
        for (int q = 0; q < PPW; ++q) {
            const int pc = wave + q * NW;
            const bool kok = k0 + (int)pc_chunk[q] * EPC < p.K;           // K tail (ks == 1 with K % BK != 0)
            const char* src;
            if (pc < PA) {
                if (!gather) {
                    src = (kok && pc_y[q] >= 0) ? abase + (pc_off[q] * lda_b + pc_chunk[q] * 16u) : zero;
                } else if (cin < BK) {
                    // narrow input (VAE conv_in: 3 -> 8 padded channels): a K step spans several taps, so the tap is
                    // decoded PER CHUNK: k = k0 + chunk*EPC -> tap = k / cin (cin a power of two: host check)
                    const int k = k0 + (int)pc_chunk[q] * EPC;
                    const int tp = k >> cin_shift, ci = k & (cin - 1);
                    const int tky = tp / p.ks, tkx = tp - tky * p.ks;
                    const int iy = pc_y[q] + tky, ix = pc_x[q] + tkx;
                    const bool ok = kok && (unsigned)iy < (unsigned)hin_up && (unsigned)ix < (unsigned)win_up;
                    const unsigned pix = pc_off[q] + (unsigned)(up_src(iy, p.hin, hin_up, p.ups) * p.win + up_src(ix, p.win, win_up, p.ups));
                    src = ok ? a0 + (pix * ((unsigned)p.lda0 * (unsigned)sizeof(T)) + (unsigned)ci * (unsigned)sizeof(T)) : zero;
                } else {
                    const int iy = pc_y[q] + ky, ix = pc_x[q] + kx;
                    const bool ok = kok && (unsigned)iy < (unsigned)hin_up && (unsigned)ix < (unsigned)win_up;
                    const unsigned pix = pc_off[q] + (unsigned)(up_src(iy, p.hin, hin_up, p.ups) * p.win + up_src(ix, p.win, win_up, p.ups));
                    src = ok ? abase + (pix * lda_b + pc_chunk[q] * 16u) : zero;
                }
            } else {
                src = kok ? bw + (size_t)k0 * sizeof(T) + pc_off[q] : zero;
            }
            glds16(src, dst + pc * 1024);
        }
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- per-lane fragment offsets: "register + immediate" for every read (rows are 16-aligned here) ----
    const int x_off = lds_chunk_off2(wm * WTM + lr, lq);                  // A rows (m):  + i*2048, ^64 for k-group 1
    constexpr bool PERM = (EPC == 8) && (FN % 2 == 0);    // 16-bit: weight rows permuted for 16-byte stores (i2i_dev.h)
    const int w_off = BM * 128 + lds_chunk_off2(wn * WTN + (PERM ? frag_row_perm(lr) : lr), lq);   // B rows (n): + j*2048
    // Two fragment register sets: {xa, w0} = k-group 0 of a stage, {xb, w1} = k-group 1.  A stage is READ
    // only between the barrier that publishes it (P_{s-1}) and its own barrier P_s: phase A of step s
    // (k-group 0 MFMAs) loads {xb, w1} of stage s; phase B (k-group 1 MFMAs, after P_s) loads {xa, w0} of
    // stage s+1.  Every read is issued a whole phase before its first use, and nothing touches stage s after
    // P_s, so the DMA of step s+3 can reuse its slot immediately.
    chunk_t xa[FM], xb[FM], w0[FN], w1[FN];
    constexpr int NL = FM + FN, LPG = (NL + FM - 1) / FM;     // fragment loads per phase / per row group

    // Row group i of a phase: LPG fragment loads for the NEXT phase (bases xl / wl: register + immediate), then FN
    // MFMAs of this one (order pinned).
    auto row_group = [&](auto kgc, auto ic_, int xl, int wl) __attribute__((always_inline)) {
        constexpr int kg = decltype(kgc)::value, i = decltype(ic_)::value;
        constexpr int l0 = i * LPG, l1 = (l0 + LPG < NL) ? l0 + LPG : NL;
#pragma unroll
        for (int l = l0; l < l1; ++l) {
            if (l < FN) { const chunk_t v = *(const chunk_t*)(i2i_smem + (wl + l * 2048)); if constexpr (kg == 0) w1[l] = v; else w0[l] = v; }
            else        { const chunk_t v = *(const chunk_t*)(i2i_smem + (xl + (l - FN) * 2048)); if constexpr (kg == 0) xb[l - FN] = v; else xa[l - FN] = v; }
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = mma_chunk(kg == 0 ? w0[j] : w1[j], kg == 0 ? xa[i] : xb[i], acc[i][j]);
        if constexpr (l1 > l0) __builtin_amdgcn_sched_group_barrier(0x100, l1 - l0, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, FN * MPC, 0);
    };

    // ---- epilogue: lane owns row m = mb + lr and, per fragment, channel quad lqc (n = nb + 4*lqc + 0..3); 16-bit
    // outputs with an even fragment count are widened to 8 consecutive n per lane (widen_pair, 16-byte stores)
    // Returns the number of store instructions this wave is GUARANTEED to have issued (0 = unknown): the persistent
    // stream adds it to its counted vmcnt so the next barrier does not wait for the stores to be acknowledged.
    auto epilogue = [&](int m0, int n0, int slot) __attribute__((always_inline)) -> int {
    const bool exact = m0 + BM <= p.M && n0 + BN <= p.N;
    const int lqc = PERM ? frag_quad_of_lane(lq) : lq;
    const int64_t c_off = (int64_t)zb * p.c_bs_b + (int64_t)zh * p.c_bs_h;
    const int64_t r_off = (int64_t)zb * p.r_bs_b + (int64_t)zh * p.r_bs_h;
    typedef T tx4 __attribute__((ext_vector_type(4)));
    if (nsp > 1) {     // split-K: raw fp32 partials [split][M][N]; the reduce kernel applies the epilogue
        float* ws = (float*)p.ws + (int64_t)split * p.M * p.N;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int m = m0 + wm * WTM + i * 16 + lr;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int n = n0 + wn * WTN + j * 16 + lqc * 4;
                if (m < p.M && n < p.N) {
                    if (n + 3 < p.N && (p.N & 3) == 0) *(f32x4*)(ws + (int64_t)m * p.N + n) = acc[i][j];
                    else
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (n + r < p.N) ws[(int64_t)m * p.N + n + r] = acc[i][j][r];
                }
            }
        }
        return 0;
    }
    const T* __restrict__ res = p.res ? (const T*)p.res + r_off : nullptr;
    if (p.geglu) {
        // B rows (and bias) are interleaved per 32: [16 value rows | 16 gate rows]; out col = n/2 block
        if constexpr (FN >= 2) {
#pragma unroll
            for (int j = 0; j + 1 < FN; j += 2) {
                const int nA = n0 + wn * WTN + j * 16 + lqc * 4;          // packed column of the 4 values
                const int nG = nA + 16;
                const int no = (n0 + wn * WTN + j * 16) / 2 + lqc * 4;    // output column of the 4 results
                if (nG >= p.N) continue;
                const f32x4 bA = bias4(nA, n0, slot), bG = bias4(nG, n0, slot);
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const int m = m0 + wm * WTM + i * 16 + lr;
                    if (m >= p.M) continue;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float a = p.alpha * acc[i][j][r] + bA[r];
                        const float g = p.alpha * acc[i][j + 1][r] + bG[r];
                        v[r] = a * gelu_erf_f(g);
                    }
                    if (res) {
                        const tx4 rv = *(const tx4*)(res + (int64_t)m * p.ldr + no);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += to_f32<T>(rv[r]);
                    }
                    tx4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = from_f32<T>(v[r]);
                    *(tx4*)((T*)p.c + c_off + (int64_t)m * p.ldc + no) = o;
                }
            }
        }
        if (exact && FN >= 2) note_vmem(FM * (FN / 2));      // (emulator's async model: the stores the next waits count)
        return (exact && FN >= 2) ? FM * (FN / 2) : 0;
    }
    const bool wide = PERM && !p.out_f32 && (p.N % 8 == 0) && (p.ldc % 8 == 0) && (!res || p.ldr % 8 == 0) &&
                      (((c_off | r_off) & 7) == 0) && (((uintptr_t)p.res & 15) == 0);
    // GroupNorm partial sums of the stored output (next layer's norm), as in conv3x3.hip: per lane and 4-channel quad,
    // then 16 rows (shuffles) -> wave (LDS) -> workgroup -> one slot per (image, row tile, group).  Wide path only (host check).
    const bool do_stats = p.gn_part != nullptr;
    float gs[FN], gq[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) { gs[j] = 0.f; gq[j] = 0.f; }
    if (wide) {
        if constexpr (PERM) {
            // residual chunks first, all in flight together (the compiler cannot hoist them over the stores itself)
            chunk_t rres[FM][FN / 2];
            if (res) {
#pragma unroll
                for (int jp = 0; jp < FN / 2; ++jp) {
                    const int n = n0 + wn * WTN + (2 * jp + (lq >> 1)) * 16 + (lq & 1) * 8;
#pragma unroll
                    for (int i = 0; i < FM; ++i) {
                        const int m = m0 + wm * WTM + i * 16 + lr;
                        const bool ok = m < p.M && n < p.N;
                        rres[i][jp] = *(const chunk_t*)(res + (int64_t)(ok ? m : m0) * p.ldr + (ok ? n : 0));
                    }
                }
            }
#pragma unroll
            for (int jp = 0; jp < FN / 2; ++jp) {
                const int n = n0 + wn * WTN + (2 * jp + (lq >> 1)) * 16 + (lq & 1) * 8;
                float bv[8];
                {
                    const f32x4 b0 = bias4(n < p.N ? n : n0, n0, slot), b1 = bias4(n < p.N ? n + 4 : n0, n0, slot);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { bv[r] = b0[r]; bv[4 + r] = b1[r]; }
                }
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    float v[8];
                    widen_pair(acc[i][2 * jp], acc[i][2 * jp + 1], v);        // wave-wide: before any lane drops out
                    const int m = m0 + wm * WTM + i * 16 + lr;
                    if (m >= p.M || n >= p.N) continue;
                    const float bm = (p.bias_mode == 2) ? p.bias[m] : 0.f;
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] = p.alpha * v[r] + bv[r] + bm;
                    if (p.act_out) {                       // one uniform branch per 8 values, not per value
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] = act_out_f(v[r], p.act_out);
                    }
                    if (res) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] += to_f32<T>(rres[i][jp][r]);
                    }
                    chunk_t o;
#pragma unroll
                    for (int r = 0; r < 8; ++r) o[r] = from_f32<T>(v[r]);
                    *(chunk_t*)((T*)p.c + c_off + (int64_t)m * p.ldc + n) = o;
                    if (do_stats) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            const float f = to_f32<T>(o[r]);
                            gs[2 * jp + (r >> 2)] += f; gq[2 * jp + (r >> 2)] += f * f;
                        }
                    }
                }
            }
            if (do_stats) {
                // quad index inside the wave's channel span of accumulator slot (2*jp + h): channel = (2*jp + (lq>>1))*16 + (lq&1)*8 + 4*h
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int msk = 1; msk < 16; msk <<= 1) { gs[j] += __shfl_xor(gs[j], msk); gq[j] += __shfl_xor(gq[j], msk); }
                float* st = (float*)(i2i_smem + bias_base + (PERSIST ? 4096 : 1024));      // [NW][FN*4 quads][2]
                lds_barrier();
                if (lr == 0) {
#pragma unroll
                    for (int jp = 0; jp < FN / 2; ++jp)
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) {
                            const int quad = (((2 * jp + (lq >> 1)) * 16 + (lq & 1) * 8) >> 2) + h2;
                            st[(wave * FN * 4 + quad) * 2 + 0] = gs[2 * jp + h2];
                            st[(wave * FN * 4 + quad) * 2 + 1] = gq[2 * jp + h2];
                        }
                }
                lds_barrier();
                const int groups = p.gn_part_groups, cpg = p.N / groups;
                const int ng_tile = BN / cpg;
                const int g = n0 / cpg + tid;
                if (tid < ng_tile && g < groups) {
                    const int c0w = tid * cpg;
                    const int wnn = c0w / WTN, q0 = (c0w - wnn * WTN) >> 2, nq = cpg >> 2;
                    float S = 0.f, Q = 0.f;
                    for (int wmm = 0; wmm < WM; ++wmm)
                        for (int q = q0; q < q0 + nq; ++q) {
                            S += st[((wmm * WN + wnn) * FN * 4 + q) * 2 + 0];
                            Q += st[((wmm * WN + wnn) * FN * 4 + q) * 2 + 1];
                        }
                    const int hw = p.ho * p.wo, parts = hw / BM;
                    const int img = m0 / hw, part = (m0 - img * hw) / BM;
                    float* out = p.gn_part + (((int64_t)img * parts + part) * groups + g) * 2;
                    out[0] = S;
                    out[1] = Q;
                }
                // (a persistent stream reuses the scratch: the barrier ahead of the next tile's writes orders them after these reads)
            }
        }
        if (exact) note_vmem(FM * (FN / 2));
        return exact ? FM * (FN / 2) : 0;
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int n = n0 + wn * WTN + j * 16 + lqc * 4;
        if (n >= p.N) continue;
        const f32x4 bv = bias4(n, n0, slot);
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int m = m0 + wm * WTM + i * 16 + lr;
            if (m >= p.M) continue;
            const float bm = (p.bias_mode == 2) ? p.bias[m] : 0.f;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = p.alpha * acc[i][j][r] + bv[r] + bm;
            if (p.act_out) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = act_out_f(v[r], p.act_out);
            }
            if (n + 3 < p.N) {
                if (res) {
                    const tx4 rv = *(const tx4*)(res + (int64_t)m * p.ldr + n);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += to_f32<T>(rv[r]);
                }
                if (p.out_f32) {
                    *(f32x4*)((float*)p.c + c_off + (int64_t)m * p.ldc + n) = f32x4{v[0], v[1], v[2], v[3]};
                } else {
                    tx4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = from_f32<T>(v[r]);
                    *(tx4*)((T*)p.c + c_off + (int64_t)m * p.ldc + n) = o;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (n + r < p.N) {
                        float t = v[r];
                        if (res) t += to_f32<T>(res[(int64_t)m * p.ldr + n + r]);
                        if (p.out_f32) ((float*)p.c + c_off)[(int64_t)m * p.ldc + n + r] = t;
                        else ((T*)p.c + c_off)[(int64_t)m * p.ldc + n + r] = from_f32<T>(t);
                    }
                }
            }
        }
    }
    return 0;
    };

    // One K step.  `cur` = ring slot of the step (runtime: ONE copy of the step in the instruction stream).  The DMA
    // batch of step s+3 (B_s) goes out right after P_s and is waited for at P_{s+2} with a counted vmcnt that leaves
    // everything issued after it in flight: two steps of latency cover.  vmcnt retires in issue order on gfx9, so
    // "B_{s-2} has landed" == "at most (ops issued after B_{s-2}) outstanding" = OPB (batch B_{s-1}) plus, in the
    // persistent stream, the stores of the tile epilogues that ran after steps s-2 and s-1 (`est` groups of NST
    // guaranteed stores; epilogues that cannot guarantee a count contribute 0, which only waits longer).
    // (Past the last step phase B prefetches a stale stage; those fragments are never used.)
    constexpr int OPB = PPW + (PERSIST ? 1 : 0);       // DMA ops per in-loop batch (persistent: + the bias slot)
    constexpr int NST = FM * (FN / 2);
    int cur = 0;
    auto step = [&](bool has2, int est, auto&& dma) __attribute__((always_inline)) {
        const int nxt = cur == 2 ? 0 : cur + 1;
        const int xA = (x_off ^ 64) + cur * STAGE, wA = (w_off ^ 64) + cur * STAGE;     // k-group 1 of this stage
        const int xB = x_off + nxt * STAGE, wB = w_off + nxt * STAGE;                   // k-group 0 of the next one
        __builtin_amdgcn_sched_barrier(0);
        static_for<FM>([&](auto gc) __attribute__((always_inline)) { row_group(ic<0>{}, gc, xA, wA); });
        __builtin_amdgcn_sched_barrier(0);
        if (!has2) wait_vmcnt<0>();
        else if (!PERSIST || est == 0) wait_vmcnt<OPB>();
        else if (est == 1) wait_vmcnt<OPB + NST>();
        else wait_vmcnt<OPB + 2 * NST>();
        lds_barrier();
        dma(cur);
        __builtin_amdgcn_sched_barrier(0);
        static_for<FM>([&](auto gc) __attribute__((always_inline)) { row_group(ic<1>{}, gc, xB, wB); });
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
    };
    auto first_frags = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < FN; ++j) w0[j] = *(const chunk_t*)(i2i_smem + (w_off + j * 2048));
#pragma unroll
        for (int i = 0; i < FM; ++i) xa[i] = *(const chunk_t*)(i2i_smem + (x_off + i * 2048));
    };

    if constexpr (!PERSIST) {
        if (s_begin < s_end) {
            // ---- prologue: up to three stages (+ the bias slot) in flight, drain, publish ----
            for (int i = 0; i < 3 && s_begin + i < s_end; ++i) dma_stage(s_begin + i, i);
            if (nsp == 1) bias_dma(n0, 0);
            wait_vmcnt<0>();
            lds_barrier();
            first_frags();
            for (int s = s_begin; s < s_end; ++s)
                step(s + 2 < s_end, 0, [&](int st) __attribute__((always_inline)) { if (s + 3 < s_end) dma_stage(s + 3, st); });
        }
        epilogue(m0, n0, 0);
    } else {
        // ---- persistent stream over this workgroup's tiles (whole K per tile: no split-K, one z) ----
        const int G = gridDim.x;
        const int total = ((ntiles - bid + G - 1) / G) * nk;      // K steps of all my tiles
        int d_tile = bid, d_k = 0, d_ord = 0;                     // DMA cursor: (tile, K step) of the next stage to fetch
        auto dma_next = [&](int st) __attribute__((always_inline)) {
            dma_stage(d_k, st);
            bias_dma((d_tile % ntn) * BN, d_ord & 3);
            if (++d_k == nk) {
                d_k = 0; d_tile += G; ++d_ord;
                if (d_tile < ntiles) setup_pieces((d_tile / ntn) * BM, (d_tile % ntn) * BN);
            }
        };
        for (int i = 0; i < 3 && i < total; ++i) dma_next(i);
        wait_vmcnt<0>();
        lds_barrier();
        first_frags();
        int gs = 0, c_ord = 0;
        int e1 = 0, e2 = 0;             // 1 if the epilogue after the previous / second-previous step left NST counted stores
        for (int c_tile = bid; c_tile < ntiles; c_tile += G, ++c_ord) {
            for (int k = 0; k < nk; ++k, ++gs) {
                step(gs + 2 < total, e1 + e2, [&](int st) __attribute__((always_inline)) { if (gs + 3 < total) dma_next(st); });
                e2 = e1; e1 = 0;
            }
            e1 = epilogue((c_tile / ntn) * BM, (c_tile % ntn) * BN, c_ord & 3) == NST ? 1 : 0;
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
}

// split-K reduce: out = epilogue(alpha * sum_s ws[s][m][n]); fixed summation order (deterministic).
// Every load of a thread is in flight before the first is used: the slabs eight at a time (the adds keep the slice order, so the bits
// do not depend on the batching), the bias quad and the residual quad with them.  The first form -- one slab load per loop iteration,
// bias and residual behind per-element branches after the sum -- was a chain of `splitk` + 2 dependent L2 / HBM round trips: 6.75 us per
// launch, 151 launches in a batch-1 forward.
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const i2i_igemm_params p) {
    const int64_t total = (int64_t)p.M * p.N;
    const int64_t e0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e0 >= total) return;
    const float* ws = (const float*)p.ws;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    const bool vec = (p.N & 3) == 0;
    typedef T tx4 __attribute__((ext_vector_type(4)));
    const bool bvec_ok = p.bias_mode != 1 || (((uintptr_t)p.bias & 15) == 0);
    const bool rvec_ok = !p.res || ((p.ldr & 3) == 0 && (((uintptr_t)p.res & (4 * sizeof(T) - 1)) == 0));
    if (vec && bvec_ok && rvec_ok) {
        // a quad lies in one row: m, n .. n + 3.  Bias and residual are loaded UNCONDITIONALLY (from the quad's own slab address when
        // the launch has none: valid, aligned, unused) -- a load behind a runtime branch is waited for at the join, in front of the slabs
        const int m = (int)(e0 / p.N), n = (int)(e0 - (int64_t)m * p.N);
        const f32x4 bq = *(const f32x4*)(p.bias_mode == 1 ? p.bias + n : ws + e0);
        const float bm = *(p.bias_mode == 2 ? p.bias + m : ws + e0);
        const tx4 r4 = *(const tx4*)(p.res ? (const T*)p.res + (int64_t)m * p.ldr + n : (const T*)(ws + e0));
        int s = 0;
        for (; s + 8 <= p.splitk; s += 8) {
            f32x4 t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = *(const f32x4*)(ws + (int64_t)(s + k) * total + e0);
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += t[k][r];
        }
        if (s < p.splitk) {
            f32x4 t[8];      // the last batch: slices past the end re-read the last one and are not added
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = *(const f32x4*)(ws + (int64_t)(s + k < p.splitk ? s + k : p.splitk - 1) * total + e0);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (s + k < p.splitk) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += t[k][r];
                }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float t = p.alpha * v[r];
            if (p.bias_mode == 1) t += bq[r];
            else if (p.bias_mode == 2) t += bm;
            t = act_out_f(t, p.act_out);
            if (p.res) t += to_f32<T>(r4[r]);
            v[r] = t;
        }
        if (p.out_f32) {
            if ((p.ldc & 3) == 0 && (((uintptr_t)p.c & 15) == 0)) *(f32x4*)((float*)p.c + (int64_t)m * p.ldc + n) = f32x4{v[0], v[1], v[2], v[3]};
            else
#pragma unroll
                for (int r = 0; r < 4; ++r) ((float*)p.c)[(int64_t)m * p.ldc + n + r] = v[r];
        } else {
            tx4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = from_f32<T>(v[r]);
            if ((p.ldc & 3) == 0 && (((uintptr_t)p.c & (4 * sizeof(T) - 1)) == 0)) *(tx4*)((T*)p.c + (int64_t)m * p.ldc + n) = o;
            else
#pragma unroll
                for (int r = 0; r < 4; ++r) ((T*)p.c)[(int64_t)m * p.ldc + n + r] = o[r];
        }
        return;
    }
    for (int s = 0; s < p.splitk; ++s) {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (e0 + r < total) v[r] += ws[(int64_t)s * total + e0 + r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t e = e0 + r;
        if (e >= total) break;
        const int m = (int)(e / p.N), n = (int)(e - (int64_t)m * p.N);
        float t = p.alpha * v[r];
        if (p.bias_mode == 1) t += p.bias[n];
        else if (p.bias_mode == 2) t += p.bias[m];
        t = act_out_f(t, p.act_out);
        if (p.res) t += to_f32<T>(((const T*)p.res)[(int64_t)m * p.ldr + n]);
        if (p.out_f32) ((float*)p.c)[(int64_t)m * p.ldc + n] = t;
        else ((T*)p.c)[(int64_t)m * p.ldc + n] = from_f32<T>(t);
    }
}

// Workgroups of a persistent launch: all that are resident at once.  I2I_PERSIST_WGS overrides (0 = one tile per
// workgroup always): a test / A-B hook, read per launch (launches are captured into a hipGraph anyway).
unsigned persist_wgs(unsigned resident) {
    const char* e = getenv("I2I_PERSIST_WGS");
    return e ? (unsigned)atoi(e) : resident;
}

constexpr size_t GNP_LDS = 2048;      // [NW <= 8][FN*4 <= 16 quads][2] floats of GroupNorm partial scratch

template <typename T, int BM, int BN, int WM, int WN, int MINW>
int launch_dma(const i2i_igemm_params& p, hipStream_t s) {
    const unsigned tiles = (unsigned)(((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN));
    const unsigned nsp = p.splitk > 1 ? (unsigned)p.splitk : 1u;
    // ring slots actually used: a split with fewer than 3 K steps needs fewer (short-K GEMMs are latency
    // bound: less LDS = more workgroups per CU to overlap DMA latency, MFMAs and epilogue stores)
    const int nk = (p.K + 8 * Elem<T>::EPC - 1) / (8 * Elem<T>::EPC);
    const int per = (nk + (int)nsp - 1) / (int)nsp;
    constexpr size_t STAGE = (size_t)(BM + BN) * 128;
    // persistent stream when the tiles outnumber the resident workgroups (one z, whole K per tile)
    constexpr unsigned OCC = (unsigned)((160 * 1024) / (3 * STAGE + 4096 + GNP_LDS)), RES = 256u * (OCC < 1 ? 1u : OCC);
    const unsigned wgs = persist_wgs(RES);
    if (nsp == 1 && p.zcount == 1 && wgs > 0 && tiles >= 2 * wgs) {     // measured: 1.5 rounds gain nothing (tail imbalance)
        hipLaunchKernelGGL((igemm_dma_kernel<T, BM, BN, WM, WN, MINW, true>), dim3(wgs), dim3(WM * WN * 64), 3 * STAGE + 4096 + GNP_LDS, s, p);
        return i2i::check_launch("igemm_dma(persistent)");
    }
    const size_t smem = (size_t)(per < 3 ? (per < 1 ? 1 : per) : 3) * STAGE + 1024 + GNP_LDS;      // + the bias slot (+ GroupNorm partial scratch)
    hipLaunchKernelGGL((igemm_dma_kernel<T, BM, BN, WM, WN, MINW, false>), dim3(tiles, (unsigned)p.zcount, nsp), dim3(WM * WN * 64), smem, s, p);
    int rc = i2i::check_launch("igemm_dma");
    if (rc != I2I_OK || nsp == 1) return rc;
    const int64_t total = (int64_t)p.M * p.N;
    hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3((unsigned)((total + 1023) / 1024)), dim3(256), 0, s, p);
    return i2i::check_launch("splitk_reduce");
}

// tile ids 20..29 (i2i_igemm_params.tile): 20 = auto
int dma_cfg(const i2i_igemm_params& p) {
    int cfg = p.tile;
    if (cfg == 0 || cfg == 20) {
        if (p.M <= 2048) cfg = (p.N <= 64) ? 24 : 23;          // measured (profiles/): BM = 64 for the small-M shapes
        else if (p.N <= 64) cfg = 22;
        else cfg = 25;
    }
    return cfg;
}
template <typename T>
int launch_dma_t(const i2i_igemm_params& p, hipStream_t s) {
    const int cfg = dma_cfg(p);
    switch (cfg) {
        case 21: return launch_dma<T, 128, 128, 2, 2, 2>(p, s);   // 96 KiB ring, 1 workgroup / CU
        case 22: return launch_dma<T, 128, 64, 2, 2, 2>(p, s);    // 72 KiB ring, 2 workgroups / CU
        case 23: return launch_dma<T, 64, 128, 2, 2, 2>(p, s);    // small M (weight streaming), 72 KiB
        case 24: return launch_dma<T, 64, 64, 2, 2, 2>(p, s);     // 48 KiB, 3 workgroups / CU
        case 25: return launch_dma<T, 256, 128, 4, 2, 2>(p, s);   // 8 waves, 144 KiB
        case 26: return launch_dma<T, 64, 32, 4, 1, 2>(p, s);     // few rows x short K (batch-1 linears): 36 KiB ring, enough tiles to fill the chip WITHOUT K slices
    }
    return i2i::fail(I2I_ERR_BAD_ARG, "igemm_dma: unknown tile config %d", cfg);
}

}  // namespace

namespace i2i {
// The split-K reduce + epilogue launch, shared with the wide GEMM (gemm_w32.hip): out = epilogue(alpha * sum_s ws[s][m][n]).
int splitk_reduce(const i2i_igemm_params& p, int dtype, hipStream_t s) {
    const int64_t total = (int64_t)p.M * p.N;
    const dim3 g((unsigned)((total + 1023) / 1024)), b(256);
    switch (dtype) {
        case I2I_F32: hipLaunchKernelGGL((splitk_reduce_kernel<float>), g, b, 0, s, p); break;
        case I2I_BF16: hipLaunchKernelGGL((splitk_reduce_kernel<__bf16>), g, b, 0, s, p); break;
        case I2I_F16: hipLaunchKernelGGL((splitk_reduce_kernel<_Float16>), g, b, 0, s, p); break;
        default: return fail(I2I_ERR_BAD_ARG, "splitk_reduce: bad dtype");
    }
    return check_launch("splitk_reduce");
}
// What the DMA engine takes: no GroupNorm prologue, slab-aligned taps, vector-aligned outputs, 32-bit offsets.
bool igemm_dma_eligible(const i2i_igemm_params& p, int dtype) {
    const int epc = (dtype == I2I_F32) ? 4 : 8, bk = 8 * epc;
    const size_t sz = (dtype == I2I_F32) ? 4 : 2;
    const int cin = p.c0 + p.c1;
    if (p.gn_ss) return false;
    const bool narrow = p.ks != 1 && cin < bk && p.c1 == 0 && (cin & (cin - 1)) == 0 && cin >= epc;   // per-chunk tap decode
    if (p.ks != 1 && !narrow && (cin % bk || p.c0 % bk)) return false;
    if (p.ks == 1 && p.c1 && (p.c0 % bk)) return false;
    if (p.ldc % 4 || (p.res && p.ldr % 4)) return false;
    if ((p.c_bs_b | p.c_bs_h | p.r_bs_b | p.r_bs_h) & 3) return false;
    if (((uintptr_t)p.c & 15) || ((uintptr_t)p.res & 7)) return false;
    if (p.geglu && (p.N % 32)) return false;
    const uint64_t a_bytes = (uint64_t)p.nimg * p.hin * p.win * (uint64_t)(p.lda0 > p.lda1 ? p.lda0 : p.lda1) * sz;
    const uint64_t b_bytes = (uint64_t)p.N * p.ldb * sz;
    if (a_bytes >= (1ull << 32) || b_bytes >= (1ull << 32)) return false;
    if (p.splitk > 1 && (!p.ws || p.zcount > 1 || p.geglu)) return false;
    if (p.act_out && p.geglu) return false;
    return true;
}
// GroupNorm partial-sum slots per image the LDS-DMA igemm writes for this op (0 = it cannot): 16-bit wide-store path, one
// z, no split-K / GEGLU, row tiles that do not straddle images, channels-per-group a multiple of 4 dividing a wave's span.
int igemm_dma_gn_parts(const i2i_igemm_params& p, int dtype, int groups) {
    if (!igemm_dma_eligible(p, dtype) || dtype == I2I_F32 || p.out_f32 || p.geglu || p.splitk > 1 || p.zcount > 1) return 0;
    if (groups < 1 || p.N % groups || p.N % 8 || p.ldc % 8 || (p.res && p.ldr % 8) || ((uintptr_t)p.res & 15)) return 0;
    int bm, wtn;
    switch (dma_cfg(p)) {
        case 21: bm = 128; wtn = 64; break;
        case 22: bm = 128; wtn = 32; break;
        case 23: bm = 64; wtn = 64; break;
        case 24: bm = 64; wtn = 32; break;
        case 25: bm = 256; wtn = 64; break;
        case 26: bm = 64; wtn = 32; break;
        default: return 0;
    }
    const int cpg = p.N / groups, hw = p.ho * p.wo;
    if (cpg % 4 || wtn % cpg || hw % bm) return 0;
    return hw / bm;
}
int igemm_dma(const i2i_igemm_params& p, int dtype, hipStream_t s) {
    switch (dtype) {
        case I2I_F32: return launch_dma_t<float>(p, s);
        case I2I_BF16: return launch_dma_t<__bf16>(p, s);
        case I2I_F16: return launch_dma_t<_Float16>(p, s);
    }
    return fail(I2I_ERR_BAD_ARG, "igemm_dma: bad dtype");
}
}  // namespace i2i
