// Flash-style fused attention for gfx950: softmax(q k^T * scale) v, no mask, not causal, head dim 64
// (every UNet attention of SD-Turbo: 5/10/20 heads x 64; self Tk = HW, cross Tk = 77).
// Replaces F.scaled_dot_product_attention / xformers memory_efficient_attention
// (diffusers AttnProcessor2_0; enabled at src/inference_unpaired.py:36).
//
// Both matmuls are issued "swapped" so that every softmax quantity is lane-local:
//   S^T = K . Q^T   (A = K tile from LDS, B = Q from registers)  -> lane holds S^T[key=4q+r][query=l&15]
//   O^T = V^T . P^T (A = V^T tile from LDS, B = P^T = the S^T accumulators, re-used in place)
// The MFMA k->lane assignment is free as long as A and B agree, so P never leaves its registers: for
// 16-bit types the 8-key chunk of lane-quad q is {keys 16(2g)+4q..+3, 16(2g+1)+4q..+3} and V^T is read
// with the matching two 8-byte LDS reads; for f32 the 4-key chunk is one accumulator fragment as is.
// Running max / sum / rescale all belong to query l&15 = this lane: no cross-lane traffic except two
// xor-shuffles (16, 32) per tile for the max.  V arrives transposed (V^T [heads*d][Tk]) straight from the
// projection GEMM (weights as the A operand), so no transpose pass exists anywhere.
#include <stdlib.h>
#include <type_traits>
#include "i2i_dev.h"
#include "launch.h"

namespace {

template <int CPR> __device__ __forceinline__ int att_off(int row, int kc) {
    // rows of CPR 16-byte chunks, XOR-swizzled so a ds_read_b128 over 16 rows hits 16 distinct slots
    if (CPR == 8) return row * 128 + ((kc ^ ((row >> 1) & 7)) << 4);
    return row * (CPR * 16) + ((kc ^ (row & 15)) << 4);
}

__device__ __forceinline__ bf16x8 pack_p(f32x4 lo, f32x4 hi, __bf16) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) { r[j] = (__bf16)lo[j]; r[4 + j] = (__bf16)hi[j]; }
    return r;
}
__device__ __forceinline__ f16x8 pack_p(f32x4 lo, f32x4 hi, _Float16) {
    f16x8 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) { r[j] = (_Float16)lo[j]; r[4 + j] = (_Float16)hi[j]; }
    return r;
}

template <typename T, int D>
__global__ __launch_bounds__(256) void attention_kernel(const i2i_attention_params p) {
    typedef typename Elem<T>::chunk_t chunk_t;
    constexpr int EPC = Elem<T>::EPC;
    constexpr int BQ = 64, BKV = 64;
    constexpr int CPR_K = D / EPC;        // chunks per K-tile row (d contiguous)
    constexpr int CPR_V = BKV / EPC;      // chunks per V^T-tile row (keys contiguous)
    constexpr int KG = D / (4 * EPC);     // k-groups (4 lane-quads x EPC) over d
    constexpr int DF = D / 16;            // output fragments over d
    constexpr bool F32 = (EPC == 4);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * BQ + wave * 16;

    const T* qp = (const T*)p.q + (int64_t)b * p.q_bs + (int64_t)h * D;
    const T* kp = (const T*)p.k + (int64_t)b * p.k_bs + (int64_t)h * D;
    const T* vp = (const T*)p.vt + (int64_t)b * p.vt_bs + (int64_t)h * D * p.ldvt;

    char* Ks = i2i_smem;                               // [BKV][D]
    char* Vs = i2i_smem + BKV * D * (int)sizeof(T);    // [D][BKV]

    // Q^T fragments for query q0+lr: chunk kg*4+lq
    chunk_t qf[KG];
    {
        const int qi = q0 + lr;
#pragma unroll
        for (int kg = 0; kg < KG; ++kg)
            qf[kg] = (qi < p.tq) ? *(const chunk_t*)(qp + (int64_t)qi * p.ldq + (kg * 4 + lq) * EPC) : zero_chunk<T>();
    }

    f32x4 oacc[DF];
#pragma unroll
    for (int i = 0; i < DF; ++i) oacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -1e30f, l_run = 0.f;

    for (int kv0 = 0; kv0 < p.tk; kv0 += BKV) {
        __syncthreads();   // previous tile fully consumed
        // ---- stage K tile [BKV][D] and V^T tile [D][BKV] ----
        for (int v = tid; v < BKV * CPR_K; v += 256) {
            const int r = v / CPR_K, c = v % CPR_K;
            const int key = kv0 + r;
            chunk_t x = (key < p.tk) ? *(const chunk_t*)(kp + (int64_t)key * p.ldk + c * EPC) : zero_chunk<T>();
            *(chunk_t*)(Ks + att_off<CPR_K>(r, c)) = x;
        }
        for (int v = tid; v < D * CPR_V; v += 256) {
            const int r = v / CPR_V, c = v % CPR_V;
            const int key = kv0 + c * EPC;
            chunk_t x = zero_chunk<T>();
            if (key < p.tk) {
                x = *(const chunk_t*)(vp + (int64_t)r * p.ldvt + key);
                if (key + EPC > p.tk) {
#pragma unroll
                    for (int j = 0; j < EPC; ++j) if (key + j >= p.tk) x[j] = (T)0.0f;
                }
            }
            *(chunk_t*)(Vs + att_off<CPR_V>(r, c)) = x;
        }
        __syncthreads();

        // ---- S^T = K Q^T : sacc[kf][r] = S^T[key = kv0 + 16kf + 4lq + r][query = q0 + lr] ----
        f32x4 sacc[4];
#pragma unroll
        for (int kf = 0; kf < 4; ++kf) {
            sacc[kf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kg = 0; kg < KG; ++kg) {
                const chunk_t a = *(const chunk_t*)(Ks + att_off<CPR_K>(kf * 16 + lr, kg * 4 + lq));
                sacc[kf] = mma_chunk(a, qf[kg], sacc[kf]);
            }
        }
        // ---- online softmax for query lr (lane-local apart from the max) ----
        float mt = -1e30f;
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kv0 + kf * 16 + lq * 4 + r;
                float s = sacc[kf][r] * p.scale;
                if (key >= p.tk || (p.causal && key > q0 + lr)) s = -1e30f;
                sacc[kf][r] = s;
                mt = fmaxf(mt, s);
            }
        mt = fmaxf(mt, __shfl_xor(mt, 16));
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = __expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = __expf(sacc[kf][r] - m_new);
                sacc[kf][r] = pv;
                psum += pv;
            }
        l_run = l_run * alpha + psum;   // per-lane partial; quads are summed once at the end
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < DF; ++i) oacc[i] *= alpha;

        // ---- O^T += V^T P^T ----
        if constexpr (F32) {
#pragma unroll
            for (int kf = 0; kf < 4; ++kf) {
                chunk_t pb;
#pragma unroll
                for (int j = 0; j < EPC; ++j) pb[j] = (T)sacc[kf][j % 4];
#pragma unroll
                for (int i = 0; i < DF; ++i) {
                    const chunk_t a = *(const chunk_t*)(Vs + att_off<CPR_V>(i * 16 + lr, kf * 4 + lq));
                    oacc[i] = mma_chunk(a, pb, oacc[i]);
                }
            }
        } else {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const auto pb = pack_p(sacc[2 * g], sacc[2 * g + 1], T());
#pragma unroll
                for (int i = 0; i < DF; ++i) {
                    // keys 32g + 4lq..+3 (first half) and 32g + 16 + 4lq..+3 (second half) of V^T row i*16+lr
                    const int row = i * 16 + lr;
                    const char* lo = Vs + att_off<CPR_V>(row, g * 4 + (lq >> 1)) + (lq & 1) * 8;
                    const char* hi = Vs + att_off<CPR_V>(row, g * 4 + 2 + (lq >> 1)) + (lq & 1) * 8;
                    union { chunk_t c; uint64_t u[2]; } a;
                    a.u[0] = *(const uint64_t*)lo;
                    a.u[1] = *(const uint64_t*)hi;
                    oacc[i] = mma_chunk(a.c, pb, oacc[i]);
                }
            }
        }
    }

    // ---- finalize: sum the per-quad partial row sums, normalise, store 4 consecutive d per lane ----
    float l_tot = l_run;
    l_tot += __shfl_xor(l_tot, 16);
    l_tot += __shfl_xor(l_tot, 32);
    const float inv = 1.0f / l_tot;
    const int qi = q0 + lr;
    if (qi < p.tq) {
        T* op = (T*)p.o + (int64_t)b * p.o_bs + (int64_t)qi * p.ldo + (int64_t)h * D;
#pragma unroll
        for (int i = 0; i < DF; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) op[i * 16 + lq * 4 + r] = from_f32<T>(oacc[i][r] * inv);
    }
}

// ---------------------------------------------------------------------------------------------------
// LDS-DMA flash attention for the 16-bit types (d = 64).  Same swapped-matmul scheme as attention_kernel
// (softmax lane-local), different engine:
//   * a wave owns 32 queries (two query fragments): every K / V^T fragment read from LDS feeds two MFMAs, half the
//     LDS traffic per FLOP of the 16-query version; workgroup = 4 waves = 128 queries;
//   * K tile [64 keys][64 d] and V^T tile [64 d][64 keys] (8 KiB each) arrive by global_load_lds_dwordx4 into a
//     3-stage ring, issued two tiles ahead, waited for with a counted vmcnt; ONE raw barrier per key tile;
//   * rows / chunks past tk read a 16-byte zero block; the partially valid last V^T chunk (tk % 8 != 0) is cleaned
//     in LDS before use (its padding may hold anything, and 0 * NaN would poison the output).
// Round 6: the softmax between the two contractions was the bound of this kernel (profiles/r5g_pmc_attention_summary.txt: matrix
// pipe 21 % busy; ~280 VALU instructions per wave and key tile against 32 MFMAs), so the work per score is cut to what cannot be
// avoided -- one v_max3 per two scores, one v_exp, one v_cvt_pk per two:
//   * scores leave the MFMA in log2 units (scale * log2(e) rides in q: the model's to_q epilogue produces q pre-multiplied and
//     passes scale = ln 2; any other caller's q is multiplied once at load) and ALREADY SHIFTED by the running reference of their
//     query: the accumulator of the first MFMA of a score fragment starts from -m_ref instead of 0;
//   * the reference is LAZY: it only moves when some score of the tile exceeds it by more than 2^8 (wave-uniform test on the
//     lane-local maxima, no cross-lane traffic); then -- and on the first tile -- the slow path finds the query maxima, shifts the
//     tile's scores and rescales O and the row sums.  Softmax is shift invariant, so any reference is exact as long as nothing
//     overflows: P <= 2^8 in bf16 / fp16 keeps its relative precision, O and the sums accumulate in fp32;
//   * the row sums come out of the matrix pipe: one more MFMA per 32 keys against an all-ones A fragment leaves the sum of the
//     ROUNDED probabilities of query lr in every lane of its column (the same numbers the P.V product uses), no per-score add and
//     no final quad reduction;
//   * keys are taken in a permuted order inside a tile -- row m of score fragment kf is key 32(kf >> 1) + 8(m >> 2) + 4(kf & 1) +
//     (m & 3) -- so that the eight probabilities a lane packs for the second contraction are eight CONSECUTIVE keys: its V^T
//     operand is one ds_read_b128 (it was two ds_read_b64 plus register moves).  The K tile's XOR swizzle is chosen for that row
//     pattern (att_kswz: the 16 rows of a fragment read hit 16 distinct 16-byte slots).
// XCD-aware workgroup order for the DMA attention kernels (1-D grid; workgroup id -> XCD id % 8): every (batch, head)
// group keeps all its query tiles on ONE XCD, so its K / V^T tiles are fetched into that XCD's L2 once and hit by the
// other query tiles, instead of every XCD streaming every group from the fabric.  Groups g, g + 8, ... share an XCD
// one after the other.  Grid = ceil(groups / 8) * 8 * nqt; returns false for the padding workgroups.
__device__ __forceinline__ bool att_group_of_block(int nqt, int heads, int batch, int& qt, int& h, int& b) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int g = (slot / nqt) * 8 + xcd;
    qt = slot % nqt;
    h = g % heads;
    b = g / heads;
    return g < heads * batch;
}
inline unsigned att_grid(int nqt, int heads, int batch) { return (unsigned)(((heads * batch + 7) / 8) * 8 * nqt); }

__device__ __attribute__((aligned(16))) uint32_t g_att_zero16[4] = {0u, 0u, 0u, 0u};

// attention_dma_kernel's K tile: rows of 8 chunks, chunk kc of row r at physical chunk kc ^ att_kswz(r)
__device__ __forceinline__ int att_kswz(int row) { return ((row >> 1) & 1) | (((row >> 3) & 3) << 1); }
__device__ __forceinline__ int att_koff(int row, int kc) { return row * 128 + ((kc ^ att_kswz(row)) << 4); }
// the key (inside its 64-key tile) that row m of score fragment kf holds
__device__ __forceinline__ int att_key_of(int kf, int m) { return 32 * (kf >> 1) + 8 * (m >> 2) + 4 * (kf & 1) + (m & 3); }

// QF = query fragments per wave: 2 (128 queries per workgroup) for the launches that fill the chip, 1 (64 queries, twice the
// workgroups) for the small grids of a batch-1 forward, where a workgroup's 64 serial key tiles are the launch.
// SPLIT (p.ksplit > 1, p.ws; round 6): the key tiles are divided among ksplit workgroups per query tile, as in attention_wide_kernel below --
// at batch 1 the T = 4096 self-attention is 320 workgroups of 64 SERIAL key tiles, ~1 us each when a wave has its SIMD to itself; a split
// workgroup runs its share and writes the UNNORMALISED O^T (fp32) with (reference, row sum) of every query to p.ws, attention_combine64_kernel
// merges the partials.  Not with the causal mask (the text tower's 77 keys are two tiles).
template <typename T, int QF, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void attention_dma_kernel(const i2i_attention_params p) {
    typedef typename Elem<T>::chunk_t chunk_t;
    static_assert(Elem<T>::EPC == 8, "16-bit types only");
    constexpr int D = 64, BKV = 64, QT = 64 * QF, STAGE = 2 * BKV * 128, PPW = 4;   // 16 one-KiB pieces per stage, 4 per wave
    constexpr float LAZY = 8.0f;                                              // log2 units a score may exceed its reference by
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    int qt, h, b, ks = 0;
    if constexpr (!SPLIT) {
        if (!att_group_of_block((p.tq + QT - 1) / QT, p.heads, p.batch, qt, h, b)) return;
    } else {
        int hs;
        if (!att_group_of_block((p.tq + QT - 1) / QT, p.heads * p.ksplit, p.batch, qt, hs, b)) return;
        h = hs / p.ksplit;
        ks = hs - h * p.ksplit;
    }
    const int q0 = qt * QT + wave * (16 * QF);

    const T* qp = (const T*)p.q + (int64_t)b * p.q_bs + (int64_t)h * D;
    const char* kp = (const char*)((const T*)p.k + (int64_t)b * p.k_bs + (int64_t)h * D);
    const char* vp = (const char*)((const T*)p.vt + (int64_t)b * p.vt_bs + (int64_t)h * D * p.ldvt);
    int t_begin = 0, ntile = (p.tk + BKV - 1) / BKV;          // this workgroup's key tiles: t_begin .. ntile - 1
    if constexpr (SPLIT) {
        const int per = (ntile + p.ksplit - 1) / p.ksplit;
        t_begin = ks * per;
        ntile = t_begin + per < ntile ? t_begin + per : ntile;
    }

    // this wave's DMA pieces: pc = wave*4 + q; pc < 8 (waves 0, 1): K rows pc*8.. (row = key), else V^T rows (pc-8)*8.. (row = d).
    // The lane offsets of tile 0 are computed ONCE; a tile adds a wave-uniform stride to the scalar base (64 keys: 64 rows of K,
    // 128 bytes of every V^T row).  (Round 5: the per-tile form -- key / chunk validity tests and 64-bit address arithmetic per
    // piece, compiled to exec-mask branches -- was ~190 of the ~800 instructions a wave issued per tile in a VALU-bound loop.)
    // Keys past tk exist only in the LAST tile: there K rows past the end re-read the last key (masked in the softmax) and V^T
    // chunks past the end come from a zero block (the partially valid chunk is cleaned in LDS below).
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const bool k_wave = wave_u < 2;
    auto src_chunk = [&](int row) __attribute__((always_inline)) {           // source chunk of physical chunk lane & 7
        return (lane & 7) ^ (k_wave ? att_kswz(row) : ((row >> 1) & 7));
    };
    unsigned d_off[PPW];
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        const int pc = wave_u * PPW + q;
        const int row = (pc & 7) * 8 + (lane >> 3);
        const int sc = src_chunk(row);
        d_off[q] = k_wave ? (unsigned)(row * p.ldk + sc * 8) * (unsigned)sizeof(T) : (unsigned)(row * p.ldvt + sc * 8) * (unsigned)sizeof(T);
    }
    auto dma_tile = [&](int t, int stage) __attribute__((always_inline)) {
        const int kv0 = t * BKV;
        char* dst = i2i_smem + stage * STAGE;
        if (kv0 + BKV <= p.tk) {                                          // (wave-uniform)
            const char* base = k_wave ? kp + (size_t)kv0 * p.ldk * sizeof(T) : vp + (size_t)kv0 * sizeof(T);
#pragma unroll
            for (int q = 0; q < PPW; ++q) glds16_sv(base, d_off[q], dst + (wave_u * PPW + q) * 1024);
        } else {
#pragma unroll
            for (int q = 0; q < PPW; ++q) {
                const int pc = wave_u * PPW + q;
                const int row = (pc & 7) * 8 + (lane >> 3);
                const int sc = src_chunk(row);
                if (k_wave) {                                             // rows past tk: the last key again (masked below)
                    const int key = kv0 + row < p.tk ? kv0 + row : p.tk - 1;
                    glds16_sv(kp, (unsigned)(key * p.ldk + sc * 8) * (unsigned)sizeof(T), dst + pc * 1024);
                } else {                                                  // chunks past tk: zeros (the pad columns of V^T may hold anything)
                    const int key0 = kv0 + sc * 8;
                    glds16(key0 < p.tk ? vp + (size_t)(row * p.ldvt + key0) * sizeof(T) : (const char*)g_att_zero16, dst + pc * 1024);
                }
            }
        }
    };

    // Q^T fragments (B operand of S^T = K Q^T): query q0 + 16*qf + lr, chunk kg*4 + lq, in log2 score units
    const float c2 = p.scale * 1.44269504088896341f;
    const bool q_ready = fabsf(c2 - 1.0f) < 1e-6f;           // (uniform) the caller's q already carries scale * log2(e)
    chunk_t qfr[QF][2];
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        const int qi = q0 + f * 16 + lr;
#pragma unroll
        for (int kg = 0; kg < 2; ++kg) {
            chunk_t x = (qi < p.tq) ? *(const chunk_t*)(qp + (int64_t)qi * p.ldq + (kg * 4 + lq) * 8) : zero_chunk<T>();
            if (!q_ready) {
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = (T)((float)x[j] * c2);
            }
            qfr[f][kg] = x;
        }
    }
    // the compiler must see the Q loads retired BEFORE the tile loop: otherwise (it cannot see the hand-written counted
    // waits) it drains vmcnt to 0 at their first use INSIDE the loop, every tile, and the DMA ring never overlaps
#pragma unroll
    for (int f = 0; f < QF; ++f)
#pragma unroll
        for (int kg = 0; kg < 2; ++kg) reg_fence(qfr[f][kg]);
    chunk_t ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (T)1.0f;
    f32x4 oacc[QF][4], lacc[QF], negm[QF];
    float m_ref[QF];
#pragma unroll
    for (int f = 0; f < QF; ++f) {
#pragma unroll
        for (int i = 0; i < 4; ++i) oacc[f][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        lacc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
        negm[f] = f32x4{0.f, 0.f, 0.f, 0.f};
        m_ref[f] = 0.f;
    }

    if (t_begin < ntile) dma_tile(t_begin, 0);
    if (t_begin + 1 < ntile) dma_tile(t_begin + 1, 1);
    for (int t = t_begin; t < ntile; ++t) {
        const int st = (t - t_begin) % 3, kv0 = t * BKV;
        // tile t landed (leave the batch of tile t+1 in flight), everyone is done with tile t-1 -> its slot is free
        if (t + 1 < ntile) wait_vmcnt<PPW>(); else wait_vmcnt<0>();
        lds_barrier();
        if (t + 2 < ntile) dma_tile(t + 2, (t + 2 - t_begin) % 3);
        const char* Ks = i2i_smem + st * STAGE;
        const char* Vs = Ks + BKV * 128;
        if (kv0 + BKV > p.tk && (p.tk & 7)) {              // last tile, partially valid V^T chunk: clean its tail in LDS
            const int kc = (p.tk - kv0) >> 3, first = (p.tk - kv0) & 7;          // chunk index inside the tile, valid elements
            if (tid < D) {
                T* e = (T*)(i2i_smem + st * STAGE + BKV * 128 + att_off<8>(tid, kc));
                for (int j = first; j < 8; ++j) e[j] = (T)0.0f;
            }
            lds_barrier();
        }
        // ---- S^T = K Q^T - m_ref : sacc[f][kf][r] = log2-score of (query q0+16f+lr, key kv0 + att_key_of(kf, 4lq + r)) relative to
        // the query's reference ----
        f32x4 sacc[QF][4];
#pragma unroll
        for (int kf = 0; kf < 4; ++kf) {
            chunk_t ka[2];
#pragma unroll
            for (int kg = 0; kg < 2; ++kg) ka[kg] = *(const chunk_t*)(Ks + att_koff(att_key_of(kf, lr), kg * 4 + lq));
#pragma unroll
            for (int f = 0; f < QF; ++f) {
                f32x4 a = negm[f];
#pragma unroll
                for (int kg = 0; kg < 2; ++kg) a = mma_chunk(ka[kg], qfr[f][kg], a);
                sacc[f][kf] = a;
            }
        }
        // the key masks (key tail, causal) are only evaluated on the tiles that can be masked at all
        if ((kv0 + BKV > p.tk) || (p.causal && kv0 + BKV - 1 > q0)) {
#pragma unroll
            for (int f = 0; f < QF; ++f)
#pragma unroll
                for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kv0 + att_key_of(kf, lq * 4 + r);
                        if (key >= p.tk || (p.causal && key > q0 + f * 16 + lr)) sacc[f][kf][r] = -1e30f;
                    }
        }
        float mt[QF];
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            mt[f] = -1e30f;
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) mt[f] = fmaxf(mt[f], sacc[f][kf][r]);
        }
        // ---- slow path (first tile; some score more than 2^LAZY above its reference): move the references ----
        if (t == t_begin || wave_any(fmaxf(mt[0], mt[QF - 1]) > LAZY)) {
#pragma unroll
            for (int f = 0; f < QF; ++f) {
                const float m = quad_max(mt[f]);
                const float m_new = m_ref[f] + (t == t_begin ? m : fmaxf(m, 0.f));
                const float dd = m_new - m_ref[f];
#pragma unroll
                for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sacc[f][kf][r] -= dd;
                if (t > t_begin) {
                    const float alpha = exp2_fast(-dd);
#pragma unroll
                    for (int i = 0; i < 4; ++i) oacc[f][i] *= alpha;
                    lacc[f] *= alpha;
                }
                m_ref[f] = m_new;
                negm[f] = f32x4{-m_new, -m_new, -m_new, -m_new};
            }
        }
#pragma unroll
        for (int f = 0; f < QF; ++f)
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) sacc[f][kf][r] = exp2_fast(sacc[f][kf][r]);
        // ---- O^T += V^T P^T, row sums += 1 P^T : the lane's 8 probabilities of group g are keys kv0 + 32g + 8lq .. +7 ----
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            chunk_t pb[QF];
#pragma unroll
            for (int f = 0; f < QF; ++f) {
                pb[f] = pack_p(sacc[f][2 * g], sacc[f][2 * g + 1], T());
                lacc[f] = mma_chunk(ones, pb[f], lacc[f]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const chunk_t a = *(const chunk_t*)(Vs + att_off<8>(i * 16 + lr, g * 4 + lq));
#pragma unroll
                for (int f = 0; f < QF; ++f) oacc[f][i] = mma_chunk(a, pb[f], oacc[f][i]);
            }
        }
    }
    // ---- finalize: every lane holds the row sum of its query; normalise, store 4 consecutive d per lane ----
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        const int qi = q0 + f * 16 + lr;
        if constexpr (SPLIT) {
            // partial results: ws = [G][tq][64] fp32 unnormalised O, then [G][tq][2] = (reference in log2 units, row sum); an empty split
            // (more splits than key tiles) contributes weight 0
            const int64_t G = (int64_t)p.batch * p.heads * p.ksplit, gq = (((int64_t)b * p.heads + h) * p.ksplit + ks) * p.tq + qi;
            if (qi < p.tq) {
                float* wo = (float*)p.ws + gq * D;
#pragma unroll
                for (int i = 0; i < 4; ++i) *(f32x4*)(wo + i * 16 + lq * 4) = oacc[f][i];
                if (lq == 0) {
                    float* ml = (float*)p.ws + G * p.tq * D + gq * 2;
                    ml[0] = t_begin < ntile ? m_ref[f] : -1e30f;
                    ml[1] = lacc[f][0];
                }
            }
            continue;
        }
        const float inv = 1.0f / lacc[f][0];
        if (qi < p.tq) {
            T* op = (T*)p.o + (int64_t)b * p.o_bs + (int64_t)qi * p.ldo + (int64_t)h * D;
            typedef T tx4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                tx4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = from_f32<T>(oacc[f][i][r] * inv);
                *(tx4*)(op + i * 16 + lq * 4) = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Wide-head flash attention for the 16-bit types: ONE head of width D = 512, the AutoencoderKL mid-block attention
// (diffusers Attention with heads = 1 over the 64x64 latent plane: T = 4096 at 512x512).  Replaces the materialised
// scores / softmax / p.v chain (fp32 [B][T][T] scores through HBM) with the same swapped-matmul scheme as above:
//   * a wave owns 16 queries: Q^T fragments (D/32 chunks = 64 VGPRs) and the whole O^T accumulator (D/16 fragments =
//     128 VGPRs) live in registers; workgroup = 8 waves = 128 queries, 2 waves / SIMD;
//   * key tiles of 32: K tile [32 keys][D] (32 KiB, rows of D/8 chunks, chunk ^= row & 15) and V^T tile [D][32 keys]
//     (32 KiB, rows of 4 chunks, chunk ^= (row >> 2) & 3) arrive by global_load_lds_dwordx4, double buffered:
//     tile t+1 is requested right after the barrier that opens tile t, and waited for one whole tile later;
//   * every fragment read feeds one MFMA here (16 queries per wave), so LDS read bandwidth and the matrix pipe are
//     balanced by construction; the O^T rescale is skipped when no running max moved in the wave (the common case
//     after the first tiles).
// SPLIT (p.ksplit > 1, p.ws): the key tiles are divided among ksplit workgroups per query tile ("flash decoding"): a batch-1
// forward has 32 query tiles -- 32 of 256 CUs, all on one XCD with the group-per-XCD order -- and 128 serial key tiles each.
// Split workgroup (b, h, ks) runs tiles ks*per .. and writes its UNNORMALISED O^T (fp32) plus the running (max, sum) of every
// query to p.ws; attention_combine_kernel merges the ksplit partials.  A (b, h, ks) triple is a "group" of the XCD-aware order,
// so the splits of one head land on different XCDs and each L2 streams only its share of K / V^T.
template <typename T, int D, int QF, int NW, bool SPLIT = false>
__global__ __launch_bounds__(NW * 64, (NW + 3) / 4) void attention_wide_kernel(const i2i_attention_params p) {
    typedef typename Elem<T>::chunk_t chunk_t;
    static_assert(Elem<T>::EPC == 8 && D % 128 == 0, "16-bit types, D a multiple of 128");
    constexpr int BKV = 32;
    constexpr int CPRK = D / 8;                     // 16-byte chunks per K row
    constexpr int KC = D / 32, DF = D / 16;         // k-chunks over d (S^T), output fragments over d (O^T)
    constexpr int KTILE = BKV * D * 2, VTILE = D * 64, STAGE = KTILE + VTILE;
    constexpr int OPS = (BKV + D / 16) / NW;        // one-KiB DMA ops per wave and stage
    constexpr int LOOK = 4;                         // fragment reads in flight ahead of the MFMAs that consume them
    constexpr int BQ = NW * QF * 16;                // queries per workgroup
    static_assert((BKV + D / 16) % NW == 0, "");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    int qt, h, b, ks = 0;
    if constexpr (!SPLIT) {
        if (!att_group_of_block((p.tq + BQ - 1) / BQ, p.heads, p.batch, qt, h, b)) return;
    } else {
        int hs;
        if (!att_group_of_block((p.tq + BQ - 1) / BQ, p.heads * p.ksplit, p.batch, qt, hs, b)) return;
        h = hs / p.ksplit;
        ks = hs - h * p.ksplit;
    }
    const int q0 = qt * BQ + wave * (QF * 16);

    const T* qp = (const T*)p.q + (int64_t)b * p.q_bs + (int64_t)h * D;
    const char* kp = (const char*)((const T*)p.k + (int64_t)b * p.k_bs + (int64_t)h * D);
    const char* vp = (const char*)((const T*)p.vt + (int64_t)b * p.vt_bs + (int64_t)h * D * p.ldvt);
    const int ntile_all = (p.tk + BKV - 1) / BKV;
    int t_begin = 0, ntile = ntile_all;                   // this workgroup's key tiles: t_begin .. ntile - 1
    if constexpr (SPLIT) {
        const int per = (ntile_all + p.ksplit - 1) / p.ksplit;
        t_begin = ks * per;
        ntile = t_begin + per < ntile_all ? t_begin + per : ntile_all;
    }
    auto voff = [](int row, int c) __attribute__((always_inline)) { return row * 64 + ((c ^ ((row >> 2) & 3)) << 4); };

    // op pc = q*NW + wave; pc < BKV: K row pc (one key, D/8 chunks = 64 lanes), else 16 V^T rows (4 chunks each).
    // Sources are uniform base + 32-bit lane offset (no per-lane 64-bit pointers: the O^T accumulator needs the
    // registers).  Keys past tk: the K row is clamped to the last key (its scores are masked), a V^T chunk entirely
    // past tk re-reads the row's chunk 0 (finite, and its probabilities are exact zeros; the host checks tk >= 8).
    auto dma_tile = [&](int t, int stage) __attribute__((always_inline)) {
        const int kv0 = t * BKV;
        char* dst = i2i_smem + stage * STAGE;
        int ln = lane;
#ifndef I2I_EMU
        asm volatile("" : "+v"(ln));     // opaque: the per-op lane offsets are recomputed every tile (a few VALU) instead of
                                         // being hoisted out of the loop into registers the fragment look-ahead needs
#endif
#pragma unroll
        for (int q = 0; q < OPS; ++q) {
            const int pc = q * NW + wave;
            if (pc < BKV) {
                int key = kv0 + pc;
                key = key < p.tk ? key : p.tk - 1;
                const unsigned sc = (unsigned)(ln ^ (pc & 15));          // source chunk of physical chunk `lane` (att_off<CPRK>)
                glds16(kp + ((unsigned)key * (unsigned)p.ldk + sc * 8u) * (unsigned)sizeof(T), dst + pc * 1024);
            } else {
                const int row = (pc - BKV) * 16 + (ln >> 2);
                const int sc = (ln & 3) ^ ((row >> 2) & 3);
                int key0 = kv0 + sc * 8;
                key0 = key0 < p.tk ? key0 : 0;
                glds16(vp + ((unsigned)row * (unsigned)p.ldvt + (unsigned)key0) * (unsigned)sizeof(T), dst + pc * 1024);
            }
        }
    };
    static_assert(CPRK == 64, "one K row per DMA op (64 lanes x 16 bytes)");

    // K fragment (key row 16kf + lr, chunk 4kc + lq) sits at physical chunk (4kc + lq) ^ lr = 16(kc>>2) + 4((kc&3) ^ (lr>>2))
    // + (lq ^ (lr&3)): four per-lane bases, everything else is an immediate
    int kb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) kb[j] = lr * (CPRK * 16) + ((((j ^ (lr >> 2)) << 2) + (lq ^ (lr & 3))) << 4);
    chunk_t qfr[QF][KC];
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        const int qi = q0 + f * 16 + lr;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
            qfr[f][kc] = (qi < p.tq) ? *(const chunk_t*)(qp + (int64_t)qi * p.ldq + (kc * 4 + lq) * 8) : zero_chunk<T>();
    }
#pragma unroll
    for (int f = 0; f < QF; ++f)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) reg_fence(qfr[f][kc]);     // Q loads retired before the loop (see attention_dma_kernel)
    f32x4 oacc[QF][DF];
#pragma unroll
    for (int f = 0; f < QF; ++f)
#pragma unroll
        for (int i = 0; i < DF; ++i) oacc[f][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[QF], l_run[QF];
#pragma unroll
    for (int f = 0; f < QF; ++f) { m_run[f] = -1e30f; l_run[f] = 0.f; }
    const float c2 = p.scale * 1.44269504088896341f;

    if (t_begin < ntile) dma_tile(t_begin, 0);
    for (int t = t_begin; t < ntile; ++t) {
        const int st = (t - t_begin) & 1, kv0 = t * BKV;
        wait_vmcnt<0>();                 // tile t has landed (requested one tile ago)
        lds_barrier();                   // ... for everyone, and everyone is done reading tile t-1
        if (t + 1 < ntile) dma_tile(t + 1, st ^ 1);
        const char* Ks = i2i_smem + st * STAGE;
        const char* Vs = Ks + KTILE;
        if (kv0 + BKV > p.tk && (p.tk & 7)) {              // last tile, partially valid V^T chunk: clean its tail in LDS
            const int kc = (p.tk - kv0) >> 3, first = (p.tk - kv0) & 7;
            for (int row = tid; row < D; row += NW * 64) {
                T* e = (T*)(i2i_smem + st * STAGE + KTILE + voff(row, kc));
                for (int j = first; j < 8; ++j) e[j] = (T)0.0f;
            }
            lds_barrier();
        }
        // ---- S^T = K Q^T : sacc[f][kf][r] = S[query q0+16f+lr][key kv0 + 16kf + 4lq + r] ----
        // (2*QF independent accumulation chains, interleaved; every K fragment read feeds QF MFMAs; reads run LOOK
        // fragments ahead, pinned so the compiler neither serialises read -> wait -> MFMA nor hoists a whole tile)
        f32x4 sacc[QF][2];
#pragma unroll
        for (int f = 0; f < QF; ++f) sacc[f][0] = sacc[f][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            // explicit software pipeline: fragment n = (kc = n>>1, kf = n&1) is read LOOK fragments before its MFMAs
            auto kread = [&](int n) __attribute__((always_inline)) -> chunk_t {
                const int kc = n >> 1, kf = n & 1;
                return *(const chunk_t*)(Ks + (kb[kc & 3] + kf * 16 * CPRK * 16 + (kc >> 2) * 256));
            };
            chunk_t fr[LOOK];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < LOOK; ++n) fr[n] = kread(n);
            __builtin_amdgcn_sched_group_barrier(0x100, LOOK, 0);
#pragma unroll
            for (int n = 0; n < 2 * KC; ++n) {
#pragma unroll
                for (int f = 0; f < QF; ++f) sacc[f][n & 1] = mma_chunk(fr[n % LOOK], qfr[f][n >> 1], sacc[f][n & 1]);
                __builtin_amdgcn_sched_group_barrier(0x008, QF, 0);
                if (n + LOOK < 2 * KC) {
                    fr[n % LOOK] = kread(n + LOOK);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- online softmax (lane-local per query, two xor-shuffles for the max) ----
        const bool tail = kv0 + BKV > p.tk;
        float alpha[QF];
        bool moved = false;
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            float mt = -1e30f;
#pragma unroll
            for (int kf = 0; kf < 2; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float sv = sacc[f][kf][r] * c2;
                    if (tail && kv0 + kf * 16 + lq * 4 + r >= p.tk) sv = -1e30f;
                    sacc[f][kf][r] = sv;
                    mt = fmaxf(mt, sv);
                }
            mt = quad_max(mt);
            const float m_new = fmaxf(m_run[f], mt);
            alpha[f] = exp2_fast(m_run[f] - m_new);
            float psum = 0.f;
#pragma unroll
            for (int kf = 0; kf < 2; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = exp2_fast(sacc[f][kf][r] - m_new);
                    sacc[f][kf][r] = pv;
                    psum += pv;
                }
            l_run[f] = l_run[f] * alpha[f] + psum;
            moved = moved || (m_new != m_run[f]);
            m_run[f] = m_new;
        }
        if (wave_any(moved)) {           // wave-uniform: rescale O^T only when some query's running max moved
#pragma unroll
            for (int f = 0; f < QF; ++f)
#pragma unroll
                for (int i = 0; i < DF; ++i) oacc[f][i] *= alpha[f];
        }
        // ---- O^T += V^T P^T : keys 4lq..+3 and 16 + 4lq..+3 of V^T row 16i + lr ----
        chunk_t pb[QF];
#pragma unroll
        for (int f = 0; f < QF; ++f) pb[f] = pack_p(sacc[f][0], sacc[f][1], T());
        {
            union vfrag { chunk_t c; uint64_t u[2]; };
            auto vread = [&](int i) __attribute__((always_inline)) -> chunk_t {
                const int row = i * 16 + lr;
                vfrag a;
                a.u[0] = *(const uint64_t*)(Vs + voff(row, lq >> 1) + (lq & 1) * 8);
                a.u[1] = *(const uint64_t*)(Vs + voff(row, 2 + (lq >> 1)) + (lq & 1) * 8);
                return a.c;
            };
            chunk_t fr[LOOK];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < LOOK; ++n) fr[n] = vread(n);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * LOOK, 0);
#pragma unroll
            for (int i = 0; i < DF; ++i) {
#pragma unroll
                for (int f = 0; f < QF; ++f) oacc[f][i] = mma_chunk(fr[i % LOOK], pb[f], oacc[f][i]);
                __builtin_amdgcn_sched_group_barrier(0x008, QF, 0);
                if (i + LOOK < DF) {
                    fr[i % LOOK] = vread(i + LOOK);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        float l_tot = l_run[f];
        l_tot += __shfl_xor(l_tot, 16);
        l_tot += __shfl_xor(l_tot, 32);
        const float inv = 1.0f / l_tot;
        const int qi = q0 + f * 16 + lr;
        if constexpr (SPLIT) {
            // partial results: ws = [G][tq][D] fp32 unnormalised O, then [G][tq][2] = (running max in scaled log2 units, sum)
            const int64_t G = (int64_t)p.batch * p.heads * p.ksplit, gq = (((int64_t)b * p.heads + h) * p.ksplit + ks) * p.tq + qi;
            if (qi < p.tq) {
                float* wo = (float*)p.ws + gq * D;
#pragma unroll
                for (int i = 0; i < DF; ++i) *(f32x4*)(wo + i * 16 + lq * 4) = oacc[f][i];
                if (lq == 0) {
                    float* ml = (float*)p.ws + G * p.tq * D + gq * 2;
                    ml[0] = m_run[f];
                    ml[1] = l_tot;
                }
            }
            continue;
        }
        if (qi < p.tq) {
            T* op = (T*)p.o + (int64_t)b * p.o_bs + (int64_t)qi * p.ldo + (int64_t)h * D;
            typedef T tx4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int i = 0; i < DF; ++i) {
                tx4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = from_f32<T>(oacc[f][i][r] * inv);
                *(tx4*)(op + i * 16 + lq * 4) = o;
            }
        }
    }
}

// Merge of the key-split partials: one wave per query, a lane owns D / 64 = 8 channels.  With m_s the running max of split s
// (scaled log2 units) and M = max_s m_s:  O = sum_s 2^(m_s - M) O_s / sum_s 2^(m_s - M) l_s.  An empty split has l = 0, O = 0.
template <typename T, int D>
__global__ __launch_bounds__(256) void attention_combine_kernel(const i2i_attention_params p) {
    static_assert(D == 512, "8 channels per lane");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t nq = (int64_t)p.batch * p.heads * p.tq, q = (int64_t)blockIdx.x * 4 + wave;
    if (q >= nq) return;
    const int64_t bh = q / p.tq;
    const int qi = (int)(q - bh * p.tq), b = (int)(bh / p.heads), h = (int)(bh - (int64_t)b * p.heads);
    const int S = p.ksplit;
    const float* part = (const float*)p.ws;
    const float* ml = part + (int64_t)p.batch * p.heads * S * p.tq * D;
    float M = -1e30f;
    for (int s = 0; s < S; ++s) M = fmaxf(M, ml[((bh * S + s) * p.tq + qi) * 2]);
    float L = 0.f, acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < S; ++s) {
        const int64_t gq = (bh * S + s) * p.tq + qi;
        const float w = exp2_fast(ml[gq * 2] - M);
        L += w * ml[gq * 2 + 1];
        const f32x4 v0 = *(const f32x4*)(part + gq * D + lane * 8), v1 = *(const f32x4*)(part + gq * D + lane * 8 + 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc[r] += w * v0[r]; acc[4 + r] += w * v1[r]; }
    }
    const float inv = 1.0f / L;
    typedef T tx8 __attribute__((ext_vector_type(8)));
    tx8 o;
#pragma unroll
    for (int r = 0; r < 8; ++r) o[r] = from_f32<T>(acc[r] * inv);
    *(tx8*)((T*)p.o + (int64_t)b * p.o_bs + (int64_t)qi * p.ldo + (int64_t)h * D + lane * 8) = o;
}

template <typename T>
int launch_att_wide(const i2i_attention_params& p, hipStream_t s) {
    constexpr int QF = 1, NW = 8;                  // 8 waves x 16 queries, two waves per SIMD (256 registers each)
    const int nqt = (p.tq + NW * QF * 16 - 1) / (NW * QF * 16);
    const size_t smem = (size_t)2 * (32 * 512 * 2 + 512 * 64);
    if (p.ksplit > 1) {
        hipLaunchKernelGGL((attention_wide_kernel<T, 512, QF, NW, true>), dim3(att_grid(nqt, p.heads * p.ksplit, p.batch)), dim3(NW * 64), smem, s, p);
        int rc = i2i::check_launch("attention_wide(split)");
        if (rc != I2I_OK) return rc;
        const int64_t nq = (int64_t)p.batch * p.heads * p.tq;
        hipLaunchKernelGGL((attention_combine_kernel<T, 512>), dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s, p);
        return i2i::check_launch("attention_combine");
    }
    hipLaunchKernelGGL((attention_wide_kernel<T, 512, QF, NW>), dim3(att_grid(nqt, p.heads, p.batch)), dim3(NW * 64), smem, s, p);
    return i2i::check_launch("attention_wide");
}

// Merge of the key-split partials of the d = 64 kernel: 8 lanes per query, a lane owns 8 channels (see attention_combine_kernel).
template <typename T>
__global__ __launch_bounds__(256) void attention_combine64_kernel(const i2i_attention_params p) {
    constexpr int D = 64;
    const int64_t nq = (int64_t)p.batch * p.heads * p.tq, idx = (int64_t)blockIdx.x * 256 + threadIdx.x, q = idx >> 3;
    const int c8 = (int)(idx & 7) * 8;
    if (q >= nq) return;
    const int64_t bh = q / p.tq;
    const int qi = (int)(q - bh * p.tq), b = (int)(bh / p.heads), h = (int)(bh - (int64_t)b * p.heads);
    const int S = p.ksplit;
    const float* part = (const float*)p.ws;
    const float* ml = part + (int64_t)p.batch * p.heads * S * p.tq * D;
    float M = -1e30f;
    for (int s = 0; s < S; ++s) M = fmaxf(M, ml[((bh * S + s) * p.tq + qi) * 2]);
    float L = 0.f, acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < S; ++s) {
        const int64_t gq = (bh * S + s) * p.tq + qi;
        const float w = exp2_fast(ml[gq * 2] - M);
        L += w * ml[gq * 2 + 1];
        const f32x4 v0 = *(const f32x4*)(part + gq * D + c8), v1 = *(const f32x4*)(part + gq * D + c8 + 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc[r] += w * v0[r]; acc[4 + r] += w * v1[r]; }
    }
    const float inv = 1.0f / L;
    typedef T tx8 __attribute__((ext_vector_type(8)));
    tx8 o;
#pragma unroll
    for (int r = 0; r < 8; ++r) o[r] = from_f32<T>(acc[r] * inv);
    *(tx8*)((T*)p.o + (int64_t)b * p.o_bs + (int64_t)qi * p.ldo + (int64_t)h * D + c8) = o;
}

template <typename T>
int launch_att_dma(const i2i_attention_params& p, hipStream_t s) {
    const int S = p.ksplit > 1 ? p.ksplit : 1;
    const unsigned g2 = att_grid((p.tq + 127) / 128, p.heads * S, p.batch);
    const size_t smem = (size_t)3 * 2 * 64 * 128;
    const char* e = getenv("I2I_ATT_QF");                  // test / A-B hook: force the 128-query (2) or the 64-query (1) workgroup
    const int force = e ? atoi(e) : 0;
    if (S > 1) {
        if (force == 2 || (force != 1 && g2 >= 384)) hipLaunchKernelGGL((attention_dma_kernel<T, 2, true>), dim3(g2), dim3(256), smem, s, p);
        else hipLaunchKernelGGL((attention_dma_kernel<T, 1, true>), dim3(att_grid((p.tq + 63) / 64, p.heads * S, p.batch)), dim3(256), smem, s, p);
        const int rc = i2i::check_launch("attention_dma(split)");
        if (rc != I2I_OK) return rc;
        const int64_t nthr = (int64_t)p.batch * p.heads * p.tq * 8;
        hipLaunchKernelGGL((attention_combine64_kernel<T>), dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, s, p);
        return i2i::check_launch("attention_combine64");
    }
    if (force == 2 || (force != 1 && g2 >= 384)) hipLaunchKernelGGL((attention_dma_kernel<T, 2>), dim3(g2), dim3(256), smem, s, p);
    else hipLaunchKernelGGL((attention_dma_kernel<T, 1>), dim3(att_grid((p.tq + 63) / 64, p.heads, p.batch)), dim3(256), smem, s, p);
    return i2i::check_launch("attention_dma");
}

template <typename T>
int launch_att(const i2i_attention_params& p, hipStream_t s) {
    const dim3 grid((unsigned)((p.tq + 63) / 64), (unsigned)p.heads, (unsigned)p.batch);
    const size_t smem = 2 * 64 * 64 * sizeof(T);
    hipLaunchKernelGGL((attention_kernel<T, 64>), grid, dim3(256), smem, s, p);
    return i2i::check_launch("attention");
}

}  // namespace

extern "C" int i2i_attention(const i2i_attention_params* p, int dtype, void* stream) {
    if (!p || !p->q || !p->k || !p->vt || !p->o) return i2i::fail(I2I_ERR_BAD_ARG, "attention: null pointer");
    if (p->d != 64 && !(p->d == 512 && dtype != I2I_F32))
        return i2i::fail(I2I_ERR_UNSUPPORTED, "attention: fused kernels support head dim 64, and 512 for the 16-bit types (got %d)", p->d);
    const int epc = dtype == I2I_F32 ? 4 : 8;
    if (p->ldq % epc || p->ldk % epc || p->ldvt % epc || p->ldo % 4 || p->tk < 1 || p->tq < 1)
        return i2i::fail(I2I_ERR_BAD_ARG, "attention: bad leading dims");
    if (p->ldvt < ((p->tk + epc - 1) / epc) * epc) return i2i::fail(I2I_ERR_BAD_ARG, "attention: ldvt must cover tk rounded up to a chunk");
    if (!(p->scale > 0.f)) return i2i::fail(I2I_ERR_BAD_ARG, "attention: scale must be positive");
    if (p->causal && p->tq != p->tk) return i2i::fail(I2I_ERR_BAD_ARG, "attention: causal needs tq == tk");
    hipStream_t s = (hipStream_t)stream;
    // 16-bit types: LDS-DMA kernel (needs 8-byte aligned output rows for its vector stores); f32 parity mode: the
    // register-staged kernel
    const bool dma_ok = (p->ldo % 4 == 0) && (p->o_bs % 4 == 0) && (((uintptr_t)p->o & 7) == 0) && (((uintptr_t)p->k | (uintptr_t)p->vt) & 15) == 0 &&
                        (p->ldk % 8 == 0) && (p->k_bs % 8 == 0) && (p->vt_bs % 8 == 0);
    if (p->ksplit > 1 && (dtype == I2I_F32 || !p->ws || (((uintptr_t)p->ws) & 15) || p->ldo % 8 || p->o_bs % 8 || (((uintptr_t)p->o) & 15) ||
                          (p->d == 64 && (!dma_ok || p->causal))))
        return i2i::fail(I2I_ERR_UNSUPPORTED, "attention: ksplit is implemented by the 16-bit LDS-DMA kernels (no causal mask) and needs a 16-byte aligned workspace and output rows");
    if (p->d == 512) {
        if (!dma_ok || p->causal || p->ldq % 8 || (((uintptr_t)p->q) & 15) || p->q_bs % 8 || p->tk < 8 ||
            (int64_t)p->tk * p->ldk * 2 >= (int64_t(1) << 32) || (int64_t)p->d * p->heads * p->ldvt * 2 >= (int64_t(1) << 32))
            return i2i::fail(I2I_ERR_UNSUPPORTED, "attention: the d = 512 kernel needs 16-byte aligned q / k / v^T rows, tk >= 8, "
                                                  "per-batch K / V^T under 4 GiB and no causal mask");
        return dtype == I2I_BF16 ? launch_att_wide<__bf16>(*p, s) : launch_att_wide<_Float16>(*p, s);
    }
    switch (dtype) {
        case I2I_F32: return launch_att<float>(*p, s);
        case I2I_BF16: return dma_ok ? launch_att_dma<__bf16>(*p, s) : launch_att<__bf16>(*p, s);
        case I2I_F16: return dma_ok ? launch_att_dma<_Float16>(*p, s) : launch_att<_Float16>(*p, s);
    }
    return i2i::fail(I2I_ERR_BAD_ARG, "attention: bad dtype");
}
