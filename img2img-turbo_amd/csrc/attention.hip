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
                if (key >= p.tk) s = -1e30f;
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
    if (p->d != 64) return i2i::fail(I2I_ERR_UNSUPPORTED, "attention: fused kernel supports head dim 64 only (got %d)", p->d);
    const int epc = dtype == I2I_F32 ? 4 : 8;
    if (p->ldq % epc || p->ldk % epc || p->ldvt % epc || p->ldo % 4 || p->tk < 1 || p->tq < 1)
        return i2i::fail(I2I_ERR_BAD_ARG, "attention: bad leading dims");
    if (p->ldvt < ((p->tk + epc - 1) / epc) * epc) return i2i::fail(I2I_ERR_BAD_ARG, "attention: ldvt must cover tk rounded up to a chunk");
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case I2I_F32: return launch_att<float>(*p, s);
        case I2I_BF16: return launch_att<__bf16>(*p, s);
        case I2I_F16: return launch_att<_Float16>(*p, s);
    }
    return i2i::fail(I2I_ERR_BAD_ARG, "attention: bad dtype");
}
