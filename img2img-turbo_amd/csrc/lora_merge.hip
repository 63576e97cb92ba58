// Device-side LoRA re-merge (HBM-bound): W' = (W0 + r * B.A) * g written straight into the packed weight tensor the
// forward kernels read, so a new LoRA scale / skip gamma r (src/pix2pix_turbo.py:206-207,211,217: set_adapters([..], [r]),
// decoder.gamma = r) costs one streaming pass over the adapted layers instead of a host re-pack + upload.
//   w0  fp32 master of the layer in PACKED layout [N][K] (K = kh*kw*cin_padded, k contiguous)
//   a   fp32 [rank][K]: lora_A in the same k order, adapter scaling lora_alpha/rank folded in; several adapters on one
//       layer are concatenated along rank
//   b   fp32 [N][rank]: lora_B
//   rg  device float[2] = (r, gamma): read at run time so the merge program itself never changes
// One thread owns 4 consecutive k of ROWS rows: the A chunk is loaded once per j for all rows, the B values are
// wave-uniform (scalar loads).  fp32 FMA chain in j order, then one rounding to `T`.
#include "i2i_dev.h"
#include "launch.h"

namespace {

constexpr int LM_ROWS = 4;

template <typename T>
__global__ __launch_bounds__(256) void lora_merge_kernel(const i2i_lora_merge_params p) {
    const int k = ((int)blockIdx.x * 256 + (int)threadIdx.x) * 4;
    if (k >= p.K) return;
    const int n0 = (int)blockIdx.y * LM_ROWS;
    const float r = p.rg ? p.rg[0] : 1.f;
    const float g = (p.rg && p.use_gamma) ? p.rg[1] : 1.f;
    float acc[LM_ROWS][4];
#pragma unroll
    for (int i = 0; i < LM_ROWS; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
    for (int j = 0; j < p.rank; ++j) {
        const f32x4 a = *(const f32x4*)(p.a + (int64_t)j * p.K + k);
#pragma unroll
        for (int i = 0; i < LM_ROWS; ++i) {
            const int n = n0 + i < p.N ? n0 + i : p.N - 1;
            const float bv = p.b[(int64_t)n * p.rank + j];
            acc[i][0] = fmaf(bv, a[0], acc[i][0]);
            acc[i][1] = fmaf(bv, a[1], acc[i][1]);
            acc[i][2] = fmaf(bv, a[2], acc[i][2]);
            acc[i][3] = fmaf(bv, a[3], acc[i][3]);
        }
    }
#pragma unroll
    for (int i = 0; i < LM_ROWS; ++i) {
        const int n = n0 + i;
        if (n >= p.N) break;
        const f32x4 w = *(const f32x4*)(p.w0 + (int64_t)n * p.K + k);
        T* d = (T*)p.dst + (int64_t)n * p.K + k;
        d[0] = from_f32<T>((w[0] + r * acc[i][0]) * g);
        d[1] = from_f32<T>((w[1] + r * acc[i][1]) * g);
        d[2] = from_f32<T>((w[2] + r * acc[i][2]) * g);
        d[3] = from_f32<T>((w[3] + r * acc[i][3]) * g);
    }
}

// LayerNorm-fold form (i2i_lora_merge_params.kscale ..): one workgroup owns LM_ROWS whole rows and walks K in 1024-element trips, so that the
// row sums the consumer GEMM needs -- colsum[n] = sum_k of the STORED dst[n][k], bias_out[n] = bias0[n] + sum_k w'[n][k] * kshift[k] -- are
// reduced inside the block in a fixed order (thread partials in k order, then 256 partials in thread order through LDS).
template <typename T>
__global__ __launch_bounds__(256) void lora_merge_lnf_kernel(const i2i_lora_merge_params p) {
    const int tid = (int)threadIdx.x;
    const int n0 = (int)blockIdx.y * LM_ROWS;
    const float r = p.rg ? p.rg[0] : 1.f;
    const float g = (p.rg && p.use_gamma) ? p.rg[1] : 1.f;
    float cs[LM_ROWS], bs[LM_ROWS];
#pragma unroll
    for (int i = 0; i < LM_ROWS; ++i) { cs[i] = 0.f; bs[i] = 0.f; }
    for (int k = tid * 4; k < p.K; k += 1024) {
        float acc[LM_ROWS][4];
#pragma unroll
        for (int i = 0; i < LM_ROWS; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
        for (int j = 0; j < p.rank; ++j) {
            const f32x4 a = *(const f32x4*)(p.a + (int64_t)j * p.K + k);
#pragma unroll
            for (int i = 0; i < LM_ROWS; ++i) {
                const int n = n0 + i < p.N ? n0 + i : p.N - 1;
                const float bv = p.b[(int64_t)n * p.rank + j];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][e] = fmaf(bv, a[e], acc[i][e]);
            }
        }
        const f32x4 ks = *(const f32x4*)(p.kscale + k), kh = *(const f32x4*)(p.kshift + k);
#pragma unroll
        for (int i = 0; i < LM_ROWS; ++i) {
            const int n = n0 + i;
            if (n >= p.N) break;
            const f32x4 w = *(const f32x4*)(p.w0 + (int64_t)n * p.K + k);
            T* d = (T*)p.dst + (int64_t)n * p.K + k;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float wm = (w[e] + r * acc[i][e]) * g;
                const T st = from_f32<T>(wm * ks[e]);
                d[e] = st;
                cs[i] += to_f32<T>(st);
                bs[i] = fmaf(wm, kh[e], bs[i]);
            }
        }
    }
    float* red = (float*)i2i_smem;                // [256][LM_ROWS][2]
#pragma unroll
    for (int i = 0; i < LM_ROWS; ++i) { red[(tid * LM_ROWS + i) * 2] = cs[i]; red[(tid * LM_ROWS + i) * 2 + 1] = bs[i]; }
    __syncthreads();
    if (tid < LM_ROWS && n0 + tid < p.N) {
        float C = 0.f, B = 0.f;
        for (int t = 0; t < 256; ++t) { C += red[(t * LM_ROWS + tid) * 2]; B += red[(t * LM_ROWS + tid) * 2 + 1]; }
        p.colsum[n0 + tid] = C;
        p.bias_out[n0 + tid] = (p.bias0 ? p.bias0[n0 + tid] : 0.f) + B;
    }
}

}  // namespace

extern "C" int i2i_lora_merge(const i2i_lora_merge_params* p, int dtype, void* stream) {
    if (!p || !p->dst || !p->w0) return i2i::fail(I2I_ERR_BAD_ARG, "lora_merge: null operand");
    if (p->rank > 0 && (!p->a || !p->b)) return i2i::fail(I2I_ERR_BAD_ARG, "lora_merge: rank %d without A/B", p->rank);
    if (p->N <= 0 || p->K <= 0 || (p->K & 3) || p->rank < 0) return i2i::fail(I2I_ERR_BAD_ARG, "lora_merge: bad shape N=%d K=%d rank=%d", p->N, p->K, p->rank);
    const dim3 grid((unsigned)((p->K / 4 + 255) / 256), (unsigned)((p->N + LM_ROWS - 1) / LM_ROWS));
    hipStream_t s = (hipStream_t)stream;
    if (p->kscale || p->kshift || p->colsum || p->bias_out) {
        if (!p->kscale || !p->kshift || !p->colsum || !p->bias_out) return i2i::fail(I2I_ERR_BAD_ARG, "lora_merge: the LayerNorm fold needs kscale, kshift, colsum and bias_out");
        if (((uintptr_t)p->kscale | (uintptr_t)p->kshift) & 15) return i2i::fail(I2I_ERR_BAD_ARG, "lora_merge: kscale / kshift must be 16-byte aligned");
        const dim3 gl(1u, grid.y);
        const size_t smem = 256 * LM_ROWS * 2 * sizeof(float);
        switch (dtype) {
            case I2I_F32: hipLaunchKernelGGL((lora_merge_lnf_kernel<float>), gl, dim3(256), smem, s, *p); break;
            case I2I_BF16: hipLaunchKernelGGL((lora_merge_lnf_kernel<__bf16>), gl, dim3(256), smem, s, *p); break;
            case I2I_F16: hipLaunchKernelGGL((lora_merge_lnf_kernel<_Float16>), gl, dim3(256), smem, s, *p); break;
            default: return i2i::fail(I2I_ERR_BAD_ARG, "lora_merge: dtype %d", dtype);
        }
        return i2i::check_launch("lora_merge<ln fold>");
    }
    switch (dtype) {
        case I2I_F32: hipLaunchKernelGGL((lora_merge_kernel<float>), grid, dim3(256), 0, s, *p); break;
        case I2I_BF16: hipLaunchKernelGGL((lora_merge_kernel<__bf16>), grid, dim3(256), 0, s, *p); break;
        case I2I_F16: hipLaunchKernelGGL((lora_merge_kernel<_Float16>), grid, dim3(256), 0, s, *p); break;
        default: return i2i::fail(I2I_ERR_BAD_ARG, "lora_merge: dtype %d", dtype);
    }
    return i2i::check_launch("lora_merge");
}
