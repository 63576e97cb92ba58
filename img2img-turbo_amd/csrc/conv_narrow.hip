// 3x3 convolution of a NARROW input (cin <= 8: one 16-byte chunk per pixel) into many channels: the VAE encoder's conv_in
// (diffusers Encoder.conv_in, 3 -> 128 at full resolution, reached from src/model.py:16), replacing F.conv2d there.
//
// Why its own kernel: the launch is a pure WRITE stream -- 537 MB of output for 33 MB of input at batch 8, 512 x 512 -- with 17 GFLOP of
// matrix work that the chip does in 10 us.  On the LDS-DMA igemm it ran as a generic im2col GEMM (K = 72 padded to 128: two K stages of
// per-lane gathers, a 128 x 128 tile epilogue) at 1.6 TB/s of stores (0.34 ms, profiles/r5_per_op_bs8.txt).  Here
//   * a workgroup (4 waves) owns 8 x 32 output pixels x 128 channels; the 10 x 34 halo (5.4 KB) is staged ONCE in LDS, zero padded;
//   * K = 9 taps x 8 channels = 72, padded to 80: five k16 steps of v_mfma_f32_32x32x16 in which a lane's 8 k elements ARE one
//     tap's pixel chunk (tap 2s + lh of step s; tap 9 = zeros), so the B fragment of a step is one ds_read_b128 at
//     (row + ky, column + kx) and the weights' A fragments ([cout][tap][8] rows, 16 bytes per tap) are 20 global loads per lane, once;
//   * operands swapped (A = weight rows) as in the wide kernels: after the half exchange a lane owns 8 consecutive channels of one
//     pixel and stores 16 bytes; 40 MFMAs per wave, then 16 stores per lane;
//   * the GroupNorm partial sums of the STORED tile (the first resnet's norm1) leave with the epilogue like the wide GEMM's
//     (v_dot2c on the stored pairs, rows -> wave through an LDS transpose in a fixed order, one slot per (tile, group)).
// 16-bit types; everything else (and every other shape) stays on the LDS-DMA igemm.
#include <stdlib.h>

#include "i2i_dev.h"
#include "launch.h"

namespace {

typedef float cn_f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ cn_f32x16 cn_mma(bf16x8 a, bf16x8 b, cn_f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ cn_f32x16 cn_mma(f16x8 a, f16x8 b, cn_f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

typedef __bf16 cn_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 cn_f16x2 __attribute__((ext_vector_type(2)));
#ifdef I2I_EMU
__device__ __forceinline__ float cn_dot2(cn_bf16x2 a, cn_bf16x2 b, float c) { return fmaf((float)a[1], (float)b[1], fmaf((float)a[0], (float)b[0], c)); }
__device__ __forceinline__ float cn_dot2(cn_f16x2 a, cn_f16x2 b, float c) { return fmaf((float)a[1], (float)b[1], fmaf((float)a[0], (float)b[0], c)); }
#else
__device__ __forceinline__ float cn_dot2(cn_bf16x2 a, cn_bf16x2 b, float c) { return __builtin_amdgcn_fdot2_f32_bf16(a, b, c, false); }
__device__ __forceinline__ float cn_dot2(cn_f16x2 a, cn_f16x2 b, float c) { return __builtin_amdgcn_fdot2(a, b, c, false); }
#endif

constexpr int CN_TH = 8, CN_TW = 32, CN_BN = 128, CN_HW = CN_TW + 2, CN_HH = CN_TH + 2;
constexpr int CN_HALO_BYTES = ((CN_HH * CN_HW * 16 + 255) / 256) * 256;

template <typename T>
__global__ __launch_bounds__(256, 2) void conv_narrow_kernel(const i2i_igemm_params p) {
    typedef typename Elem<T>::chunk_t chunk_t;
    typedef T tx2 __attribute__((ext_vector_type(2)));
    typedef T tx8 __attribute__((ext_vector_type(8)));
    static_assert(Elem<T>::EPC == 8, "16-bit types only");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int tw = (p.wo + CN_TW - 1) / CN_TW, th = (p.ho + CN_TH - 1) / CN_TH;
    const int tile = blockIdx.x, img = tile / (tw * th), trem = tile - img * (tw * th), ty = trem / tw, tx = trem - ty * tw;
    const int y0 = ty * CN_TH, x0 = tx * CN_TW, n0 = (int)blockIdx.y * CN_BN;

    // ---- halo: (y0 - 1 .. y0 + 8) x (x0 - 1 .. x0 + 32), one 16-byte chunk per pixel, zeros outside the plane
    chunk_t* halo = (chunk_t*)i2i_smem;
    const T* src = (const T*)p.a0 + (int64_t)img * p.hin * p.win * p.lda0;
    for (int c = tid; c < CN_HH * CN_HW; c += 256) {
        const int hr = c / CN_HW, hc = c - hr * CN_HW;
        const int y = y0 - 1 + hr, x = x0 - 1 + hc;
        const bool ok = (unsigned)y < (unsigned)p.hin && (unsigned)x < (unsigned)p.win;
        halo[c] = ok ? *(const chunk_t*)(src + ((int64_t)y * p.win + x) * p.lda0) : zero_chunk<T>();
    }
    // ---- weights: A fragments of the 4 channel blocks x 5 k16 steps (row n0 + 32j + l31, tap 2s + lh; tap 9 does not exist)
    chunk_t wf[4][5];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + j * 32 + l31;
        const T* wr = (const T*)p.b + (int64_t)(n < p.N ? n : p.N - 1) * p.ldb;
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const int t = 2 * s + lh;
            wf[j][s] = t < 9 ? *(const chunk_t*)(wr + t * 8) : zero_chunk<T>();
        }
    }
    cn_f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    __syncthreads();
    // ---- 5 k16 steps: the B fragment of pixel row 2*wave + i is the halo chunk at (row + ky, l31 + kx) of this lane's tap
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int t = 2 * s + lh, tt = t < 9 ? t : 0, ky = tt / 3, kx = tt - ky * 3;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            chunk_t xf = halo[(2 * wave + i + ky) * CN_HW + l31 + kx];
            if (t >= 9) xf = zero_chunk<T>();
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = cn_mma(wf[j][s], xf, acc[i][j]);
        }
    }
    // ---- epilogue.  acc[i][j][r]: pixel (y0 + 2*wave + i, x0 + l31), channel n0 + 32j + 8*(r>>2) + 4*lh + (r&3); after the half exchange
    // of register quads (2pr, 2pr+1) the lane owns channels 32j + 16pr + 8lh .. +7 of its pixel
    T* out = (T*)p.c + (int64_t)img * p.ho * p.wo * p.ldc;
    const bool has_bias = p.bias_mode == 1 && p.bias;
    float gacc[32];                                          // (sum, sum of squares) per 4-channel quad: index ((j*2 + pr)*2 + h)*2 + {0,1}
#pragma unroll
    for (int v = 0; v < 32; ++v) gacc[v] = 0.f;
    tx2 ones;
    ones[0] = (T)1.0f; ones[1] = (T)1.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int n = n0 + j * 32 + 16 * pr + 8 * lh;
            const bool nok = n < p.N;                        // N % 8 == 0 (host check)
            f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
            if (has_bias && nok) { b0 = *(const f32x4*)(p.bias + n); b1 = *(const f32x4*)(p.bias + n + 4); }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f32x4 qa, qb;
#pragma unroll
                for (int r = 0; r < 4; ++r) { qa[r] = acc[i][j][8 * pr + r]; qb[r] = acc[i][j][8 * pr + 4 + r]; }
                float v[8];
                widen_pair(qa, qb, v);                       // wave-wide: before any lane drops out
                const int y = y0 + 2 * wave + i, x = x0 + l31;
                if (!nok || y >= p.ho || x >= p.wo) continue;
                tx8 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    o[r] = from_f32<T>(__builtin_fmaf(p.alpha, v[r], b0[r]));
                    o[4 + r] = from_f32<T>(__builtin_fmaf(p.alpha, v[4 + r], b1[r]));
                }
                *(tx8*)(out + ((int64_t)y * p.wo + x) * p.ldc + n) = o;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    tx2 d0, d1;
                    d0[0] = o[4 * h]; d0[1] = o[4 * h + 1]; d1[0] = o[4 * h + 2]; d1[1] = o[4 * h + 3];
                    const int vi = ((j * 2 + pr) * 2 + h) * 2;
                    gacc[vi] = cn_dot2(d0, ones, gacc[vi]);         gacc[vi] = cn_dot2(d1, ones, gacc[vi]);
                    gacc[vi + 1] = cn_dot2(d0, d0, gacc[vi + 1]);   gacc[vi + 1] = cn_dot2(d1, d1, gacc[vi + 1]);
                }
            }
        }
    if (!p.gn_part) return;
    // ---- GroupNorm partial sums of the stored tile: pixels -> wave through an LDS transpose (lane t adds value t % 32 of the 32 lanes
    // of its half: fixed order), waves -> workgroup -> one slot per (image, tile, group)
    constexpr int LST = 36;                                  // lane stride in floats: 16-byte writes of 16 lanes hit 16 distinct bank quads
    float* st = (float*)(i2i_smem + CN_HALO_BYTES);          // [4][64][LST]
    float* st2 = st + 4 * 64 * LST;                          // [4][32 quads][2]
#pragma unroll
    for (int v = 0; v < 32; v += 4) *(f32x4*)(st + (wave * 64 + lane) * LST + v) = f32x4{gacc[v], gacc[v + 1], gacc[v + 2], gacc[v + 3]};
    __syncthreads();
    {
        const int half = lane >> 5, v = lane & 31;
        float tot = 0.f;
        for (int l = 0; l < 32; ++l) tot += st[(wave * 64 + half * 32 + l) * LST + v];
        const int jpr = v >> 2, h = (v >> 1) & 1, sq = v & 1;
        const int quad = jpr * 4 + half * 2 + h;             // channel quad of the tile: j*8 + pr*4 + lh*2 + h
        st2[(wave * 32 + quad) * 2 + sq] = tot;
    }
    __syncthreads();
    const int groups = p.gn_part_groups, cpg = p.N / groups, ng_tile = CN_BN / cpg;
    const int g = n0 / cpg + tid;
    if (tid < ng_tile && g < groups) {
        const int q0 = (tid * cpg) >> 2, nq = cpg >> 2;
        float S = 0.f, Q = 0.f;
        for (int w = 0; w < 4; ++w)
            for (int q = q0; q < q0 + nq; ++q) { S += st2[(w * 32 + q) * 2]; Q += st2[(w * 32 + q) * 2 + 1]; }
        float* o2 = p.gn_part + (((int64_t)img * (tw * th) + trem) * groups + g) * 2;
        o2[0] = S;
        o2[1] = Q;
    }
}


}  // namespace

namespace i2i {
// What the narrow conv takes: a 16-bit 3x3 stride-1 pad-1 convolution of ONE source with 8 (padded) input channels into a multiple of
// 128 output channels, per-column bias or none, nothing else fused.
bool conv_narrow_eligible(const i2i_igemm_params& p, int dtype) {
    if (dtype != I2I_BF16 && dtype != I2I_F16) return false;
    if (p.ks != 3 || p.stride != 1 || p.pad != 1 || p.ups || p.up_h || p.up_w || p.subpix) return false;
    if (p.c0 != 8 || p.c1 || p.a1 || p.lda0 % 8 || p.K != 72 || p.ldb % 8 || p.ldb < 72) return false;
    if (p.N % CN_BN || p.ldc % 8 || p.ho != p.hin || p.wo != p.win || p.M != p.nimg * p.ho * p.wo) return false;
    if (p.gn_ss || p.act || p.act_out || p.res || p.geglu || p.out_f32 || p.splitk > 1 || p.k2_a || p.ln_cs || p.n_trans || p.c2) return false;
    if (p.zcount > 1 || p.bias_mode == 2 || (p.bias_mode == 1 && (!p.bias || ((uintptr_t)p.bias & 15)))) return false;
    if (((uintptr_t)p.a0 | (uintptr_t)p.b | (uintptr_t)p.c) & 15) return false;
    return true;
}
// tile == 0: the full-resolution planes this kernel was written for (a launch of at least a few hundred tiles); small planes stay on
// the LDS-DMA igemm with its split-K.  I2I_CONV_NARROW=0 switches the route off (A/B hook).
bool conv_narrow_auto(const i2i_igemm_params& p, int dtype) {
    if (!conv_narrow_eligible(p, dtype)) return false;
    const char* e = getenv("I2I_CONV_NARROW");
    if (e && atoi(e) == 0) return false;
    const long tiles = (long)p.nimg * ((p.ho + CN_TH - 1) / CN_TH) * ((p.wo + CN_TW - 1) / CN_TW);
    return tiles >= 256;
}
int conv_narrow_gn_parts(const i2i_igemm_params& p, int dtype, int groups) {
    i2i_igemm_params q = p;
    q.gn_part = nullptr;
    if (!conv_narrow_eligible(q, dtype) || groups < 1 || p.N % groups) return 0;
    const int cpg = p.N / groups;
    if (cpg % 4 || CN_BN % cpg) return 0;
    return ((p.ho + CN_TH - 1) / CN_TH) * ((p.wo + CN_TW - 1) / CN_TW);
}
int conv_narrow(const i2i_igemm_params& p, int dtype, hipStream_t s) {
    const dim3 grid((unsigned)(p.nimg * ((p.ho + CN_TH - 1) / CN_TH) * ((p.wo + CN_TW - 1) / CN_TW)), (unsigned)(p.N / CN_BN)), block(256);
    const size_t smem = CN_HALO_BYTES + (size_t)(4 * 64 * 36 + 4 * 32 * 2) * sizeof(float);
    switch (dtype) {
        case I2I_BF16: hipLaunchKernelGGL((conv_narrow_kernel<__bf16>), grid, block, smem, s, p); break;
        case I2I_F16: hipLaunchKernelGGL((conv_narrow_kernel<_Float16>), grid, block, smem, s, p); break;
        default: return fail(I2I_ERR_BAD_ARG, "conv_narrow: bad dtype");
    }
    return check_launch("conv_narrow");
}
}  // namespace i2i
