// Calibration micro-kernels behind bench.py's `calib` block (not on the forward path): the boxes of the GPU pool differ by +-8 % on the
// same binary, so every bench line carries what ITS box delivers on three elementary loads -- a bare MFMA loop on random
// operands (matrix pipe + power management), a copy stream (HBM) and an empty kernel (launch path: microseconds per hipGraph node).
#include "i2i_dev.h"
#include "launch.h"

namespace {

typedef float cal_f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ cal_f32x16 cal_mma(bf16x8 a, bf16x8 b, cal_f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ cal_f32x16 cal_mma(f16x8 a, f16x8 b, cal_f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

__global__ void nop_kernel(int) {}

// One wave per SIMD (256 workgroups x 4 waves): 8 independent accumulators, operands held in registers for the whole loop.
template <typename T>
__global__ __launch_bounds__(256, 1) void calib_mfma_kernel(const void* operands, float* sink, int iters) {
    typedef typename Elem<T>::chunk_t chunk_t;
    const int tid = threadIdx.x, gid = (int)blockIdx.x * 256 + tid;
    const chunk_t* src = (const chunk_t*)operands;
    chunk_t a[4], b[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = src[(gid * 6 + i) & 4095];
#pragma unroll
    for (int i = 0; i < 2; ++i) b[i] = src[(gid * 6 + 4 + i) & 4095];
    cal_f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = cal_mma(a[i & 3], b[i >> 2], acc[i]);
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[i][r];
    sink[gid] = t;
}

__global__ __launch_bounds__(256) void calib_stream_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

}  // namespace

extern "C" int i2i_nop(void* stream) {
    hipLaunchKernelGGL(nop_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, 0);
    return i2i::check_launch("nop");
}

extern "C" int i2i_calib_mfma(int dtype, int iters, const void* operands, float* sink, void* stream) {
    if (!operands || !sink || iters < 1 || (((uintptr_t)operands) & 15)) return i2i::fail(I2I_ERR_BAD_ARG, "calib_mfma: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case I2I_BF16: hipLaunchKernelGGL((calib_mfma_kernel<__bf16>), dim3(256), dim3(256), 0, s, operands, sink, iters); break;
        case I2I_F16: hipLaunchKernelGGL((calib_mfma_kernel<_Float16>), dim3(256), dim3(256), 0, s, operands, sink, iters); break;
        default: return i2i::fail(I2I_ERR_BAD_ARG, "calib_mfma: 16-bit dtypes only");
    }
    return i2i::check_launch("calib_mfma");
}

extern "C" int i2i_calib_stream(const void* src, void* dst, size_t bytes, void* stream) {
    if (!src || !dst || (bytes & 15) || ((((uintptr_t)src) | ((uintptr_t)dst)) & 15)) return i2i::fail(I2I_ERR_BAD_ARG, "calib_stream: bad arguments");
    const size_t n16 = bytes / 16;
    const size_t want = (n16 + 255) / 256;
    const unsigned grid = (unsigned)(want < 8192 ? (want ? want : 1) : 8192);
    hipLaunchKernelGGL(calib_stream_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const u32x4*)src, (u32x4*)dst, n16);
    return i2i::check_launch("calib_stream");
}
