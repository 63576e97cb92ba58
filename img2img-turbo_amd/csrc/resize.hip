// Device-side LANCZOS resize of uint8 HWC image batches, bit-identical to Pillow's Image.resize(..., Image.LANCZOS)
// (the reference resizes on the host: src/inference_paired.py:38-41 to a multiple of 8, src/inference_unpaired.py:40,53 to the
// model size and back).  HBM-bound byte work: Pillow's two separable passes (horizontal, then vertical) in 32-bit integer
// arithmetic on 22-bit fixed-point weights; the per-coordinate tap windows and weights come from the host
// (img2img_turbo_amd/image_ops.py restates Pillow's precompute_coeffs / normalize_coeffs_8bpc in double precision).
//   out = clip8((2^21 + sum_t in[first + t] * k[t]) >> 22)
// Horizontal pass: a thread per output pixel, one 8-byte load per tap; vertical pass: a thread per 4 bytes of the flattened row.
#include "i2i_dev.h"
#include "launch.h"

namespace {

__device__ __forceinline__ uint8_t clip8(int32_t acc) {
    const int v = acc >> 22;                                  // arithmetic shift, as Pillow's clip8 table index
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// Horizontal pass: one thread per output pixel.  A tap's c <= 4 channel bytes are fetched as ONE 8-byte load from the enclosing
// aligned dword pair (the tensor base is at least 4-byte aligned) instead of c byte loads.
__global__ __launch_bounds__(256) void resize_h_u8_kernel(const i2i_resize_u8_params p) {
    const int wo = p.nout;
    const int64_t total = (int64_t)p.n * p.hin * wo;
    const uint8_t* __restrict__ base = (const uint8_t*)p.src;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int xo = (int)(i % wo);
        const int64_t row = i / wo;                           // (image, y) flattened
        const int first = p.bounds[2 * xo], cnt = p.bounds[2 * xo + 1];
        const int32_t* __restrict__ k = p.coeffs + (int64_t)xo * p.ksize;
        int64_t a = (row * p.win + first) * p.c;              // byte offset of the first tap's channel 0
        int32_t acc[4];
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) acc[ch] = 1 << 21;
        const int64_t nbytes = (int64_t)p.n * p.hin * p.win * p.c;
        if (a + (int64_t)cnt * p.c + 8 <= nbytes) {
            for (int t = 0; t < cnt; ++t) {
                const uint32_t* q = (const uint32_t*)(base + (a & ~(int64_t)3));
                const uint64_t v = ((uint64_t)q[0] | ((uint64_t)q[1] << 32)) >> (8 * (int)(a & 3));
                const int32_t w = k[t];
#pragma unroll
                for (int ch = 0; ch < 4; ++ch)
                    if (ch < p.c) acc[ch] += (int32_t)((v >> (8 * ch)) & 0xff) * w;
                a += p.c;
            }
        } else {                                              // the last pixels of the batch: no read past the tensor's end
            for (int t = 0; t < cnt; ++t) {
                const int32_t w = k[t];
#pragma unroll
                for (int ch = 0; ch < 4; ++ch)
                    if (ch < p.c) acc[ch] += (int32_t)base[a + ch] * w;
                a += p.c;
            }
        }
        uint8_t* d = (uint8_t*)p.dst + i * p.c;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch)
            if (ch < p.c) d[ch] = clip8(acc[ch]);
    }
}

// Vertical pass: every byte of an output row takes the same taps, so a thread owns 4 consecutive bytes of the flattened
// [w * c] row (dword loads / one dword store) when the row length is a multiple of 4, else single bytes.
template <int VEC>
__global__ __launch_bounds__(256) void resize_v_u8_kernel(const i2i_resize_u8_params p) {
    const int rowb = p.win * p.c, units = rowb / VEC;
    const int64_t total = (int64_t)p.n * p.nout * units;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int u = (int)(i % units);
        const int64_t r = i / units;
        const int yo = (int)(r % p.nout), img = (int)(r / p.nout);
        const int first = p.bounds[2 * yo], cnt = p.bounds[2 * yo + 1];
        const int32_t* __restrict__ k = p.coeffs + (int64_t)yo * p.ksize;
        const uint8_t* __restrict__ s = (const uint8_t*)p.src + ((int64_t)img * p.hin + first) * rowb + (int64_t)u * VEC;
        int32_t acc[VEC];
#pragma unroll
        for (int b = 0; b < VEC; ++b) acc[b] = 1 << 21;
        for (int t = 0; t < cnt; ++t) {
            const int32_t w = k[t];
            if constexpr (VEC == 4) {
                const uint32_t v = *(const uint32_t*)s;
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[b] += (int32_t)((v >> (8 * b)) & 0xff) * w;
            } else {
                acc[0] += (int32_t)s[0] * w;
            }
            s += rowb;
        }
        uint8_t* d = (uint8_t*)p.dst + ((int64_t)img * p.nout + yo) * rowb + (int64_t)u * VEC;
        if constexpr (VEC == 4) {
            *(uint32_t*)d = (uint32_t)clip8(acc[0]) | ((uint32_t)clip8(acc[1]) << 8) | ((uint32_t)clip8(acc[2]) << 16) | ((uint32_t)clip8(acc[3]) << 24);
        } else {
            d[0] = clip8(acc[0]);
        }
    }
}

}  // namespace

extern "C" int i2i_resize_u8(const i2i_resize_u8_params* p, int dtype, void* stream) {
    (void)dtype;
    if (!p || !p->src || !p->dst || !p->bounds || !p->coeffs) return i2i::fail(I2I_ERR_BAD_ARG, "resize_u8: null pointer");
    if (p->c < 1 || p->c > 4 || p->n < 1 || p->hin < 1 || p->win < 1 || p->nout < 1 || p->ksize < 1 || (p->axis != 0 && p->axis != 1))
        return i2i::fail(I2I_ERR_BAD_ARG, "resize_u8: bad geometry");
    if (((uintptr_t)p->src | (uintptr_t)p->dst) & 3) return i2i::fail(I2I_ERR_BAD_ARG, "resize_u8: image batches must be 4-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    auto blocks = [](int64_t total) { const int64_t b = (total + 255) / 256; return (unsigned)(b < 65536 ? b : 65536); };
    if (p->axis == 1) {
        hipLaunchKernelGGL(resize_h_u8_kernel, dim3(blocks((int64_t)p->n * p->hin * p->nout)), dim3(256), 0, s, *p);
    } else {
        const int rowb = p->win * p->c;
        if (rowb % 4 == 0) hipLaunchKernelGGL(resize_v_u8_kernel<4>, dim3(blocks((int64_t)p->n * p->nout * (rowb / 4))), dim3(256), 0, s, *p);
        else hipLaunchKernelGGL(resize_v_u8_kernel<1>, dim3(blocks((int64_t)p->n * p->nout * rowb)), dim3(256), 0, s, *p);
    }
    return i2i::check_launch("resize_u8");
}
