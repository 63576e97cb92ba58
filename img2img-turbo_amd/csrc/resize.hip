// Device-side LANCZOS resize of uint8 HWC image batches, bit-identical to Pillow's Image.resize(..., Image.LANCZOS)
// (the reference resizes on the host: src/inference_paired.py:38-41 to a multiple of 8, src/inference_unpaired.py:40,53 to the
// model size and back).  HBM-bound byte work: Pillow's two separable passes (horizontal, then vertical) in 32-bit integer
// arithmetic on 22-bit fixed-point weights; the per-coordinate tap windows and weights come from the host
// (img2img_turbo_amd/image_ops.py restates Pillow's precompute_coeffs / normalize_coeffs_8bpc in double precision).
//   out = clip8((2^21 + sum_t in[first + t] * k[t]) >> 22)
// One thread per output pixel (all channels); neighbouring threads read neighbouring bytes in both passes.
#include "i2i_dev.h"
#include "launch.h"

namespace {

__global__ __launch_bounds__(256) void resize_u8_kernel(const i2i_resize_u8_params p) {
    const int ho = p.axis == 0 ? p.nout : p.hin, wo = p.axis == 1 ? p.nout : p.win;
    const int64_t total = (int64_t)p.n * ho * wo;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int xo = (int)(i % wo);
        const int64_t r = i / wo;
        const int yo = (int)(r % ho), img = (int)(r / ho);
        const int o = p.axis == 1 ? xo : yo;
        const int first = p.bounds[2 * o], cnt = p.bounds[2 * o + 1];
        const int32_t* __restrict__ k = p.coeffs + (int64_t)o * p.ksize;
        // source walk: along x (stride c bytes) or along y (stride win*c bytes)
        const uint8_t* __restrict__ s = (const uint8_t*)p.src + ((int64_t)img * p.hin * p.win + (p.axis == 1 ? (int64_t)yo * p.win + first : (int64_t)first * p.win + xo)) * p.c;
        const int64_t step = p.axis == 1 ? p.c : (int64_t)p.win * p.c;
        int32_t acc[4];
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) acc[ch] = 1 << 21;
        for (int t = 0; t < cnt; ++t) {
            const int32_t w = k[t];
#pragma unroll
            for (int ch = 0; ch < 4; ++ch)
                if (ch < p.c) acc[ch] += (int32_t)s[ch] * w;
            s += step;
        }
        uint8_t* d = (uint8_t*)p.dst + i * p.c;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch)
            if (ch < p.c) {
                int v = acc[ch] >> 22;                      // arithmetic shift, as Pillow's clip8 table index
                d[ch] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
    }
}

}  // namespace

extern "C" int i2i_resize_u8(const i2i_resize_u8_params* p, int dtype, void* stream) {
    (void)dtype;
    if (!p || !p->src || !p->dst || !p->bounds || !p->coeffs) return i2i::fail(I2I_ERR_BAD_ARG, "resize_u8: null pointer");
    if (p->c < 1 || p->c > 4 || p->n < 1 || p->hin < 1 || p->win < 1 || p->nout < 1 || p->ksize < 1 || (p->axis != 0 && p->axis != 1))
        return i2i::fail(I2I_ERR_BAD_ARG, "resize_u8: bad geometry");
    const int64_t total = (int64_t)p->n * (p->axis == 0 ? (int64_t)p->nout * p->win : (int64_t)p->hin * p->nout);
    const unsigned grid = (unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipLaunchKernelGGL(resize_u8_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, *p);
    return i2i::check_launch("resize_u8");
}
