// Boundary and latent-space elementwise kernels for gfx950 (all HBM-bound, one thread per pixel):
//   NCHW <-> NHWC conversion at the .forward() boundary (torch callers hand NCHW; the kernels run NHWC),
//   DiagonalGaussianDistribution.sample()*scaling_factor + the stochastic mix (src/pix2pix_turbo.py:198,210),
//   DDPMScheduler.step + /scaling_factor + post_quant_conv (src/pix2pix_turbo.py:200-203).
#include "i2i_dev.h"
#include "launch.h"

namespace {

template <typename T, typename S>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const i2i_nchw_to_nhwc_params p) {
    const int64_t hw = (int64_t)p.h * p.w, total = (int64_t)p.n * hw;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t img = i / hw, px = i - img * hw;
        const S* x = (const S*)p.x + img * p.c * hw + px;
        T* y = (T*)p.y + i * p.cpad;
        for (int c = 0; c < p.cpad; ++c) y[c] = (c < p.c) ? from_f32<T>((float)x[(int64_t)c * hw] * p.mul + p.add) : from_f32<T>(0.f);
    }
}

template <typename T, typename D>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const i2i_nhwc_to_nchw_params p) {
    const int64_t hw = (int64_t)p.h * p.w, total = (int64_t)p.n * hw;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t img = i / hw, px = i - img * hw;
        const T* x = (const T*)p.x + i * p.ldx;
        D* y = (D*)p.y + img * p.c * hw + px;
        for (int c = 0; c < p.c; ++c) {
            float v = to_f32<T>(x[c]);
            if (p.clamp) v = fminf(fmaxf(v, -1.f), 1.f);
            y[(int64_t)c * hw] = (D)v;
        }
    }
}

// uint8 HWC image batch -> NHWC `T` (channels padded with zeros): F.to_tensor (+ Normalize) of the callers folded in
template <typename T>
__global__ __launch_bounds__(256) void u8hwc_to_nhwc_kernel(const i2i_nchw_to_nhwc_params p) {
    const int64_t total = (int64_t)p.n * p.h * p.w;
    const float k = p.mul * (1.0f / 255.0f);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const uint8_t* x = (const uint8_t*)p.x + i * p.c;
        T* y = (T*)p.y + i * p.cpad;
        for (int c = 0; c < p.cpad; ++c) {
            float v = 0.f;
            if (c < p.c) v = p.binarize_below > 0 ? ((int)x[c] < p.binarize_below ? p.mul + p.add : p.add) : (float)x[c] * k + p.add;
            y[c] = from_f32<T>(v);
        }
    }
}
// NHWC `T` -> uint8 HWC: clamp, x*mul+add, ToPILImage's mul(255).byte() (truncation)
template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_u8hwc_kernel(const i2i_nhwc_to_nchw_params p) {
    const int64_t total = (int64_t)p.n * p.h * p.w;
    const float mul = (p.mul == 0.f && p.add == 0.f) ? 1.f : p.mul;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const T* x = (const T*)p.x + i * p.ldx;
        uint8_t* y = (uint8_t*)p.y + i * p.c;
        for (int c = 0; c < p.c; ++c) {
            float v = to_f32<T>(x[c]);
            if (p.clamp) v = fminf(fmaxf(v, -1.f), 1.f);
            v = fminf(fmaxf(v * mul + p.add, 0.f), 1.f);
            y[c] = (uint8_t)(v * 255.0f);
        }
    }
}

// CLIP text embeddings: one 8-element unit per thread, y = tok[ids[row]] + pos[row % T] (sum in fp32)
template <typename T>
__global__ __launch_bounds__(256) void embed_kernel(const i2i_embed_params p) {
    typedef typename Elem<T>::chunk_t chunk_t;
    constexpr int EPC = Elem<T>::EPC;
    const int upr = p.c / EPC;
    const int64_t total = (int64_t)p.rows * upr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int row = (int)(i / upr), u = (int)(i - (int64_t)row * upr);
        int64_t id = p.ids[row];
        id = id < 0 ? 0 : (id >= p.vocab ? p.vocab - 1 : id);
        const chunk_t a = *(const chunk_t*)((const T*)p.tok + id * p.c + u * EPC);
        const chunk_t b = *(const chunk_t*)((const T*)p.pos + (int64_t)(row % p.T) * p.c + u * EPC);
        chunk_t o;
#pragma unroll
        for (int e = 0; e < EPC; ++e) o[e] = from_f32<T>(to_f32<T>(a[e]) + to_f32<T>(b[e]));
        *(chunk_t*)((T*)p.y + (int64_t)row * p.c + u * EPC) = o;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void posterior_kernel(const i2i_posterior_params p) {
    const int64_t total = (int64_t)p.n * p.hw;
    const float r = p.r_dev ? p.r_dev[0] : p.r;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t img = i / p.hw, px = i - img * p.hw;
        const T* m = (const T*)p.moments + i * p.ldm;
        const float* mf = (const float*)p.moments + i * p.ldm;
        T* u = (T*)p.u + i * p.ldu;
        for (int c = 0; c < p.ldu; ++c) {
            float v = 0.f;
            if (c < p.lat) {
                const float mean = p.moments_f32 ? mf[c] : to_f32<T>(m[c]);
                const float logvar = fminf(fmaxf(p.moments_f32 ? mf[p.lat + c] : to_f32<T>(m[p.lat + c]), -30.f), 20.f);
                const float e = p.eps[(img * p.lat + c) * p.hw + px];
                v = (mean + __expf(0.5f * logvar) * e) * p.sf;
                if (p.noise) {
                    const int64_t nimg = (p.noise_n == 1) ? 0 : img;
                    v = v * r + p.noise[(nimg * p.lat + c) * p.hw + px] * (1.f - r);
                }
            }
            u[c] = from_f32<T>(v);
            if (p.u_f32 && c < p.lat) p.u_f32[i * p.lat + c] = v;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void ddpm_kernel(const i2i_ddpm_params p) {
    const int64_t total = (int64_t)p.n * p.hw;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const T* u = (const T*)p.u + i * p.ldu;
        const T* e = (const T*)p.e + i * p.lde;
        T* y = (T*)p.y + i * p.ldy;
        float x0[8];
        const float* uf = (const float*)p.u + i * p.ldu;
        const float* ef = (const float*)p.e + i * p.lde;
        for (int c = 0; c < p.lat; ++c) {
            const float uv = p.u_f32 ? uf[c] : to_f32<T>(u[c]);
            const float ev = p.e_f32 ? ef[c] : to_f32<T>(e[c]);
            x0[c] = ((uv - p.sqrt_1m_abar * ev) / p.sqrt_abar) / p.sf;
        }
        for (int o = 0; o < p.ldy; ++o) {
            float v = 0.f;
            if (o < p.lat) {
                v = p.bpq[o];
                for (int c = 0; c < p.lat; ++c) v += p.wpq[o * p.lat + c] * x0[c];
            }
            y[o] = from_f32<T>(v);
        }
    }
}

inline unsigned grid_for(int64_t n) {
    const int64_t b = (n + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

extern "C" int i2i_nchw_to_nhwc(const i2i_nchw_to_nhwc_params* p, int dtype, void* stream) {
    if (!p || !p->x || !p->y || p->cpad < p->c) return i2i::fail(I2I_ERR_BAD_ARG, "nchw_to_nhwc: bad args");
    hipStream_t s = (hipStream_t)stream;
    const unsigned g = grid_for((int64_t)p->n * p->h * p->w);
    if (p->src_dtype == I2I_U8) {
        switch (dtype) {
            case I2I_F32: hipLaunchKernelGGL((u8hwc_to_nhwc_kernel<float>), dim3(g), dim3(256), 0, s, *p); break;
            case I2I_BF16: hipLaunchKernelGGL((u8hwc_to_nhwc_kernel<__bf16>), dim3(g), dim3(256), 0, s, *p); break;
            case I2I_F16: hipLaunchKernelGGL((u8hwc_to_nhwc_kernel<_Float16>), dim3(g), dim3(256), 0, s, *p); break;
            default: return i2i::fail(I2I_ERR_BAD_ARG, "u8hwc_to_nhwc: bad dtype");
        }
        return i2i::check_launch("u8hwc_to_nhwc");
    }
    const bool src_f32 = p->src_dtype == I2I_F32;
    if (!src_f32 && p->src_dtype != dtype) return i2i::fail(I2I_ERR_BAD_ARG, "nchw_to_nhwc: src dtype must be f32 or the compute dtype");
    switch (dtype) {
        case I2I_F32: hipLaunchKernelGGL((nchw_to_nhwc_kernel<float, float>), dim3(g), dim3(256), 0, s, *p); break;
        case I2I_BF16:
            if (src_f32) hipLaunchKernelGGL((nchw_to_nhwc_kernel<__bf16, float>), dim3(g), dim3(256), 0, s, *p);
            else hipLaunchKernelGGL((nchw_to_nhwc_kernel<__bf16, __bf16>), dim3(g), dim3(256), 0, s, *p);
            break;
        case I2I_F16:
            if (src_f32) hipLaunchKernelGGL((nchw_to_nhwc_kernel<_Float16, float>), dim3(g), dim3(256), 0, s, *p);
            else hipLaunchKernelGGL((nchw_to_nhwc_kernel<_Float16, _Float16>), dim3(g), dim3(256), 0, s, *p);
            break;
        default: return i2i::fail(I2I_ERR_BAD_ARG, "nchw_to_nhwc: bad dtype");
    }
    return i2i::check_launch("nchw_to_nhwc");
}

extern "C" int i2i_nhwc_to_nchw(const i2i_nhwc_to_nchw_params* p, int dtype, void* stream) {
    if (!p || !p->x || !p->y || p->ldx < p->c) return i2i::fail(I2I_ERR_BAD_ARG, "nhwc_to_nchw: bad args");
    hipStream_t s = (hipStream_t)stream;
    const unsigned g = grid_for((int64_t)p->n * p->h * p->w);
    if (p->dst_dtype == I2I_U8) {
        switch (dtype) {
            case I2I_F32: hipLaunchKernelGGL((nhwc_to_u8hwc_kernel<float>), dim3(g), dim3(256), 0, s, *p); break;
            case I2I_BF16: hipLaunchKernelGGL((nhwc_to_u8hwc_kernel<__bf16>), dim3(g), dim3(256), 0, s, *p); break;
            case I2I_F16: hipLaunchKernelGGL((nhwc_to_u8hwc_kernel<_Float16>), dim3(g), dim3(256), 0, s, *p); break;
            default: return i2i::fail(I2I_ERR_BAD_ARG, "nhwc_to_u8hwc: bad dtype");
        }
        return i2i::check_launch("nhwc_to_u8hwc");
    }
    const bool dst_f32 = p->dst_dtype == I2I_F32;
    if (!dst_f32 && p->dst_dtype != dtype) return i2i::fail(I2I_ERR_BAD_ARG, "nhwc_to_nchw: dst dtype must be f32 or the compute dtype");
    switch (dtype) {
        case I2I_F32: hipLaunchKernelGGL((nhwc_to_nchw_kernel<float, float>), dim3(g), dim3(256), 0, s, *p); break;
        case I2I_BF16:
            if (dst_f32) hipLaunchKernelGGL((nhwc_to_nchw_kernel<__bf16, float>), dim3(g), dim3(256), 0, s, *p);
            else hipLaunchKernelGGL((nhwc_to_nchw_kernel<__bf16, __bf16>), dim3(g), dim3(256), 0, s, *p);
            break;
        case I2I_F16:
            if (dst_f32) hipLaunchKernelGGL((nhwc_to_nchw_kernel<_Float16, float>), dim3(g), dim3(256), 0, s, *p);
            else hipLaunchKernelGGL((nhwc_to_nchw_kernel<_Float16, _Float16>), dim3(g), dim3(256), 0, s, *p);
            break;
        default: return i2i::fail(I2I_ERR_BAD_ARG, "nhwc_to_nchw: bad dtype");
    }
    return i2i::check_launch("nhwc_to_nchw");
}

extern "C" int i2i_posterior(const i2i_posterior_params* p, int dtype, void* stream) {
    if (!p || !p->moments || !p->eps || !p->u) return i2i::fail(I2I_ERR_BAD_ARG, "posterior: null pointer");
    if (p->ldm < 2 * p->lat || p->ldu < p->lat) return i2i::fail(I2I_ERR_BAD_ARG, "posterior: bad leading dims");
    hipStream_t s = (hipStream_t)stream;
    const unsigned g = grid_for((int64_t)p->n * p->hw);
    switch (dtype) {
        case I2I_F32: hipLaunchKernelGGL((posterior_kernel<float>), dim3(g), dim3(256), 0, s, *p); break;
        case I2I_BF16: hipLaunchKernelGGL((posterior_kernel<__bf16>), dim3(g), dim3(256), 0, s, *p); break;
        case I2I_F16: hipLaunchKernelGGL((posterior_kernel<_Float16>), dim3(g), dim3(256), 0, s, *p); break;
        default: return i2i::fail(I2I_ERR_BAD_ARG, "posterior: bad dtype");
    }
    return i2i::check_launch("posterior");
}

extern "C" int i2i_ddpm_postquant(const i2i_ddpm_params* p, int dtype, void* stream) {
    if (!p || !p->u || !p->e || !p->y || !p->wpq || !p->bpq) return i2i::fail(I2I_ERR_BAD_ARG, "ddpm: null pointer");
    if (p->lat > 8 || p->ldy < p->lat) return i2i::fail(I2I_ERR_BAD_ARG, "ddpm: latent channels must be <= 8");
    hipStream_t s = (hipStream_t)stream;
    const unsigned g = grid_for((int64_t)p->n * p->hw);
    switch (dtype) {
        case I2I_F32: hipLaunchKernelGGL((ddpm_kernel<float>), dim3(g), dim3(256), 0, s, *p); break;
        case I2I_BF16: hipLaunchKernelGGL((ddpm_kernel<__bf16>), dim3(g), dim3(256), 0, s, *p); break;
        case I2I_F16: hipLaunchKernelGGL((ddpm_kernel<_Float16>), dim3(g), dim3(256), 0, s, *p); break;
        default: return i2i::fail(I2I_ERR_BAD_ARG, "ddpm: bad dtype");
    }
    return i2i::check_launch("ddpm");
}

extern "C" int i2i_embed(const i2i_embed_params* p, int dtype, void* stream) {
    if (!p || !p->ids || !p->tok || !p->pos || !p->y || p->c % 8 || p->T < 1 || p->vocab < 1) return i2i::fail(I2I_ERR_BAD_ARG, "embed: bad args");
    hipStream_t s = (hipStream_t)stream;
    const unsigned g = grid_for((int64_t)p->rows * (p->c / 4));
    switch (dtype) {
        case I2I_F32: hipLaunchKernelGGL((embed_kernel<float>), dim3(g), dim3(256), 0, s, *p); break;
        case I2I_BF16: hipLaunchKernelGGL((embed_kernel<__bf16>), dim3(g), dim3(256), 0, s, *p); break;
        case I2I_F16: hipLaunchKernelGGL((embed_kernel<_Float16>), dim3(g), dim3(256), 0, s, *p); break;
        default: return i2i::fail(I2I_ERR_BAD_ARG, "embed: bad dtype");
    }
    return i2i::check_launch("embed");
}
