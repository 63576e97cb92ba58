// Normalisation and softmax kernels for gfx950: GroupNorm statistics (two-stage, atomics-free),
// standalone GroupNorm apply, LayerNorm, row softmax.
//
// GroupNorm is split the MI355X way: the reduction (F.group_norm's mean/var) is its own streaming
// pass that emits per-(image, channel) (scale, shift); the affine + SiLU is applied by the consumer
// igemm while it stages its A operand, so the normalised tensor is never written to HBM
// (diffusers runs norm, silu and conv as three library kernels: resnet.py ResnetBlock2D.forward).
#include "i2i_dev.h"
#include "launch.h"

namespace {

constexpr int GN_JMAX = 5;   // up to 5*64*8 = 2560 channels (UNet 2560-channel concat inputs)

// lanes of a wave cover LPP 8-channel units of one pixel; 64/LPP pixels per wave step.
template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(const i2i_gn_stats_params p) {
    typedef typename Elem<T>::chunk_t chunk_t;
    constexpr int EPC = Elem<T>::EPC;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ct = p.c0 + p.c1, cc = ct >> 3;
    int lpp = 64;
    while (lpp > cc) lpp >>= 1;
    const int ppw = 64 / lpp, jn = (cc + lpp - 1) / lpp;
    const int lane_cp = lane & (lpp - 1), lane_px = lane / lpp;
    const int img = blockIdx.y, part = blockIdx.x;
    const int per = (p.hw + p.nparts - 1) / p.nparts;
    const int px0 = part * per, px1 = min(p.hw, px0 + per);

    float s[GN_JMAX][8], q[GN_JMAX][8];
#pragma unroll
    for (int j = 0; j < GN_JMAX; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[j][e] = 0.f; q[j][e] = 0.f; }

    const T* x0 = (const T*)p.x0 + (int64_t)img * p.hw * p.ld0;
    const T* x1 = p.x1 ? (const T*)p.x1 + (int64_t)img * p.hw * p.ld1 : nullptr;
    for (int px = px0 + wave * ppw + lane_px; px < px1; px += 4 * ppw) {
#pragma unroll
        for (int j = 0; j < GN_JMAX; ++j) {
            const int cp = lane_cp + j * lpp;
            if (j < jn && cp < cc) {
                const int c = cp << 3;
                const T* src = (c < p.c0) ? x0 + (int64_t)px * p.ld0 + c : x1 + (int64_t)px * p.ld1 + (c - p.c0);
#pragma unroll
                for (int h = 0; h < 8 / EPC; ++h) {
                    const chunk_t v = *(const chunk_t*)(src + h * EPC);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        const float f = to_f32<T>(v[e]);
                        s[j][h * EPC + e] += f;
                        q[j][h * EPC + e] += f * f;
                    }
                }
            }
        }
    }
    // reduce over the pixel sub-lanes of the wave (same channel unit lives at lane_cp + m*lpp)
#pragma unroll
    for (int j = 0; j < GN_JMAX; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e)
            for (int m = lpp; m < 64; m <<= 1) {
                s[j][e] += __shfl_xor(s[j][e], m);
                q[j][e] += __shfl_xor(q[j][e], m);
            }
    // cross-wave: fixed order wave 0,1,2,3 (deterministic)
    float* chs = (float*)i2i_smem;
    float* chq = chs + ct;
    for (int w = 0; w < 4; ++w) {
        if (wave == w && lane_px == 0) {
#pragma unroll
            for (int j = 0; j < GN_JMAX; ++j) {
                const int cp = lane_cp + j * lpp;
                if (j < jn && cp < cc) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int c = (cp << 3) + e;
                        if (w == 0) { chs[c] = s[j][e]; chq[c] = q[j][e]; }
                        else { chs[c] += s[j][e]; chq[c] += q[j][e]; }
                    }
                }
            }
        }
        __syncthreads();
    }
    const int cpg = ct / p.groups;
    for (int g = tid; g < p.groups; g += 256) {
        float S = 0.f, Q = 0.f;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) { S += chs[c]; Q += chq[c]; }
        float* out = p.partial + (((int64_t)img * p.nparts + part) * p.groups + g) * 2;
        out[0] = S;
        out[1] = Q;
    }
}

// ---- second pass for groups whose one-pass variance cannot be trusted.  E[x^2] - mu^2 in fp32 loses log2(mu^2/var) bits to
// cancellation on top of the rounding the sums already carry (~1e-6 relative: per-tile partial sums combined in a fixed tree):
// the relative error of the variance is ~1e-6 * mu^2/var.  F.group_norm is two-pass / Welford, and real SD-VAE decoder
// activations do sit on large DC offsets.  The flagged groups -- and only those: the decision is a deterministic function of
// the first pass -- are re-read by the whole block against the first-pass mean: mu' = mu + E[x - mu],
// var' = E[(x - mu)^2] - E[x - mu]^2, exact to fp32 round-off.  Needs the tensor (p.x0 / p.x1), also in finalize_only mode;
// without it (x0 == NULL) the one-pass numbers stand.
//
// WHEN a group is flagged depends on the storage type, because the consumer only keeps so much: the normalised value is applied
// to / stored as a `dtype` number, i.e. rounded to 2^-p of its magnitude (p = 8 mantissa bits for bf16, 11 for fp16), while a
// variance error of relative size delta moves a normalised value x^ by |x^| delta / 2.  MEASURED on this kernel family: delta ~ 2.8e-7 *
// mu^2 / var (7.8e-3 absolute on outputs up to 5.7 at mu^2 / var = 1e4, 512 conv-epilogue style parts: tests/opcheck.py
// check_gn_stats_offset), i.e. a relative output error of ~1.4e-7 * mu^2 / var.  The second pass is worth its re-read of the tensor only
// while that error can reach HALF a rounding step of the consumer's type: 2^-12 / 1.4e-7 = 1 744 for fp16, 2^-9 / 1.4e-7 = 13 950 for bf16
// -- the constants below are the next powers of two (2 048, 16 384); fp32 storage keeps the round-2 setting (256: a relative error of
// 3.6e-5, well inside north_star's 1e-3).  Below these ratios the re-read would polish digits the 16-bit output does not have; both sides
// of each boundary are pinned by tests (test_gn_stats_flag_boundaries: one rounding step of the type just under the ratio, the
// second pass just above it).  The price of a flagged tensor: bench.py --activation-offset.
template <typename T> struct GnRefine;
template <> struct GnRefine<float> { static constexpr float RATIO = 256.f; };
template <> struct GnRefine<_Float16> { static constexpr float RATIO = 2048.f; };
template <> struct GnRefine<__bf16> { static constexpr float RATIO = 16384.f; };
// One group = cpg CONSECUTIVE channels of every pixel (2 * cpg bytes for the 16-bit types).  A thread takes whole pixels and
// reads the group's channels in the widest pieces the group's first channel allows (16 / 8 / 4 bytes: cpg = 16 / 8 / 4 for the
// VAE's 512 / 256 / 128-channel planes, 4-byte pieces for the UNet's 10 / 20 / 40-channel groups), four pixels in flight per trip
// (a lone dependent stream of loads is latency bound).  Fixed order -> deterministic.
template <typename T, int VB /* bytes per piece */>
__device__ __forceinline__ void gn_refine_span(const char* base, int64_t ldb, int hw, int nbytes, float mu, float& s1, float& s2) {
    constexpr int EPV = VB / (int)sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(EPV)));
    const int npc = nbytes / VB;
    constexpr int UNR = 4;
    for (int px0 = threadIdx.x; px0 < hw; px0 += 256 * UNR) {
        for (int k = 0; k < npc; ++k) {
            vec_t v[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int px = px0 + u * 256;
                v[u] = *(const vec_t*)(base + (int64_t)(px < hw ? px : px0) * ldb + k * VB);
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                if (px0 + u * 256 < hw) {
#pragma unroll
                    for (int e = 0; e < EPV; ++e) {
                        const float d = to_f32<T>(v[u][e]) - mu;
                        s1 += d;
                        s2 += d * d;
                    }
                }
            }
        }
    }
}
template <typename T>
__device__ void gn_refine_group(const i2i_gn_stats_params& p, int img, int g, float* red /* >= 512 floats of LDS */, float& mu, float& var) {
    const int tid = threadIdx.x, ct = p.c0 + p.c1, cpg = ct / p.groups;
    float s1 = 0.f, s2 = 0.f;
    // the group's channels inside each source: [ca, cb) of x0 and [cc, cd) of x1 (a group may straddle the concat boundary)
    const int c_lo = g * cpg, c_hi = c_lo + cpg;
    for (int src = 0; src < 2; ++src) {
        const int lo = src == 0 ? c_lo : (c_lo > p.c0 ? c_lo : p.c0), hi = src == 0 ? (c_hi < p.c0 ? c_hi : p.c0) : c_hi;
        if (hi <= lo || (src == 1 && !p.x1)) continue;
        const int ld = src == 0 ? p.ld0 : p.ld1, c_in = src == 0 ? lo : lo - p.c0;
        const char* base = (const char*)((const T*)(src == 0 ? p.x0 : p.x1) + (int64_t)img * p.hw * ld + c_in);
        const int nbytes = (hi - lo) * (int)sizeof(T);
        const int64_t ldb = (int64_t)ld * (int)sizeof(T);
        const int al = (int)(((uintptr_t)base | (uintptr_t)ldb | (uintptr_t)nbytes) & 15);
        if (al == 0) gn_refine_span<T, 16>(base, ldb, p.hw, nbytes, mu, s1, s2);
        else if ((al & 7) == 0) gn_refine_span<T, 8>(base, ldb, p.hw, nbytes, mu, s1, s2);
        else if ((al & 3) == 0) gn_refine_span<T, 4>(base, ldb, p.hw, nbytes, mu, s1, s2);
        else gn_refine_span<T, (int)sizeof(T)>(base, ldb, p.hw, nbytes, mu, s1, s2);
    }
    __syncthreads();
    red[2 * tid] = s1;
    red[2 * tid + 1] = s2;
    __syncthreads();
    float S1 = 0.f, S2 = 0.f;
    for (int k = 0; k < 256; ++k) { S1 += red[2 * k]; S2 += red[2 * k + 1]; }      // every thread, fixed order: deterministic, uniform result
    const float inv = 1.0f / ((float)cpg * (float)p.hw), m1 = S1 * inv;
    mu += m1;
    var = fmaxf(S2 * inv - m1 * m1, 0.f);
    __syncthreads();
}

// grid = (nimg, groups / gpb): a block finishes gpb groups of one image.  Each group's part range is cut into
// 256 / gpb slices (conv epilogues hand over thousands of parts) that are combined in a fixed order afterwards,
// so the result does not depend on scheduling.
template <typename T>
__global__ __launch_bounds__(256) void gn_finalize_kernel(const i2i_gn_stats_params p, int gpb) {
    const int tid = threadIdx.x, img = blockIdx.x, g0 = blockIdx.y * gpb;
    const int ct = p.c0 + p.c1, cpg = ct / p.groups;
    float* mean = (float*)i2i_smem;               // [gpb]
    float* rstd = mean + gpb;                     // [gpb]
    float* red = rstd + gpb;                      // [nsl][gpb][2] slice partials
    const int nsl = 256 / gpb;
    const int gl = tid % gpb, sl = tid / gpb;
    // gamma / beta of this thread's first channel, requested with the partial sums (at their use below they were one more dependent round
    // trip behind three barriers, in a kernel of ~5 us)
    const int cpre = g0 * cpg + tid;
    const bool pre_ok = cpre < (g0 + gpb) * cpg;
    const float gam0 = p.gamma[pre_ok ? cpre : 0], bet0 = p.beta[pre_ok ? cpre : 0];
    if (sl < nsl) {
        float S = 0.f, Q = 0.f;
        typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll 8
        for (int part = sl; part < p.nparts; part += nsl) {
            const f32x2 v = *(const f32x2*)(p.partial + (((int64_t)img * p.nparts + part) * p.groups + g0 + gl) * 2);
            S += v[0];
            Q += v[1];
        }
        red[(sl * gpb + gl) * 2 + 0] = S;
        red[(sl * gpb + gl) * 2 + 1] = Q;
    }
    __syncthreads();
    if (tid < gpb) {
        float S = 0.f, Q = 0.f;
        for (int k = 0; k < nsl; ++k) { S += red[(k * gpb + tid) * 2]; Q += red[(k * gpb + tid) * 2 + 1]; }
        const float inv = 1.0f / ((float)cpg * (float)p.hw);
        const float mu = S * inv;
        const float var = fmaxf(Q * inv - mu * mu, 0.f);
        mean[tid] = mu;
        rstd[tid] = var;          // variance for now: the refinement below may replace it
    }
    __syncthreads();
    for (int g = 0; g < gpb; ++g) {                // uniform over the block
        float mu = mean[g], var = rstd[g];
        if (p.x0 && var * GnRefine<T>::RATIO < mu * mu) gn_refine_group<T>(p, img, g0 + g, red, mu, var);
        __syncthreads();
        if (tid == 0) { mean[g] = mu; rstd[g] = rsqrtf(var + p.eps); }
    }
    __syncthreads();
    for (int c = cpre; c < (g0 + gpb) * cpg; c += 256) {
        const int g = c / cpg - g0;
        const float sc = rstd[g] * (c == cpre ? gam0 : p.gamma[c]);
        float* o = p.ss + ((int64_t)img * ct + c) * 2;
        o[0] = sc;
        o[1] = (c == cpre ? bet0 : p.beta[c]) - mean[g] * sc;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const i2i_gn_apply_params p) {
    typedef typename Elem<T>::chunk_t chunk_t;
    constexpr int EPC = Elem<T>::EPC;
    const int c1 = p.x1 ? p.c1 : 0, ct = p.c + c1;      // (a second source: its channels follow the first's in y and ss)
    const int64_t nchunk = (int64_t)p.nimg * p.hw * ct / EPC;
    const int cpp = ct / EPC;   // chunks per pixel
    const int ldx = p.ldx ? p.ldx : p.c, ldx1 = p.ldx1 ? p.ldx1 : c1, ldy = p.ldy ? p.ldy : ct;
    const int ss_ld = p.ss_ld ? p.ss_ld : ct;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nchunk; i += (int64_t)gridDim.x * 256) {
        const int64_t pix = i / cpp;
        const int c = (int)(i - pix * cpp) * EPC;
        const int img = (int)(pix / p.hw);
        chunk_t v = c < p.c ? *(const chunk_t*)((const T*)p.x + pix * ldx + c) : *(const chunk_t*)((const T*)p.x1 + pix * ldx1 + (c - p.c));
        const float* ss = p.ss + ((int64_t)img * ss_ld + p.ss_off + c) * 2;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            float f = to_f32<T>(v[e]) * ss[2 * e] + ss[2 * e + 1];
            if (p.act == 1) f = silu_f(f);
            v[e] = from_f32<T>(f);
        }
        *(chunk_t*)((T*)p.y + pix * ldy + c) = v;
    }
}

constexpr int LN_JMAX = 4;   // 4*64*8 = 2048 channels per row max
// One wave per row, a lane owns chunks lane, lane + 64, ... (8 channels each; J = chunks per lane: 1 up to 512 channels, 2 up to 1024, 4 up
// to 2048).  ALL of a lane's row loads and its gamma / beta quads are requested before the first reduction: behind `if (chunk < cc)` each
// load was waited for at its join and gamma / beta were only requested after the second reduction -- 3 to 6 dependent round trips in a
// kernel of ~10 us.  (Lanes past the row read chunk 0 and drop it.)  Arithmetic and its order are unchanged.
// (Several rows per wave -- more loads per lane in flight, fewer waves -- measured SLOWER: profiles/r5p_ab_layernorm_rows_per_wave_negative.log.)
template <typename T, int J>
__global__ __launch_bounds__(256) void layernorm_kernel(const i2i_layernorm_params p) {
    typedef typename Elem<T>::chunk_t chunk_t;
    constexpr int EPC = Elem<T>::EPC, H = 8 / EPC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= p.rows) return;
    const int cc = p.c >> 3;
    const T* x = (const T*)p.x + (int64_t)row * p.ldx;
    chunk_t raw[J][H];
    f32x4 gq[J][2], bq[J][2];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int cp = lane + j * 64, cpc = cp < cc ? cp : 0;
#pragma unroll
        for (int h = 0; h < H; ++h) raw[j][h] = *(const chunk_t*)(x + (cpc << 3) + h * EPC);
        gq[j][0] = *(const f32x4*)(p.gamma + (cpc << 3)); gq[j][1] = *(const f32x4*)(p.gamma + (cpc << 3) + 4);
        bq[j][0] = *(const f32x4*)(p.beta + (cpc << 3));  bq[j][1] = *(const f32x4*)(p.beta + (cpc << 3) + 4);
    }
    float v[J][8];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const bool ok = lane + j * 64 < cc;
#pragma unroll
        for (int h = 0; h < H; ++h)
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                v[j][h * EPC + e] = ok ? to_f32<T>(raw[j][h][e]) : 0.f;
                if (ok) sum += v[j][h * EPC + e];
            }
    }
    const float mu = wave_sum(sum) / (float)p.c;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < J; ++j)
        if (lane + j * 64 < cc) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[j][e] - mu; sq += d * d; }
        }
    const float rstd = rsqrtf(wave_sum(sq) / (float)p.c + p.eps);
    T* y = (T*)p.y + (int64_t)row * p.ldy;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int cp = lane + j * 64;
        if (cp < cc) {
#pragma unroll
            for (int h = 0; h < H; ++h) {
                chunk_t c;
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const int q = h * EPC + e;
                    c[e] = from_f32<T>((v[j][q] - mu) * rstd * gq[j][q >> 2][q & 3] + bq[j][q >> 2][q & 3]);
                }
                *(chunk_t*)(y + (cp << 3) + h * EPC) = c;
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_kernel(const i2i_softmax_params p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= p.rows) return;
    const float* s = p.s + row * p.lds;
    float mx = -INFINITY;
    for (int c = lane; c < p.cols; c += 64) mx = fmaxf(mx, s[c] * p.scale);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int c = lane; c < p.cols; c += 64) sum += __expf(s[c] * p.scale - mx);
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    T* o = (T*)p.p + row * p.ldp;
    for (int c = lane; c < p.ldp; c += 64) o[c] = (c < p.cols) ? from_f32<T>(__expf(s[c] * p.scale - mx) * inv) : from_f32<T>(0.f);
}

// Single-launch GroupNorm statistics for the small tensors (UNet levels, VAE mid block): one workgroup owns `gpb`
// groups of one image, streams their channels over every pixel, reduces through LDS in a fixed order and writes the
// per-channel (scale, shift) itself -- the two-stage partial + finalize pair costs two launches of a few
// microseconds each, which at batch 1 is a fifth of the whole forward.
// Threads: U = gpb*cpg/8 eight-channel units per pixel, 256/U pixels per sweep.
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_small_kernel(const i2i_gn_stats_params p, int gpb) {
    typedef typename Elem<T>::chunk_t chunk_t;
    constexpr int EPC = Elem<T>::EPC;
    const int tid = threadIdx.x, img = blockIdx.x, g0 = blockIdx.y * gpb;
    const int ct = p.c0 + p.c1, cpg = ct / p.groups;
    const int nch = gpb * cpg, c_first = g0 * cpg;          // this block's channel range (multiple of 8: host check)
    const float gam0 = p.gamma[c_first + (tid < nch ? tid : 0)], bet0 = p.beta[c_first + (tid < nch ? tid : 0)];      // (requested up front: see gn_finalize_kernel)
    const int U = nch >> 3, ppb = 256 / U;
    const int unit = tid % U, prow = tid / U;
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
    if (prow < ppb) {
        const int c = c_first + unit * 8;
        const T* src = (c < p.c0) ? (const T*)p.x0 + (int64_t)img * p.hw * p.ld0 + c
                                  : (const T*)p.x1 + (int64_t)img * p.hw * p.ld1 + (c - p.c0);
        const int ld = (c < p.c0) ? p.ld0 : p.ld1;
        // four pixels per trip: four independent 16-byte loads in flight per thread (a lone dependent stream of
        // loads is latency bound at ~20 GB/s per workgroup)
        constexpr int UNR = 4;
        for (int px0 = prow; px0 < p.hw; px0 += ppb * UNR) {
            chunk_t v[UNR][8 / EPC];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int px = px0 + u * ppb;
#pragma unroll
                for (int h = 0; h < 8 / EPC; ++h)
                    v[u][h] = (px < p.hw) ? *(const chunk_t*)(src + (int64_t)px * ld + h * EPC) : zero_chunk<T>();
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u)
#pragma unroll
                for (int h = 0; h < 8 / EPC; ++h)
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        const float f = to_f32<T>(v[u][h][e]);
                        s[h * EPC + e] += f;
                        q[h * EPC + e] += f * f;
                    }
        }
    }
    float* red = (float*)i2i_smem;                // [256][16]
    float* chs = red + 256 * 16;                  // [nch][2]
    float* gst = chs + 2 * nch;                   // [gpb][2] mean, rstd
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[tid * 16 + e] = s[e]; red[tid * 16 + 8 + e] = q[e]; }
    __syncthreads();
    for (int c = tid; c < nch; c += 256) {        // per-channel totals over the pixel rows, fixed order
        const int u = c >> 3, e = c & 7;
        float S = 0.f, Q = 0.f;
        for (int r = 0; r < ppb; ++r) { S += red[(r * U + u) * 16 + e]; Q += red[(r * U + u) * 16 + 8 + e]; }
        chs[2 * c] = S; chs[2 * c + 1] = Q;
    }
    __syncthreads();
    if (tid < gpb) {
        float S = 0.f, Q = 0.f;
        for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { S += chs[2 * c]; Q += chs[2 * c + 1]; }
        const float inv = 1.0f / ((float)cpg * (float)p.hw);
        const float mu = S * inv;
        const float var = fmaxf(Q * inv - mu * mu, 0.f);
        gst[2 * tid] = mu;
        gst[2 * tid + 1] = var;   // variance for now: the refinement below may replace it
    }
    __syncthreads();
    for (int g = 0; g < gpb; ++g) {                // uniform over the block
        float mu = gst[2 * g], var = gst[2 * g + 1];
        if (var * GnRefine<T>::RATIO < mu * mu) gn_refine_group<T>(p, img, g0 + g, red, mu, var);
        __syncthreads();
        if (tid == 0) { gst[2 * g] = mu; gst[2 * g + 1] = rsqrtf(var + p.eps); }
    }
    __syncthreads();
    for (int c = tid; c < nch; c += 256) {
        const int g = c / cpg, cc = c_first + c;
        const float sc = gst[2 * g + 1] * (c == tid ? gam0 : p.gamma[cc]);
        float* o = p.ss + ((int64_t)img * ct + cc) * 2;
        o[0] = sc;
        o[1] = (c == tid ? bet0 : p.beta[cc]) - gst[2 * g] * sc;
    }
}

// ---- the same, with every image's pixels cut into S slices (grid z): at batch 1 the kernel above runs on 8 workgroups (32 groups
// in sets of 4) and each of them streams 300+ KB through 256 threads -- 13.8 us per launch, 63 launches per forward.  Slice
// workgroup (img, set, s) reduces its pixels to (sum, sum of squares) per group and publishes the gpb pairs with 8-byte
// agent-scope stores (write-through: visible to every XCD once the store is acknowledged -- no release fence, whose L2 write-back
// costs more than this whole kernel), waits for them (vmcnt 0), and draws a ticket from the (img, set) counter; the workgroup that
// draws S-1 reads all S slices back with agent-scope loads, adds them IN SLICE ORDER (deterministic whichever workgroup is
// last), finalises as above and re-zeroes the counter for the next launch.  (cdna_hip_programming.md section 6, Guideline 16:
// "8-B agent atomics both sides" hand-off; the ticket is a relaxed agent-scope fetch_add.)
#ifdef I2I_EMU
__device__ __forceinline__ void gn_publish(float* dst, float a, float b) { dst[0] = a; dst[1] = b; }
__device__ __forceinline__ void gn_fetch(const float* src, float& a, float& b) { a = src[0]; b = src[1]; }
__device__ __forceinline__ int gn_ticket(int32_t* c) { const int v = *c; *c = v + 1; return v; }
__device__ __forceinline__ void gn_ticket_reset(int32_t* c) { *c = 0; }
__device__ __forceinline__ void gn_stores_done() {}
#else
__device__ __forceinline__ void gn_publish(float* dst, float a, float b) {
    const unsigned long long v = ((unsigned long long)__builtin_bit_cast(unsigned, b) << 32) | (unsigned long long)__builtin_bit_cast(unsigned, a);
    __hip_atomic_store((unsigned long long*)dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void gn_fetch(const float* src, float& a, float& b) {
    const unsigned long long v = __hip_atomic_load((const unsigned long long*)src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a = __builtin_bit_cast(float, (unsigned)v);
    b = __builtin_bit_cast(float, (unsigned)(v >> 32));
}
__device__ __forceinline__ int gn_ticket(int32_t* c) { return __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void gn_ticket_reset(int32_t* c) { __hip_atomic_store(c, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void gn_stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_sliced_kernel(const i2i_gn_stats_params p, int gpb, int S) {
    typedef typename Elem<T>::chunk_t chunk_t;
    constexpr int EPC = Elem<T>::EPC;
    const int tid = threadIdx.x, img = blockIdx.x, set = blockIdx.y, sl = blockIdx.z, g0 = set * gpb;
    const int ct = p.c0 + p.c1, cpg = ct / p.groups, nsets = p.groups / gpb;
    const int nch = gpb * cpg, c_first = g0 * cpg;
    const float gam0 = p.gamma[c_first + (tid < nch ? tid : 0)], bet0 = p.beta[c_first + (tid < nch ? tid : 0)];      // (requested up front: see gn_finalize_kernel)
    const int U = nch >> 3, ppb = 256 / U;
    const int unit = tid % U, prow = tid / U;
    const int per = (p.hw + S - 1) / S, px_lo = sl * per, px_hi = px_lo + per < p.hw ? px_lo + per : p.hw;
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
    if (prow < ppb) {
        const int c = c_first + unit * 8;
        const T* src = (c < p.c0) ? (const T*)p.x0 + (int64_t)img * p.hw * p.ld0 + c
                                  : (const T*)p.x1 + (int64_t)img * p.hw * p.ld1 + (c - p.c0);
        const int ld = (c < p.c0) ? p.ld0 : p.ld1;
        constexpr int UNR = 4;
        for (int px0 = px_lo + prow; px0 < px_hi; px0 += ppb * UNR) {
            chunk_t v[UNR][8 / EPC];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int px = px0 + u * ppb;
#pragma unroll
                for (int h = 0; h < 8 / EPC; ++h)
                    v[u][h] = (px < px_hi) ? *(const chunk_t*)(src + (int64_t)px * ld + h * EPC) : zero_chunk<T>();
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u)
#pragma unroll
                for (int h = 0; h < 8 / EPC; ++h)
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        const float f = to_f32<T>(v[u][h][e]);
                        s[h * EPC + e] += f;
                        q[h * EPC + e] += f * f;
                    }
        }
    }
    float* red = (float*)i2i_smem;                // [256][16]
    float* chs = red + 256 * 16;                  // [nch][2]
    float* gst = chs + 2 * nch;                   // [gpb][2] mean, rstd
    int* flag = (int*)(gst + 2 * gpb);            // the ticket this workgroup drew
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[tid * 16 + e] = s[e]; red[tid * 16 + 8 + e] = q[e]; }
    __syncthreads();
    for (int c = tid; c < nch; c += 256) {        // per-channel totals over the pixel rows, fixed order
        const int u = c >> 3, e = c & 7;
        float S1 = 0.f, Q1 = 0.f;
        for (int r = 0; r < ppb; ++r) { S1 += red[(r * U + u) * 16 + e]; Q1 += red[(r * U + u) * 16 + 8 + e]; }
        chs[2 * c] = S1; chs[2 * c + 1] = Q1;
    }
    __syncthreads();
    float* slot = p.partial + ((((int64_t)img * nsets + set) * S) * gpb) * 2;      // [S][gpb][2] of this (image, group set)
    if (tid < gpb) {
        float S1 = 0.f, Q1 = 0.f;
        for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { S1 += chs[2 * c]; Q1 += chs[2 * c + 1]; }
        gn_publish(slot + ((int64_t)sl * gpb + tid) * 2, S1, Q1);
    }
    gn_stores_done();
    __syncthreads();
    if (tid == 0) *flag = gn_ticket(p.counters + img * nsets + set);
    __syncthreads();
    if (*flag != S - 1) return;                   // (uniform over the workgroup)
    // all S * gpb pairs in flight at once (one per thread: a dependent chain of L2 round trips otherwise), then summed in slice order
    for (int t = tid; t < S * gpb; t += 256) {
        float a, b;
        gn_fetch(slot + (int64_t)t * 2, a, b);
        if (t < 2048) { red[2 * t] = a; red[2 * t + 1] = b; }
    }
    __syncthreads();
    if (tid < gpb) {
        float S1 = 0.f, Q1 = 0.f;
        for (int k = 0; k < S; ++k) { S1 += red[(k * gpb + tid) * 2]; Q1 += red[(k * gpb + tid) * 2 + 1]; }
        const float inv = 1.0f / ((float)cpg * (float)p.hw);
        const float mu = S1 * inv;
        gst[2 * tid] = mu;
        gst[2 * tid + 1] = fmaxf(Q1 * inv - mu * mu, 0.f);   // variance for now: the refinement below may replace it
    }
    if (tid == 0) gn_ticket_reset(p.counters + img * nsets + set);
    __syncthreads();
    for (int g = 0; g < gpb; ++g) {                // uniform over the block
        float mu = gst[2 * g], var = gst[2 * g + 1];
        if (var * GnRefine<T>::RATIO < mu * mu) gn_refine_group<T>(p, img, g0 + g, red, mu, var);
        __syncthreads();
        if (tid == 0) { gst[2 * g] = mu; gst[2 * g + 1] = rsqrtf(var + p.eps); }
    }
    __syncthreads();
    for (int c = tid; c < nch; c += 256) {
        const int g = c / cpg, cc = c_first + c;
        const float sc = gst[2 * g + 1] * (c == tid ? gam0 : p.gamma[cc]);
        float* o = p.ss + ((int64_t)img * ct + cc) * 2;
        o[0] = sc;
        o[1] = (c == tid ? bet0 : p.beta[cc]) - gst[2 * g] * sc;
    }
}


// (Round 6, measured negative, removed: GroupNorm statistics + apply as ONE op for the consumers that read a materialised operand.  First
// as a single launch -- the pixel slices of an image meeting at an arrive counter inside the launch, agent-scope publish / ticket / bounded
// poll / sc1 fetch: correct on hardware, 15-18 us per launch even on the smallest tensor (five serial fabric round trips), +0.43 ms per
// step at batch 8 and +0.81 ms at batch 1 against the gn_stats + gn_apply launches it replaced; then as two lean launches (slice sums
// with a pivot shift, finalize-and-apply): +0.11 ms at batch 8, +0.45 ms at batch 1.  profiles/r6b_ab_*, r6c_ab_*.  The sliced statistics
// kernel above already spreads a small tensor over ~1000 workgroups; its ticket tail costs less than a second prologue in the apply.)

// Row softmax with the row held in registers (cols <= 4096, multiple of 4, 16-byte aligned rows): one read of the
// fp32 scores as float4, one write of the probabilities as 4-element vectors.  The generic kernel above makes three
// passes over the row and scalar accesses.
template <typename T>
__global__ __launch_bounds__(256) void softmax_row_kernel(const i2i_softmax_params p) {
    constexpr int JM = 16;                                   // float4 per lane
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= p.rows) return;
    const f32x4* s4 = (const f32x4*)(p.s + row * p.lds);
    const int n4 = p.cols >> 2;
    f32x4 v[JM];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < JM; ++j) {
        const int c = lane + j * 64;
        if (c < n4) {
            v[j] = s4[c];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[j][e] *= p.scale; mx = fmaxf(mx, v[j][e]); }
        }
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < JM; ++j) {
        if (lane + j * 64 < n4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[j][e] = __expf(v[j][e] - mx); sum += v[j][e]; }
        }
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    typedef T tx4 __attribute__((ext_vector_type(4)));
    tx4* o4 = (tx4*)((T*)p.p + row * p.ldp);
    const int np4 = p.ldp >> 2;
#pragma unroll
    for (int j = 0; j < JM; ++j) {
        const int c = lane + j * 64;
        if (c < np4) {
            tx4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (c < n4) ? from_f32<T>(v[j][e] * inv) : from_f32<T>(0.f);
            o4[c] = o;
        }
    }
}

template <typename T>
int gn_stats_t(const i2i_gn_stats_params& p, hipStream_t s) {
    {   // small tensors: one launch (see gn_stats_small_kernel)
        const int ct0 = p.c0 + p.c1, cpg = ct0 / p.groups;
        int gpb = 0;
        for (int cand = 4; cand <= 16 && !gpb; cand <<= 1)      // fewest groups per block whose channels form whole 8-channel units
            if (p.groups % cand == 0 && (cand * cpg) % 8 == 0 && (cand * cpg) / 8 <= 256) gpb = cand;                 // (a block may straddle the two sources: every 8-channel unit picks its own)
        if (!p.finalize_only && gpb && (int64_t)p.hw * ct0 <= (3 << 20)) {
            const size_t smem = (256 * 16 + 2 * gpb * cpg + 2 * gpb + 4) * sizeof(float);
            // pixel slices (needs the ticket counters): as many as `partial` has room for, about 1024 workgroups per launch,
            // at least 64 pixels each
            if (p.counters) {
                const int nsets = p.groups / gpb;
                int S = 1024 / (p.nimg * nsets);
                S = S < p.nparts ? S : p.nparts;
                S = S < p.hw / 64 ? S : p.hw / 64;
                S = S < 2048 / gpb ? S : 2048 / gpb;      // (the last workgroup stages all pairs in its 16 KiB reduction array)
                if (S > 1) {
                    hipLaunchKernelGGL((gn_stats_sliced_kernel<T>), dim3((unsigned)p.nimg, (unsigned)nsets, (unsigned)S), dim3(256), smem, s, p, gpb, S);
                    return i2i::check_launch("gn_stats_sliced");
                }
            }
            hipLaunchKernelGGL((gn_stats_small_kernel<T>), dim3((unsigned)p.nimg, (unsigned)(p.groups / gpb)), dim3(256), smem, s, p, gpb);
            return i2i::check_launch("gn_stats_small");
        }
    }
    const int ct = p.c0 + p.c1;
    if (!p.finalize_only) {
        hipLaunchKernelGGL((gn_partial_kernel<T>), dim3((unsigned)p.nparts, (unsigned)p.nimg), dim3(256), (size_t)ct * 8, s, p);
        const int rc = i2i::check_launch("gn_partial");
        if (rc) return rc;
    }
    int gpb = p.groups;                            // groups per block: 8 when that divides, else all (<= 256)
    if (p.groups % 8 == 0) gpb = 8;
    else if (p.groups % 4 == 0) gpb = 4;
    // conv epilogues hand over thousands of parts per image (2048 at 512x512): fewer groups per block = more blocks and
    // a shorter serial chain per thread (8 parts instead of 64); the summation order stays a function of the shape only
    if (p.nparts >= 1024) gpb = 1;
    else if (p.nparts >= 256 && p.groups % 2 == 0) gpb = 2;
    const size_t red_bytes = (size_t)(256 / gpb) * gpb * 8;
    hipLaunchKernelGGL((gn_finalize_kernel<T>), dim3((unsigned)p.nimg, (unsigned)(p.groups / gpb)), dim3(256),
                       (size_t)gpb * 8 + (red_bytes < 4096 ? 4096 : red_bytes), s, p, gpb);
    return i2i::check_launch("gn_finalize");
}

}  // namespace

extern "C" int i2i_gn_stats(const i2i_gn_stats_params* p, int dtype, void* stream) {
    if (!p || (!p->x0 && !p->finalize_only) || !p->partial || !p->ss || !p->gamma || !p->beta) return i2i::fail(I2I_ERR_BAD_ARG, "gn_stats: null pointer");
    const int ct = p->c0 + p->c1;
    if (p->c0 % 8 || p->c1 % 8 || p->ld0 % 8 || (p->x1 && p->ld1 % 8)) return i2i::fail(I2I_ERR_BAD_ARG, "gn_stats: channels must be multiples of 8");
    if (ct % p->groups || ct > GN_JMAX * 64 * 8 || p->groups > 256 || p->nparts < 1) return i2i::fail(I2I_ERR_BAD_ARG, "gn_stats: bad geometry (ct=%d groups=%d)", ct, p->groups);
    if ((p->c1 != 0) != (p->x1 != nullptr)) return i2i::fail(I2I_ERR_BAD_ARG, "gn_stats: x1/c1 mismatch");
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case I2I_F32: return gn_stats_t<float>(*p, s);
        case I2I_BF16: return gn_stats_t<__bf16>(*p, s);
        case I2I_F16: return gn_stats_t<_Float16>(*p, s);
    }
    return i2i::fail(I2I_ERR_BAD_ARG, "gn_stats: bad dtype");
}

extern "C" int i2i_gn_apply(const i2i_gn_apply_params* p, int dtype, void* stream) {
    if (!p || !p->x || !p->y || !p->ss) return i2i::fail(I2I_ERR_BAD_ARG, "gn_apply: null pointer");
    if (p->c % 8 || p->ldx % 8 || p->ldy % 8 || p->ss_off % 4 || ((uintptr_t)p->y & 15)) return i2i::fail(I2I_ERR_BAD_ARG, "gn_apply: c / ld / offsets must keep 16-byte chunks aligned");
    if (p->x1 && (p->c1 < 8 || p->c1 % 8 || p->ldx1 % 8 || ((uintptr_t)p->x1 & 15))) return i2i::fail(I2I_ERR_BAD_ARG, "gn_apply: the second source needs c1 / ldx1 multiples of 8 and a 16-byte aligned base");
    const int64_t n = (int64_t)p->nimg * p->hw * (p->c + (p->x1 ? p->c1 : 0)) / 8;
    const unsigned grid = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case I2I_F32: hipLaunchKernelGGL((gn_apply_kernel<float>), dim3(grid), dim3(256), 0, s, *p); break;
        case I2I_BF16: hipLaunchKernelGGL((gn_apply_kernel<__bf16>), dim3(grid), dim3(256), 0, s, *p); break;
        case I2I_F16: hipLaunchKernelGGL((gn_apply_kernel<_Float16>), dim3(grid), dim3(256), 0, s, *p); break;
        default: return i2i::fail(I2I_ERR_BAD_ARG, "gn_apply: bad dtype");
    }
    return i2i::check_launch("gn_apply");
}


extern "C" int i2i_layernorm(const i2i_layernorm_params* p, int dtype, void* stream) {
    if (!p || !p->x || !p->y || !p->gamma || !p->beta) return i2i::fail(I2I_ERR_BAD_ARG, "layernorm: null pointer");
    if (p->c % 8 || p->c > LN_JMAX * 64 * 8 || p->ldx % 8 || p->ldy % 8) return i2i::fail(I2I_ERR_BAD_ARG, "layernorm: bad c=%d", p->c);
    if (((uintptr_t)p->gamma | (uintptr_t)p->beta) & 15) return i2i::fail(I2I_ERR_BAD_ARG, "layernorm: gamma / beta must be 16-byte aligned");
    const unsigned grid = (unsigned)((p->rows + 3) / 4);
    hipStream_t s = (hipStream_t)stream;
    const int cc = p->c >> 3, jj = cc <= 64 ? 1 : (cc <= 128 ? 2 : 4);      // chunks per lane
#define I2I_LN_LAUNCH(T)                                                                                             \
    do {                                                                                                              \
        if (jj == 1) hipLaunchKernelGGL((layernorm_kernel<T, 1>), dim3(grid), dim3(256), 0, s, *p);                   \
        else if (jj == 2) hipLaunchKernelGGL((layernorm_kernel<T, 2>), dim3(grid), dim3(256), 0, s, *p);              \
        else hipLaunchKernelGGL((layernorm_kernel<T, 4>), dim3(grid), dim3(256), 0, s, *p);                           \
    } while (0)
    switch (dtype) {
        case I2I_F32: I2I_LN_LAUNCH(float); break;
        case I2I_BF16: I2I_LN_LAUNCH(__bf16); break;
        case I2I_F16: I2I_LN_LAUNCH(_Float16); break;
        default: return i2i::fail(I2I_ERR_BAD_ARG, "layernorm: bad dtype");
    }
#undef I2I_LN_LAUNCH
    return i2i::check_launch("layernorm");
}

extern "C" int i2i_softmax(const i2i_softmax_params* p, int dtype, void* stream) {
    if (!p || !p->s || !p->p) return i2i::fail(I2I_ERR_BAD_ARG, "softmax: null pointer");
    if (p->cols < 1 || p->ldp < p->cols) return i2i::fail(I2I_ERR_BAD_ARG, "softmax: bad cols");
    const unsigned grid = (unsigned)((p->rows + 3) / 4);
    hipStream_t s = (hipStream_t)stream;
    if (p->cols % 4 == 0 && p->cols <= 4096 && p->ldp % 4 == 0 && p->ldp <= 4096 && p->lds % 4 == 0 && (((uintptr_t)p->s & 15) == 0) && (((uintptr_t)p->p & 15) == 0)) {
        switch (dtype) {
            case I2I_F32: hipLaunchKernelGGL((softmax_row_kernel<float>), dim3(grid), dim3(256), 0, s, *p); break;
            case I2I_BF16: hipLaunchKernelGGL((softmax_row_kernel<__bf16>), dim3(grid), dim3(256), 0, s, *p); break;
            case I2I_F16: hipLaunchKernelGGL((softmax_row_kernel<_Float16>), dim3(grid), dim3(256), 0, s, *p); break;
            default: return i2i::fail(I2I_ERR_BAD_ARG, "softmax: bad dtype");
        }
        return i2i::check_launch("softmax_row");
    }
    switch (dtype) {
        case I2I_F32: hipLaunchKernelGGL((softmax_kernel<float>), dim3(grid), dim3(256), 0, s, *p); break;
        case I2I_BF16: hipLaunchKernelGGL((softmax_kernel<__bf16>), dim3(grid), dim3(256), 0, s, *p); break;
        case I2I_F16: hipLaunchKernelGGL((softmax_kernel<_Float16>), dim3(grid), dim3(256), 0, s, *p); break;
        default: return i2i::fail(I2I_ERR_BAD_ARG, "softmax: bad dtype");
    }
    return i2i::check_launch("softmax");
}
