// Halo-tiled 3x3 stride-1 pad-1 implicit-GEMM convolution for gfx950 -- the kernel that carries ~85 % of
// the forward's FLOPs (every resnet conv and upsampler conv of the VAE and UNet; F.conv2d in
// diffusers ResnetBlock2D / Upsample2D).
//
// Versus the generic im2col gather of igemm.hip (which restages each input element once per tap), a
// workgroup here owns a TH x TW = 8 x 16 pixel output tile of ONE image and stages, per 64-channel slab
// (32 for f32), the (TH+2) x (TW+2) input halo ONCE: global_load_dwordx4 -> GroupNorm affine + SiLU in
// registers -> ds_write_b128 into an XOR-swizzled LDS image.  The nine taps then read their A fragments
// from that single image at shifted pixel rows, so the prologue VALU work, the A-side global traffic and the
// LDS writes all drop ~6.4x (9 taps / 1.4 halo overhead); only the [BN][64] weight slab changes per tap.
// The halo of the next slab is fetched one chunk per tap-step behind the MFMAs (register-staged, written
// at the end of the step into the other halo buffer); weights are double-buffered per step; one barrier
// per step.  Nearest-2x upsampling is an index map while staging the halo (Upsample2D never materialises).
// Each output m-fragment is one 16-pixel tile row, so the MFMA A operand of tap (dy,dx) is the halo row
// segment starting at (ty+dy, dx): 16 consecutive 128-byte LDS rows.
#include "i2i_dev.h"
#include "launch.h"

namespace {

constexpr int TH = 8, TW = 16, HW2 = TW + 2, HALO = (TH + 2) * (TW + 2);   // 180 halo pixels

template <typename T, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv3x3_halo_kernel(const i2i_igemm_params p) {
    constexpr int NT = 256;
    constexpr int EPC = Elem<T>::EPC;
    constexpr int CK = 8 * EPC;                      // channels per slab: 64 (16-bit) / 32 (f32)
    static_assert(WM * WN == 4 && TH % WM == 0 && BN % (16 * WN) == 0, "");
    constexpr int WTN = BN / WN, FN = WTN / 16;      // wave: 4 tile rows x WTN channels
    constexpr int FM = (TH / WM);                    // m-fragments per wave = tile rows per wave
    constexpr int BPT = (BN * 8 + NT - 1) / NT;
    constexpr int HPT = (HALO * 8 + NT - 1) / NT;    // halo chunks per thread per slab (6)
    typedef typename Elem<T>::chunk_t chunk_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 15, lq = lane >> 4;
    const int kc = tid & 7;

    const int tiles_x = (p.wo + TW - 1) / TW, tiles_y = (p.ho + TH - 1) / TH;
    const int ntn = (p.N + BN - 1) / BN;
    int bid = blockIdx.x;
    const int tn = bid % ntn; bid /= ntn;
    const int tx0 = (bid % tiles_x) * TW; bid /= tiles_x;
    const int ty0 = (bid % tiles_y) * TH;
    const int img = bid / tiles_y;
    const int n0 = tn * BN;

    const T* __restrict__ a0 = (const T*)p.a0;
    const T* __restrict__ a1 = (const T*)p.a1;
    const T* __restrict__ bw = (const T*)p.b;
    const int cin = p.c0 + p.c1;
    const int hin_up = p.hin << p.ups, win_up = p.win << p.ups;
    const bool has_gn = p.gn_ss != nullptr;

    char* Hs = i2i_smem;                                   // [2][HALO] rows of 128 B
    char* Bs = i2i_smem + 2 * HALO * 128;                  // [2][BN]   rows of 128 B

    // this thread's halo chunks: chunk id v = tid + j*NT -> halo pixel v>>3 (kc is constant = tid&7)
    int h_off[HPT];       // element offset of the source pixel (without channel), -1 if outside the image
#pragma unroll
    for (int j = 0; j < HPT; ++j) {
        const int hp = (tid >> 3) + j * (NT / 8);
        int off = -1;
        if (hp < HALO) {
            const int hy = hp / HW2, hx = hp - hy * HW2;
            const int iy = ty0 + hy - 1, ix = tx0 + hx - 1;           // coordinates in the (upsampled) input plane
            if ((unsigned)iy < (unsigned)hin_up && (unsigned)ix < (unsigned)win_up)
                off = (img * p.hin + (iy >> p.ups)) * p.win + (ix >> p.ups);
        }
        h_off[j] = off;
    }

    float ssr[2 * EPC];
    chunk_t rb[BPT];
    chunk_t rh = zero_chunk<T>();   // the one halo chunk in flight during a step

    const int nslab = cin / CK;
    const int nstep = nslab * 9;

    auto load_ss = [&](int slab) {
        const f32x4* s = (const f32x4*)(p.gn_ss + ((int64_t)img * cin + slab * CK + kc * EPC) * 2);
#pragma unroll
        for (int q = 0; q < EPC / 2; ++q) {
            const f32x4 v = s[q];
            ssr[4 * q + 0] = v[0]; ssr[4 * q + 1] = v[1]; ssr[4 * q + 2] = v[2]; ssr[4 * q + 3] = v[3];
        }
    };
    auto halo_load = [&](int slab, int j) -> chunk_t {
        const int ci = slab * CK + kc * EPC;
        const int off = h_off[j];
        if (off < 0) return zero_chunk<T>();
        if (ci < p.c0) return *(const chunk_t*)(a0 + (int64_t)off * p.lda0 + ci);
        return *(const chunk_t*)(a1 + (int64_t)off * p.lda1 + (ci - p.c0));
    };
    auto halo_store = [&](int buf, int j, chunk_t c) {
        const int hp = (tid >> 3) + j * (NT / 8);
        if (hp >= HALO) return;
        if (has_gn && h_off[j] >= 0) {       // zero padding stays exactly zero (conv pads the ACTIVATED tensor)
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                float v = to_f32<T>(c[e]) * ssr[2 * e] + ssr[2 * e + 1];
                if (p.act == 1) v = silu_f(v);
                c[e] = from_f32<T>(v);
            }
        }
        *(chunk_t*)(Hs + buf * HALO * 128 + lds_chunk_off(hp, kc)) = c;
    };
    auto b_load = [&](int step) {
        const int slab = step / 9, tap = step - slab * 9;
        const int k = tap * cin + slab * CK + kc * EPC;
#pragma unroll
        for (int j = 0; j < BPT; ++j) {
            const int v = tid + j * NT;
            const int n = n0 + (v >> 3);
            rb[j] = (v < BN * 8 && n < p.N) ? *(const chunk_t*)(bw + (int64_t)n * p.ldb + k) : zero_chunk<T>();
        }
    };
    auto b_store = [&](int buf) {
#pragma unroll
        for (int j = 0; j < BPT; ++j) {
            const int v = tid + j * NT;
            if (v < BN * 8) *(chunk_t*)(Bs + buf * BN * 128 + lds_chunk_off(v >> 3, kc)) = rb[j];
        }
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: slab 0 halo + step 0 weights ----
    if (has_gn) load_ss(0);
#pragma unroll
    for (int j = 0; j < HPT; ++j) halo_store(0, j, halo_load(0, j));
    b_load(0);
    b_store(0);
    __syncthreads();

    for (int slab = 0; slab < nslab; ++slab) {
        const int hb = slab & 1;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {          // fully unrolled: tap, dy, dx and every h_off[] index are constants
            const int step = slab * 9 + tap;
            const int dy = tap / 3, dx = tap - dy * 3;
            const int bb = step & 1;
            const bool more = step + 1 < nstep;
            // -- issue next step's global loads: weights always, one halo chunk of the NEXT slab on taps 0..HPT-1
            if (more) b_load(step + 1);
            const bool halo_pf = (tap < HPT) && (slab + 1 < nslab);
            if (tap < HPT) {
                if (halo_pf) {
                    if (tap == 0 && has_gn) load_ss(slab + 1);   // previous slab's halo is complete: ssr is free
                    rh = halo_load(slab + 1, tap < HPT ? tap : 0);
                }
            }
            // -- MFMAs of this (slab, tap): A = halo rows (ty+dy, dx .. dx+15), B = weight slab
            {
                const char* Hb = Hs + hb * HALO * 128;
                const char* Bb = Bs + bb * BN * 128;
#pragma unroll
                for (int kg = 0; kg < 2; ++kg) {
                    chunk_t af[FM], bf[FN];
#pragma unroll
                    for (int i = 0; i < FM; ++i) {
                        const int hp = (wm * FM + i + dy) * HW2 + dx + lr;
                        af[i] = *(const chunk_t*)(Hb + lds_chunk_off(hp, kg * 4 + lq));
                    }
#pragma unroll
                    for (int j = 0; j < FN; ++j) bf[j] = *(const chunk_t*)(Bb + lds_chunk_off(wn * WTN + j * 16 + lr, kg * 4 + lq));
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int j = 0; j < FN; ++j) acc[i][j] = mma_chunk(af[i], bf[j], acc[i][j]);
                }
            }
            // -- land the prefetched data in the other buffers
            if (more) b_store(bb ^ 1);
            if (tap < HPT) {
                if (halo_pf) halo_store(hb ^ 1, tap < HPT ? tap : 0, rh);
            }
            __syncthreads();
        }
    }

    // ---- epilogue: alpha, bias, residual, store (row = tile row wm*FM+i, pixel tx = 4*lq + r) ----
    const T* __restrict__ res = (const T*)p.res;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int oy = ty0 + wm * FM + i;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = n0 + wn * WTN + j * 16 + lr;
            if (n < p.N && oy < p.ho) {
                const float bn = (p.bias_mode == 1) ? p.bias[n] : 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ox = tx0 + lq * 4 + r;
                    if (ox < p.wo) {
                        const int64_t m = ((int64_t)img * p.ho + oy) * p.wo + ox;
                        float v = p.alpha * acc[i][j][r] + bn;
                        if (res) v += to_f32<T>(res[m * p.ldr + n]);
                        if (p.out_f32) ((float*)p.c)[m * p.ldc + n] = v;
                        else ((T*)p.c)[m * p.ldc + n] = from_f32<T>(v);
                    }
                }
            }
        }
    }
}

template <typename T, int BN, int WM, int WN>
int launch_halo(const i2i_igemm_params& p, hipStream_t s) {
    const unsigned tiles = (unsigned)(((p.wo + TW - 1) / TW) * ((p.ho + TH - 1) / TH) * p.nimg * ((p.N + BN - 1) / BN));
    const size_t smem = 2 * (HALO + BN) * 128;
    hipLaunchKernelGGL((conv3x3_halo_kernel<T, BN, WM, WN>), dim3(tiles), dim3(256), smem, s, p);
    return i2i::check_launch("conv3x3_halo");
}

template <typename T>
int launch_halo_t(const i2i_igemm_params& p, hipStream_t s) {
    if (p.N <= 16) return launch_halo<T, 16, 4, 1>(p, s);
    if (p.N <= 64) return launch_halo<T, 64, 2, 2>(p, s);
    return launch_halo<T, 128, 2, 2>(p, s);
}

}  // namespace

namespace i2i {
// Eligibility: 3x3, stride 1, pad 1, slab-aligned channel counts, a plane at least one tile wide.
bool conv3x3_halo_eligible(const i2i_igemm_params& p, int dtype) {
    const int ck = (dtype == I2I_F32) ? 32 : 64;
    if (p.ks != 3 || p.stride != 1 || p.pad != 1 || p.geglu || p.zcount > 1 || p.bias_mode == 2) return false;
    if (p.c0 % ck || p.c1 % ck || (p.c0 + p.c1) < ck) return false;
    if (p.wo < TW || p.ho < TH) return false;
    if (p.ho != (p.hin << p.ups) || p.wo != (p.win << p.ups)) return false;
    return true;
}
int conv3x3_halo(const i2i_igemm_params& p, int dtype, hipStream_t s) {
    switch (dtype) {
        case I2I_F32: return launch_halo_t<float>(p, s);
        case I2I_BF16: return launch_halo_t<__bf16>(p, s);
        case I2I_F16: return launch_halo_t<_Float16>(p, s);
    }
    return fail(I2I_ERR_BAD_ARG, "conv3x3: bad dtype");
}
}  // namespace i2i
