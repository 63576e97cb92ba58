// Implicit-GEMM convolution / linear / batched matmul on MFMA for gfx950 (MI355X).
//
// One kernel family covers every contraction of the Pix2Pix_Turbo / CycleGAN_Turbo forward
// (src/pix2pix_turbo.py:198-203): 3x3 s1/s2 and 1x1 convolutions of the VAE and UNet, every nn.Linear
// of the UNet transformers, and (on the unfused attention path) q.k^T and p.v.  What diffusers/peft run
// as separate library kernels is folded into the operand gathers and the epilogue:
//   A-operand gather : NHWC im2col on the fly, zero padding incl. F.pad(0,1,0,1) (VAE downsamplers),
//                      nearest-2x upsample (Upsample2D), channel concat of two sources (UNet up path),
//                      GroupNorm affine (+SiLU) from precomputed per-(image,channel) scale/shift
//   B operand        : weights [N][K] k-contiguous with the LoRA adapter already merged
//   epilogue         : alpha, bias (per column or per row), residual / skip add (src/model.py:41-43),
//                      GEGLU, fp32 or `dtype` store
//
// Structure (see DESIGN.md "igemm"): workgroup = WM x WN waves, block tile BM x BN, K step = 8 chunks
// of 16 B (64 halves / 32 floats).  Operands are register-staged (global_load_dwordx4 -> optional
// GN/SiLU in registers -> ds_write_b128) into a double-buffered XOR-swizzled LDS image with one barrier
// per K step; the next step's global loads are in flight while the current step's MFMAs issue.  Each
// wave owns a (BM/WM) x (BN/WN) output tile as 16x16 fp32 fragments.
#include "i2i_dev.h"
#include "launch.h"

namespace i2i {
bool conv3x3_halo_eligible(const i2i_igemm_params& p, int dtype);   // conv3x3.hip
int conv3x3_halo(const i2i_igemm_params& p, int dtype, hipStream_t s);
int conv3x3_halo_gn_parts(const i2i_igemm_params& p, int dtype, int groups);
bool conv3x3_w32_eligible(const i2i_igemm_params& p, int dtype);    // conv3x3_w32.hip
int conv3x3_w32(const i2i_igemm_params& p, int dtype, hipStream_t s);
int conv3x3_w32_gn_parts(const i2i_igemm_params& p, int dtype, int groups);
bool conv3x3_w32_auto(const i2i_igemm_params& p, int dtype);        // tile == 0: does the wide-tile kernel take this op?
int igemm_dma_gn_parts(const i2i_igemm_params& p, int dtype, int groups);
bool igemm_dma_eligible(const i2i_igemm_params& p, int dtype);      // gemm_dma.hip
int igemm_dma(const i2i_igemm_params& p, int dtype, hipStream_t s);
bool gemm_w32_eligible(const i2i_igemm_params& p, int dtype);       // gemm_w32.hip
bool gemm_w32_auto(const i2i_igemm_params& p, int dtype);           // tile == 0: does the wide GEMM take this op?
int gemm_w32_gn_parts(const i2i_igemm_params& p, int dtype, int groups);
int gemm_w32(const i2i_igemm_params& p, int dtype, hipStream_t s);
bool conv_narrow_eligible(const i2i_igemm_params& p, int dtype);    // conv_narrow.hip (tile ids 60..69)
bool conv_narrow_auto(const i2i_igemm_params& p, int dtype);
int conv_narrow_gn_parts(const i2i_igemm_params& p, int dtype, int groups);
int conv_narrow(const i2i_igemm_params& p, int dtype, hipStream_t s);
}  // namespace i2i

namespace {
// the one routing decision i2i_igemm / i2i_igemm_route / i2i_igemm_gn_parts share for the wide GEMM (tile ids 50..54)
bool routes_to_gemm_w32(const i2i_igemm_params& p, int dtype) {
    if (p.tile >= 50 && p.tile <= 56) return i2i::gemm_w32_eligible(p, dtype);
    return p.tile == 0 && i2i::gemm_w32_auto(p, dtype);
}
}  // namespace

namespace {

template <typename T, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void igemm_kernel(const i2i_igemm_params p) {
    constexpr int NT = WM * WN * 64;
    constexpr int EPC = Elem<T>::EPC;
    constexpr int BK = 8 * EPC;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int FM = WTM / 16, FN = WTN / 16;
    constexpr int APT = (BM * 8 + NT - 1) / NT;
    constexpr int BPT = (BN * 8 + NT - 1) / NT;
    constexpr int RSTEP = NT / 8;   // row distance between a thread's consecutive chunks
    static_assert(WTM % 16 == 0 && WTN % 16 == 0, "wave tile must be a multiple of 16x16");
    static_assert(NT % 8 == 0, "");
    typedef typename Elem<T>::chunk_t chunk_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int ntn = (p.N + BN - 1) / BN;
    const int tm = blockIdx.x / ntn, tn = blockIdx.x % ntn;
    const int m0 = tm * BM, n0 = tn * BN;
    const int z = blockIdx.y;
    const int zb = z / p.zh_count, zh = z % p.zh_count;

    const int64_t a_off = (int64_t)zb * p.a_bs_b + (int64_t)zh * p.a_bs_h;
    const T* __restrict__ a0 = (const T*)p.a0 + a_off;
    const T* __restrict__ a1 = p.a1 ? (const T*)p.a1 + a_off : nullptr;
    const T* __restrict__ bw = (const T*)p.b + (int64_t)zb * p.b_bs_b + (int64_t)zh * p.b_bs_h;

    char* As = i2i_smem;                          // [2][BM] rows of 128 B
    char* Bs = i2i_smem + 2 * BM * 128;           // [2][BN]

    // ---- per-thread A rows (fixed for the whole K loop) ----
    const int kc = tid & 7;
    const int cin = p.c0 + p.c1;
    const int hin_up = p.up_h ? p.up_h : (p.hin << p.ups), win_up = p.up_w ? p.up_w : (p.win << p.ups);
    int a_img[APT], a_iy0[APT], a_ix0[APT];
    {
        const int hw = p.ho * p.wo;
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int r = (tid >> 3) + i * RSTEP;
            const int m = m0 + r;
            if (r < BM && m < p.M) {
                const int img = m / hw, rem = m - img * hw;
                const int oy = rem / p.wo, ox = rem - oy * p.wo;
                a_img[i] = img;
                a_iy0[i] = oy * p.stride - p.pad;
                a_ix0[i] = ox * p.stride - p.pad;
            } else {
                a_img[i] = 0;
                a_iy0[i] = -(1 << 28);   // never in range
                a_ix0[i] = 0;
            }
        }
    }

    chunk_t ra[APT], rb[BPT];
    unsigned amask = 0;
    float ssr[2 * EPC];     // (scale, shift) of this thread's EPC channels for image ss_img
    int ss_img = -1, ss_ci = 0;
    const bool has_gn = p.gn_ss != nullptr;

    auto load_ss = [&](int img, int ci) {
        const f32x4* s = (const f32x4*)(p.gn_ss + ((int64_t)img * cin + ci) * 2);
#pragma unroll
        for (int q = 0; q < EPC / 2; ++q) {
            f32x4 v = s[q];
            ssr[4 * q + 0] = v[0]; ssr[4 * q + 1] = v[1]; ssr[4 * q + 2] = v[2]; ssr[4 * q + 3] = v[3];
        }
        ss_img = img;
    };

    auto g2r = [&](int kt) {
        const int k = kt * BK + kc * EPC;
        const bool kvalid = k < p.K;
        int tap = 0, ci = k;
        if (p.ks != 1) { tap = k / cin; ci = k - tap * cin; }
        const int ky = tap / p.ks, kx = tap - ky * p.ks;
        const T* __restrict__ src = a0;
        int ld = p.lda0, cc = ci;
        if (ci >= p.c0) { src = a1; ld = p.lda1; cc = ci - p.c0; }
        amask = 0;
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
            const bool ok = kvalid && (unsigned)iy < (unsigned)hin_up && (unsigned)ix < (unsigned)win_up;
            if (ok) {
                const int64_t off = ((int64_t)(a_img[i] * p.hin + up_src(iy, p.hin, hin_up, p.ups)) * p.win + up_src(ix, p.win, win_up, p.ups)) * ld + cc;
                ra[i] = *(const chunk_t*)(src + off);
                amask |= 1u << i;
            } else {
                ra[i] = zero_chunk<T>();
            }
        }
        if (has_gn && kvalid) { ss_ci = ci; load_ss(a_img[0], ci); }
#pragma unroll
        for (int j = 0; j < BPT; ++j) {
            const int v = tid + j * NT;
            const int r = v >> 3, n = n0 + r;
            if (v < BN * 8 && n < p.N && kvalid) rb[j] = *(const chunk_t*)(bw + (int64_t)n * p.ldb + k);
            else rb[j] = zero_chunk<T>();
        }
    };

    auto r2s = [&](int buf) {
        char* Ab = As + buf * BM * 128;
        char* Bb = Bs + buf * BN * 128;
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int r = (tid >> 3) + i * RSTEP;
            chunk_t c = ra[i];
            if (has_gn && ((amask >> i) & 1u)) {
                if (a_img[i] != ss_img) load_ss(a_img[i], ss_ci);   // tile spans images (small planes)
#pragma unroll
                for (int j = 0; j < EPC; ++j) {
                    float v = to_f32<T>(c[j]) * ssr[2 * j] + ssr[2 * j + 1];
                    if (p.act == 1) v = silu_f(v);
                    c[j] = from_f32<T>(v);
                }
            }
            if (r < BM) *(chunk_t*)(Ab + lds_chunk_off(r, kc)) = c;
        }
#pragma unroll
        for (int j = 0; j < BPT; ++j) {
            const int v = tid + j * NT;
            if (v < BN * 8) *(chunk_t*)(Bb + lds_chunk_off(v >> 3, kc)) = rb[j];
        }
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int buf) {
        const char* Ab = As + buf * BM * 128;
        const char* Bb = Bs + buf * BN * 128;
        const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
        for (int kg = 0; kg < 2; ++kg) {
            chunk_t af[FM], bf[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) af[i] = *(const chunk_t*)(Ab + lds_chunk_off(wm * WTM + i * 16 + lr, kg * 4 + lq));
#pragma unroll
            for (int j = 0; j < FN; ++j) bf[j] = *(const chunk_t*)(Bb + lds_chunk_off(wn * WTN + j * 16 + lr, kg * 4 + lq));
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = mma_chunk(af[i], bf[j], acc[i][j]);
        }
    };

    const int nk = (p.K + BK - 1) / BK;
    g2r(0);
    r2s(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) g2r(kt + 1);
        compute(cur);
        if (kt + 1 < nk) r2s(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue ----
    const int64_t c_off = (int64_t)zb * p.c_bs_b + (int64_t)zh * p.c_bs_h;
    const int64_t r_off = (int64_t)zb * p.r_bs_b + (int64_t)zh * p.r_bs_h;
    const T* __restrict__ res = p.res ? (const T*)p.res + r_off : nullptr;
    const int lr = lane & 15, lq = lane >> 4;
    if (p.geglu) {
        // B rows (and bias) are interleaved per 32: [16 value rows | 16 gate rows]; out col = n/2 block
        static_assert(FN % 2 == 0 || FN == 1, "");
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j + 1 < FN; j += 2) {
                const int nA = n0 + wn * WTN + j * 16 + lr;          // packed column of the value
                const int nG = nA + 16;
                const int no = (n0 + wn * WTN + j * 16) / 2 + lr;    // output column
                if (nG < p.N) {
                    const float bA = p.bias ? p.bias[nA] : 0.f, bG = p.bias ? p.bias[nG] : 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = m0 + wm * WTM + i * 16 + lq * 4 + r;
                        if (m < p.M) {
                            const float a = p.alpha * acc[i][j][r] + bA;
                            const float g = p.alpha * acc[i][j + 1][r] + bG;
                            float v = a * gelu_erf_f(g);
                            if (res) v += to_f32<T>(res[(int64_t)m * p.ldr + no]);
                            ((T*)p.c + c_off)[(int64_t)m * p.ldc + no] = from_f32<T>(v);
                        }
                    }
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = n0 + wn * WTN + j * 16 + lr;
            if (n < p.N) {
                const float bn = (p.bias_mode == 1) ? p.bias[n] : 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * WTM + i * 16 + lq * 4 + r;
                    if (m < p.M) {
                        float v = p.alpha * acc[i][j][r] + bn;
                        if (p.bias_mode == 2) v += p.bias[m];
                        if (res) v += to_f32<T>(res[(int64_t)m * p.ldr + n]);
                        if (p.out_f32) ((float*)p.c + c_off)[(int64_t)m * p.ldc + n] = v;
                        else ((T*)p.c + c_off)[(int64_t)m * p.ldc + n] = from_f32<T>(v);
                    }
                }
            }
        }
}

template <typename T, int BM, int BN, int WM, int WN>
int launch_cfg(const i2i_igemm_params& p, hipStream_t s) {
    const unsigned tiles = (unsigned)(((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN));
    const size_t smem = 2 * (BM + BN) * 128;
    hipLaunchKernelGGL((igemm_kernel<T, BM, BN, WM, WN>), dim3(tiles, (unsigned)p.zcount, 1), dim3(WM * WN * 64), smem, s, p);
    return i2i::check_launch("igemm");
}

template <typename T>
int launch_t(const i2i_igemm_params& p, hipStream_t s) {
    int tile = p.tile;
    if (tile == 0) {
        if (p.N <= 16) tile = 4;
        else if (p.N <= 32) tile = 3;
        else if (p.N <= 64 || p.M <= 64) tile = (p.M <= 64) ? 5 : 2;
        else tile = 1;
        if (p.geglu && (tile == 4)) tile = 3;
    }
    switch (tile) {
        case 1: return launch_cfg<T, 128, 128, 2, 2>(p, s);
        case 2: return launch_cfg<T, 128, 64, 2, 2>(p, s);
        case 3: return launch_cfg<T, 128, 32, 4, 1>(p, s);
        case 4: return launch_cfg<T, 256, 16, 4, 1>(p, s);
        case 5: return launch_cfg<T, 64, 64, 2, 2>(p, s);
        default: return i2i::fail(I2I_ERR_BAD_ARG, "igemm: unknown tile config %d", tile);
    }
}

}  // namespace

extern "C" int i2i_igemm(const i2i_igemm_params* pp, int dtype, void* stream) {
    if (!pp) return i2i::fail(I2I_ERR_BAD_ARG, "igemm: null params");
    i2i_igemm_params p = *pp;
    if (!p.a0 || !p.b || !p.c) return i2i::fail(I2I_ERR_BAD_ARG, "igemm: null operand");
    if (p.zcount < 1) p.zcount = 1;
    if (p.zh_count < 1) p.zh_count = 1;
    const int epc = (dtype == I2I_F32) ? 4 : 8;
    const int cin = p.c0 + p.c1;
    if (p.ks != 1 && p.ks != 3) return i2i::fail(I2I_ERR_BAD_ARG, "igemm: ks must be 1 or 3");
    if (p.K != p.ks * p.ks * cin) return i2i::fail(I2I_ERR_BAD_ARG, "igemm: K=%d != ks*ks*cin=%d", p.K, p.ks * p.ks * cin);
    if (p.c0 % epc || p.c1 % epc || p.lda0 % epc || (p.a1 && p.lda1 % epc) || p.ldb % epc || cin % epc)
        return i2i::fail(I2I_ERR_BAD_ARG, "igemm: channels / leading dims must be multiples of %d", epc);
    if ((p.c1 != 0) != (p.a1 != nullptr)) return i2i::fail(I2I_ERR_BAD_ARG, "igemm: a1/c1 mismatch");
    if (p.M != p.nimg * p.ho * p.wo) return i2i::fail(I2I_ERR_BAD_ARG, "igemm: M != nimg*ho*wo");
    if (p.bias_mode && !p.bias) return i2i::fail(I2I_ERR_BAD_ARG, "igemm: bias_mode without bias");
    if (p.geglu && (p.N % 32 || p.out_f32 || p.bias_mode == 2)) return i2i::fail(I2I_ERR_BAD_ARG, "igemm: bad geglu config");
    if (((uintptr_t)p.a0 | (uintptr_t)p.a1 | (uintptr_t)p.b) & 15) return i2i::fail(I2I_ERR_BAD_ARG, "igemm: operands must be 16-byte aligned");
    // the 3x3 conv kernels bring the tile's bias vector into LDS by 16-byte DMA pieces (the other routes take any 4-byte
    // aligned bias: the LDS-DMA igemm falls back to scalar bias loads, the wide GEMM is simply not eligible)
    const bool conv3 = p.ks == 3 && p.stride == 1 && (i2i::conv3x3_halo_eligible(p, dtype) || i2i::conv3x3_w32_eligible(p, dtype));
    if (conv3 && p.bias_mode == 1 && ((uintptr_t)p.bias & 15)) return i2i::fail(I2I_ERR_BAD_ARG, "igemm: the 3x3 conv kernels need a 16-byte aligned bias");
    if (p.bias_mode == 1 && ((uintptr_t)p.bias & 3)) return i2i::fail(I2I_ERR_BAD_ARG, "igemm: bias must be 4-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    if (dtype < I2I_F32 || dtype > I2I_F16) return i2i::fail(I2I_ERR_BAD_ARG, "igemm: bad dtype %d", dtype);
    // 3x3 stride-1 convolutions take the halo-tiled kernel (tile 0 = auto, 10 = force); everything else the generic gather
    if (p.gn_part && !i2i_igemm_gn_parts(&p, dtype, p.gn_part_groups))
        return i2i::fail(I2I_ERR_BAD_ARG, "igemm: gn_part requested but this op cannot produce GroupNorm partials (query i2i_igemm_gn_parts first)");
    if (p.act_out && (!i2i::igemm_dma_eligible(p, dtype) || i2i::conv3x3_halo_eligible(p, dtype)))
        return i2i::fail(I2I_ERR_BAD_ARG, "igemm: act_out is implemented by the LDS-DMA igemm only (no GN prologue, aligned output, not a halo conv)");
    if (p.subpix && !i2i::conv3x3_halo_eligible(p, dtype) && !i2i::conv3x3_w32_eligible(p, dtype))
        return i2i::fail(I2I_ERR_BAD_ARG, "igemm: subpix weights need a 3x3 conv kernel (ups=1, 3x3 s1 p1, cin %% slab == 0, source plane >= 8x16, ldb = 4*cin)");
    if (p.k2_a && !(i2i::conv3x3_w32_eligible(p, dtype) && ((p.tile >= 40 && p.tile <= 49) || (p.tile == 0 && i2i::conv3x3_w32_auto(p, dtype)))))
        return i2i::fail(I2I_ERR_UNSUPPORTED, "igemm: the second contraction (k2_a) is implemented by the wide-tile conv only (query i2i_igemm_route)");
    if ((p.ln_cs || p.n_trans || p.c2) && !routes_to_gemm_w32(p, dtype))
        return i2i::fail(I2I_ERR_UNSUPPORTED, "igemm: the LayerNorm fold (ln_cs / n_trans) is implemented by the wide GEMM only (query i2i_igemm_route)");
    const bool w32_forced = p.tile >= 40 && p.tile <= 49;      // 32x32x16-MFMA wide-tile conv (conv3x3_w32.hip)
    if (w32_forced && !i2i::conv3x3_w32_eligible(p, dtype)) return i2i::fail(I2I_ERR_BAD_ARG, "igemm: tile %d (w32 conv) not applicable", p.tile);
    if (w32_forced || (p.tile == 0 && i2i::conv3x3_w32_auto(p, dtype))) return i2i::conv3x3_w32(p, dtype, s);
    const bool halo_forced = (p.tile >= 10 && p.tile <= 19) || (p.tile >= 30 && p.tile <= 39);
    if (halo_forced && !i2i::conv3x3_halo_eligible(p, dtype)) return i2i::fail(I2I_ERR_BAD_ARG, "igemm: tile %d (halo conv) not applicable", p.tile);
    if ((p.tile == 0 || halo_forced) && i2i::conv3x3_halo_eligible(p, dtype)) return i2i::conv3x3_halo(p, dtype, s);
    // the narrow-input 3x3 conv (VAE conv_in: 8 padded input channels; conv_narrow.hip; tile 0 = auto, 60..69 = force)
    if (p.tile >= 60 && p.tile <= 69 && !i2i::conv_narrow_eligible(p, dtype)) return i2i::fail(I2I_ERR_BAD_ARG, "igemm: tile %d (narrow-input conv) not applicable", p.tile);
    if ((p.tile >= 60 && p.tile <= 69) || (p.tile == 0 && i2i::conv_narrow_auto(p, dtype))) return i2i::conv_narrow(p, dtype, s);
    // plain 16-bit GEMMs that fill the chip: the wide GEMM (32x32x16 MFMA, gemm_w32.hip; tile 0 = auto, 50..56 = force)
    if (p.tile >= 57 && p.tile <= 59) return i2i::fail(I2I_ERR_BAD_ARG, "igemm: tile %d is not a wide-GEMM configuration (50 = auto, 51..56)", p.tile);
    if (p.tile >= 50 && p.tile <= 56 && !i2i::gemm_w32_eligible(p, dtype)) return i2i::fail(I2I_ERR_BAD_ARG, "igemm: tile %d (wide GEMM) not applicable", p.tile);
    if (routes_to_gemm_w32(p, dtype)) return i2i::gemm_w32(p, dtype, s);
    // everything else without a GroupNorm prologue goes through the LDS-DMA engine (tile 0 = auto, 20..29 = force)
    const bool dma_forced = p.tile >= 20 && p.tile <= 29;
    if (dma_forced && !i2i::igemm_dma_eligible(p, dtype)) return i2i::fail(I2I_ERR_BAD_ARG, "igemm: tile %d (LDS-DMA igemm) not applicable", p.tile);
    if ((p.tile == 0 || dma_forced) && i2i::igemm_dma_eligible(p, dtype)) return i2i::igemm_dma(p, dtype, s);
    if (p.splitk > 1) return i2i::fail(I2I_ERR_BAD_ARG, "igemm: split-K needs the LDS-DMA path (no GN prologue, aligned output, ws)");
    switch (dtype) {
        case I2I_F32: return launch_t<float>(p, s);
        case I2I_BF16: return launch_t<__bf16>(p, s);
        case I2I_F16: return launch_t<_Float16>(p, s);
        default: return i2i::fail(I2I_ERR_BAD_ARG, "igemm: bad dtype %d", dtype);
    }
}

extern "C" int i2i_igemm_gn_parts(const i2i_igemm_params* pp, int dtype, int groups) {
    if (!pp) return 0;
    i2i_igemm_params p = *pp;
    if (p.zcount < 1) p.zcount = 1;
    // mirrors the routing of i2i_igemm: halo conv when eligible and not forced elsewhere, else the LDS-DMA igemm
    if ((p.tile >= 40 && p.tile <= 49) || (p.tile == 0 && i2i::conv3x3_w32_auto(p, dtype))) return i2i::conv3x3_w32_gn_parts(p, dtype, groups);
    const bool halo_ok = p.tile == 0 || (p.tile >= 10 && p.tile <= 19) || (p.tile >= 30 && p.tile <= 39);
    if (halo_ok && i2i::conv3x3_halo_eligible(p, dtype)) return i2i::conv3x3_halo_gn_parts(p, dtype, groups);
    if ((p.tile >= 60 && p.tile <= 69) || (p.tile == 0 && i2i::conv_narrow_auto(p, dtype))) return i2i::conv_narrow_gn_parts(p, dtype, groups);
    if (routes_to_gemm_w32(p, dtype)) return i2i::gemm_w32_gn_parts(p, dtype, groups);      // (0 for the 160-column tiles: a planner may force tile 20 instead)
    if (p.tile == 0 || (p.tile >= 20 && p.tile <= 29)) return i2i::igemm_dma_gn_parts(p, dtype, groups);
    return 0;
}

extern "C" const char* i2i_igemm_route(const i2i_igemm_params* pp, int dtype) {
    if (!pp) return "invalid";
    i2i_igemm_params p = *pp;
    if (p.zcount < 1) p.zcount = 1;
    if (p.zh_count < 1) p.zh_count = 1;
    if ((p.tile >= 40 && p.tile <= 49) || (p.tile == 0 && i2i::conv3x3_w32_auto(p, dtype))) return p.subpix ? "conv3x3_w32_kernel<SUBPIX>" : "conv3x3_w32_kernel";
    const bool halo_forced = (p.tile >= 10 && p.tile <= 19) || (p.tile >= 30 && p.tile <= 39);
    if ((p.tile == 0 || halo_forced) && i2i::conv3x3_halo_eligible(p, dtype)) return p.subpix ? "conv3x3_halo_kernel<SUBPIX>" : "conv3x3_halo_kernel";
    if ((p.tile >= 60 && p.tile <= 69 && i2i::conv_narrow_eligible(p, dtype)) || (p.tile == 0 && i2i::conv_narrow_auto(p, dtype))) return "conv_narrow_kernel";
    if (routes_to_gemm_w32(p, dtype)) return "gemm_w32_kernel";
    if ((p.tile == 0 || (p.tile >= 20 && p.tile <= 29)) && i2i::igemm_dma_eligible(p, dtype)) return "igemm_dma_kernel";
    return "igemm_kernel";
}
