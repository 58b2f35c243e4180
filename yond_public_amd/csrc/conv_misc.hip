// Edge layers of the denoiser (first / last convolution, 2x2 max pooling) and the sigma-conditioning MLPs.
#include "common.h"

// ------------------------------------------------------------------------------------------------------
// First layer (archs/Unet.py:431): 3x3, Cin = 4 (one 16-byte slot per pixel), Cout = 32*k.
// K = 9 taps x 4 channels is walked as 5 tap PAIRS: half h of the wave takes tap 2u+h, so one MFMA
// k-step covers channel t of both taps.  Tap 9 does not exist: its weights are zero and its A operand
// re-reads tap 8.  Memory bound (16 B in, 4*Cout B out per pixel).
// ------------------------------------------------------------------------------------------------------
template <int TH>
__global__ __launch_bounds__(256) void conv_in_kernel(const float* __restrict__ x, const float* __restrict__ ub,
                                                      int H, int W, int Cout, const float* __restrict__ wpk,
                                                      const float* __restrict__ bias, float slope,
                                                      float* __restrict__ dst, int out_p4) {
    constexpr int IH = TH + 2, IW = 34, MW = TH / 4;
    __shared__ __attribute__((aligned(16))) float s_in[IH * IW * 4];
    __shared__ __attribute__((aligned(16))) float s_w[5 * 2 * 32 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int nct = Cout / 32;
    const int ntx = (W + 31) / 32;
    int b = blockIdx.x;
    const int ct = b % nct;
    b /= nct;
    const int tx = b % ntx, ty = b / ntx;
    const int n = blockIdx.y;
    const int ox0 = tx * 32, oy0 = ty * TH;
    const float u = ub ? ub[n] : 1.0f;
    for (int it = tid; it < IH * IW; it += 256) {
        const int py = it / IW, px = it % IW;
        const int gy = oy0 - 1 + py, gx = ox0 - 1 + px;
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            v = *(const f32x4*)(x + ((size_t)(n * H + gy) * W + gx) * 4);
            if (ub) { v[0] /= u; v[1] /= u; v[2] /= u; v[3] /= u; }
        }
        *(f32x4*)(s_in + it * 4) = v;
    }
    for (int it = tid; it < 5 * 2 * 32; it += 256)
        *(f32x4*)(s_w + it * 4) = *(const f32x4*)(wpk + (size_t)ct * 1280 + it * 4);
    __syncthreads();

    f32x16 acc[MW];
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;
#pragma unroll
    for (int up = 0; up < 5; ++up) {
        int tap = 2 * up + lh;
        if (tap > 8) tap = 8;
        const int dy = tap / 3, dx = tap % 3;
        const f32x4 bb = *(const f32x4*)(s_w + ((up * 2 + lh) * 32 + li) * 4);
#pragma unroll
        for (int m = 0; m < MW; ++m) {
            const f32x4 a = *(const f32x4*)(s_in + ((wave * MW + m + dy) * IW + li + dx) * 4);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(bb[t], a[t], acc[m], 0, 0, 0);   // D = W . X^T
        }
    }
    // D rows = output channels (r&3) + 8*(r>>2) + 4*lh, column = pixel li: four 16-byte stores per lane
    const int ox = ox0 + li;
#pragma unroll
    for (int m = 0; m < MW; ++m) {
        const int oy = oy0 + wave * MW + m;
        if (oy < H && ox < W) {
            // [N][H][W][Cout], or planes of 4 channels [N][Cout/4][H*W][4] (YOND_FMT_PLANES4: a wave stores 512 contiguous bytes)
            float* op = out_p4 ? dst + (((size_t)n * (Cout / 4) + ct * 8 + lh) * H * W + (size_t)oy * W + ox) * 4
                               : dst + ((size_t)(n * H + oy) * W + ox) * Cout + ct * 32 + 4 * lh;
            const size_t gstep = out_p4 ? (size_t)8 * H * W : 8;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
                const f32x4 bv = bias ? *(const f32x4*)(bias + ct * 32 + 4 * lh + 8 * g) : zero;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = acc[m][4 * g + e] + bv[e];
                    v[e] = x > 0.0f ? x : x * slope;
                }
                *(f32x4*)(op + g * gstep) = v;
            }
        }
    }
}

extern "C" int yond_pack_conv_in_weight_f32(const float* w, int cout, float* dst) {
    if (!w || !dst || cout % 32 != 0) return YOND_EINVAL;
    size_t o = 0;
    for (int ct = 0; ct < cout / 32; ++ct)
        for (int up = 0; up < 5; ++up)
            for (int h = 0; h < 2; ++h)
                for (int j = 0; j < 32; ++j)
                    for (int e = 0; e < 4; ++e) {
                        const int tap = 2 * up + h, co = ct * 32 + j;
                        dst[o++] = tap < 9 ? w[((size_t)co * 4 + e) * 9 + tap] : 0.0f;
                    }
    return YOND_OK;
}

extern "C" int yond_conv_in_f32(const float* x, const float* ub, int N, int H, int W, int Cout, const float* wpk,
                                const float* bias, float slope, float* dst, int out_fmt, void* stream) {
    if (!x || !wpk || !dst || N <= 0 || H <= 0 || W <= 0 || N > 65535) return YOND_EINVAL;
    if (out_fmt != YOND_FMT_NHWC_F32 && out_fmt != YOND_FMT_PLANES4) return YOND_EINVAL;
    if (Cout % 32 != 0) return YOND_EUNSUPPORTED;
    constexpr int TH = 8;
    dim3 grid((Cout / 32) * ((W + 31) / 32) * ((H + TH - 1) / TH), N);
    hipLaunchKernelGGL(conv_in_kernel<TH>, grid, dim3(256), 0, (hipStream_t)stream, x, ub, H, W, Cout, wpk, bias, slope, dst,
                       out_fmt == YOND_FMT_PLANES4 ? 1 : 0);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

// ------------------------------------------------------------------------------------------------------
// Last layer (archs/Unet.py:463-468): out = (W10 . feat + b10 + x/ub) * ub, Cout = 4.  Eight lanes share a
// pixel: lane part p reads the 16-byte slot p of every 32-channel group (a wave reads 1 KiB contiguous),
// partial dot products are combined with three xor-shuffles.  Memory bound (4*Cin + 32 B per pixel).
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_out_kernel(const float* __restrict__ feat, int Cin,
                                                       const float* __restrict__ w, const float* __restrict__ bias,
                                                       const float* __restrict__ x, const float* __restrict__ ub,
                                                       size_t npix_per_image, float* __restrict__ dst) {
    const int n = blockIdx.y;
    const int part = threadIdx.x & 7;
    const float u = ub ? ub[n] : 1.0f;
    const size_t stride = (size_t)gridDim.x * 32;
    for (size_t p = (size_t)blockIdx.x * 32 + (threadIdx.x >> 3); p < npix_per_image; p += stride) {
        const size_t gp = (size_t)n * npix_per_image + p;
        const float* f = feat + gp * Cin + part * 4;
        float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
        for (int c = 0; c < Cin; c += 32) {
            const f32x4 v = *(const f32x4*)(f + c);
            const int cb = c + part * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o0 = fmaf(v[e], w[0 * Cin + cb + e], o0);
                o1 = fmaf(v[e], w[1 * Cin + cb + e], o1);
                o2 = fmaf(v[e], w[2 * Cin + cb + e], o2);
                o3 = fmaf(v[e], w[3 * Cin + cb + e], o3);
            }
        }
#pragma unroll
        for (int m = 1; m < 8; m <<= 1) {
            o0 += __shfl_xor(o0, m); o1 += __shfl_xor(o1, m); o2 += __shfl_xor(o2, m); o3 += __shfl_xor(o3, m);
        }
        if (part == 0) {
            if (bias) { o0 += bias[0]; o1 += bias[1]; o2 += bias[2]; o3 += bias[3]; }
            if (x) {
                f32x4 xv = *(const f32x4*)(x + gp * 4);
                if (ub) { xv[0] /= u; xv[1] /= u; xv[2] /= u; xv[3] /= u; }
                o0 += xv[0]; o1 += xv[1]; o2 += xv[2]; o3 += xv[3];
            }
            if (ub) { o0 *= u; o1 *= u; o2 *= u; o3 *= u; }
            f32x4 o = {o0, o1, o2, o3};
            *(f32x4*)(dst + gp * 4) = o;
        }
    }
}

extern "C" int yond_conv_out_f32(const float* feat, int Cin, const float* w, const float* bias, const float* x,
                                 const float* ub, int N, int H, int W, float* dst, void* stream) {
    if (!feat || !w || !dst || N <= 0 || H <= 0 || W <= 0 || N > 65535) return YOND_EINVAL;
    if (Cin % 32 != 0) return YOND_EUNSUPPORTED;
    const size_t npix = (size_t)H * W;
    size_t nb = (npix + 31) / 32;
    if (nb > 256 * 32) nb = 256 * 32;
    dim3 grid((unsigned)nb, N);
    hipLaunchKernelGGL(conv_out_kernel, grid, dim3(256), 0, (hipStream_t)stream, feat, Cin, w, bias, x, ub, npix, dst);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

// 2x2 max pooling, NHWC, 4 channels per thread.
__global__ __launch_bounds__(256) void maxpool2_kernel(const float* __restrict__ src, int H, int W, int C,
                                                       float* __restrict__ dst, size_t total4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int c4 = C / 4;
    const int Ho = H / 2, Wo = W / 2;
    const int c = (int)(i % c4);
    size_t p = i / c4;
    const int ox = (int)(p % Wo);
    p /= Wo;
    const int oy = (int)(p % Ho);
    const int n = (int)(p / Ho);
    const float* s = src + (((size_t)(n * H + 2 * oy) * W + 2 * ox) * C) + c * 4;
    const f32x4 a = *(const f32x4*)s, b2 = *(const f32x4*)(s + C);
    const f32x4 c2 = *(const f32x4*)(s + (size_t)W * C), d2 = *(const f32x4*)(s + (size_t)W * C + C);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = fmaxf(fmaxf(a[e], b2[e]), fmaxf(c2[e], d2[e]));
    *(f32x4*)(dst + i * 4) = o;
}

extern "C" int yond_maxpool2_f32(const float* src, int N, int H, int W, int C, float* dst, void* stream) {
    if (!src || !dst || N <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1) || C % 4 != 0) return YOND_EINVAL;
    const size_t total4 = (size_t)N * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(maxpool2_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, H, W, C, dst, total4);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

// ------------------------------------------------------------------------------------------------------
// sigma-conditioning MLPs (archs/modules.py:170-178, 190-193 / 205-214, 225-231).
//   h = SiLU(w_a0*t + b_a0); m1 = W_a2 h + b_a2; then
//   guided: tb = W_b SiLU(m1) + b_b;  (s1,t1) = (m1, cb1*m1 + tb);  (s2,t2) = (1, cb2)
//   snr:    g = SiLU(w_b0*t + b_b0); m2 = W_b g + b_b;  (s1,t1) = (m1, cb1*m1);  (s2,t2) = (m2, cb2*m2)
// Two launches (the second mat-vec needs all of m1), each gridded over (32-row tile, block, image): a
// workgroup rebuilds the C-vector it multiplies by (C SiLUs) in LDS and its 4 waves reduce 8 rows each with
// coalesced row reads + shuffles.  ~1.5 MFLOP in total: launch-latency bound (a few microseconds).
// ------------------------------------------------------------------------------------------------------
template <int STAGE>
__global__ __launch_bounds__(256) void film_kernel(const YondFilmDesc* __restrict__ descs, const float* __restrict__ t,
                                                   const float* __restrict__ ub) {
    __shared__ float s_h[1024];
    const YondFilmDesc d = descs[blockIdx.y];
    const int n = blockIdx.z;
    const int C = d.C;
    if ((int)blockIdx.x * 32 >= C) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float tv = t[n];
    if (ub) tv = tv / ub[n];
    const float* Wm;
    if (STAGE == 0) {
        for (int c = tid; c < C; c += 256) s_h[c] = silu_f(fmaf(d.w_a0[c], tv, d.b_a0[c]));
        Wm = d.w_a2;
    } else {
        if (d.kind == 0) {
            for (int c = tid; c < C; c += 256) s_h[c] = silu_f(d.s1[(size_t)n * d.ld + c]);   // m1 from stage 0
        } else {
            for (int c = tid; c < C; c += 256) s_h[c] = silu_f(fmaf(d.w_b0[c], tv, d.b_b0[c]));
        }
        Wm = d.w_b;
    }
    __syncthreads();
    // (a batch launches gridDim.x < 32 row tiles per (block, image): a workgroup then walks the tiles blockIdx.x, + gridDim.x, ... -- 32 workgroups
    // per (block, image) of which at most C / 32 do anything were 9,216 workgroups at batch 32, 5 of 6 of them dead on arrival)
    for (int row0 = blockIdx.x * 32; row0 < C; row0 += gridDim.x * 32) {
    // a wave owns 8 rows and walks them together: 8 independent loads per column step instead of 8 latency-bound passes
    float acc8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc8[i] = 0.0f;
    const int rbase = row0 + wave * 8;
    for (int j = lane; j < C; j += 64) {
        const float hj = s_h[j];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = min(rbase + i, C - 1);                     // rows past the end repeat the last one (not stored)
            acc8[i] = fmaf(Wm[(size_t)row * C + j], hj, acc8[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = rbase + i;
        float s = acc8[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (row >= C) continue;
        if (lane == 0) {
            const size_t o = (size_t)n * d.ld + row;
            if (STAGE == 0) {
                d.s1[o] = s + d.b_a2[row];
            } else {
                const float m2 = s + d.b_b[row];
                const float m1 = d.s1[o];
                if (d.kind == 0) {
                    d.t1[o] = fmaf(d.cb1[row], m1, m2);
                    d.s2[o] = 1.0f;
                    d.t2[o] = d.cb2[row];
                } else {
                    d.t1[o] = d.cb1[row] * m1;
                    d.s2[o] = m2;
                    d.t2[o] = d.cb2[row] * m2;
                }
            }
        }
    }
    }
}

extern "C" int yond_film_f32(const YondFilmDesc* descs, int nblocks, const float* t, const float* ub, int N, void* stream) {
    if (!descs || !t || nblocks <= 0 || N <= 0 || N > 65535 || nblocks > 65535) return YOND_EINVAL;
    dim3 grid(N >= 8 ? (unsigned)yond_exp_long("YOND_FILM_GX", 8) : 1024 / 32, nblocks, N);      // (row tiles per (block, image): all 32 for a few images -- latency --, 8 for a batch)
    hipLaunchKernelGGL(film_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, descs, t, ub);
    YOND_LAUNCH_CHECK();
    hipLaunchKernelGGL(film_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, descs, t, ub);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}
