// K1 / K4 and the small layout kernels around the denoiser: pure streaming, HBM bound.
//   K1  Bayer -> pack -> *scale -> bias LUT -> VST -> normalise -> reflect pad -> clamp -> NHWC4 (+ image max)
//   K4  NHWC4 -> clamp -> crop -> de-normalise -> inverse VST -> unpack -> /scale -> clip -> Bayer
// One thread per packed pixel: it reads/writes one 2x2 Bayer quad (two 8-byte accesses on adjacent rows,
// coalesced across the wave) and one 16-byte NHWC4 slot.  Arithmetic follows the dtype staging of the
// reference under NumPy 2: float32 x*scale, float64 for everything up to the single rounding to float32
// (YOND_SIDD.py:251-269, 292-299; utils/isp_algos.py:5-33).
#include <stdlib.h>
#include "common.h"

#include "lut_table.h"

__device__ __forceinline__ double lut_eval(const LutLds& L, int n, float xq) {
    // scipy interp1d(kind='linear')._call_linear: hi = clip(searchsorted(x, xq, 'left'), 1, n-1)
    int g;
    const int nseg = L.nseg;
    if (nseg > 0) {
        int sgm = 0;
        for (int i = 1; i < nseg; ++i) sgm += (xq >= L.seg_x[i]) ? 1 : 0;
        const float t = (xq - L.seg_x[sgm]) * L.seg_inv[sgm];
        const int i0 = L.seg_i[sgm], i1 = L.seg_i[sgm + 1];
        const int off = t > 0.0f ? (t < 1e9f ? (int)ceilf(t) : 1000000000) : 0;
        g = min(i0 + off, i1);
    } else {
        const double x = (double)xq;
        int lo = 0, hi = n;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (L.x[mid] < x) lo = mid + 1; else hi = mid; }
        g = lo;
    }
    const int ih = g < 1 ? 1 : (g > n - 1 ? n - 1 : g);
    const double2 c = L.ab[ih];
    return c.x + c.y * (double)xq;
}

// Foi's closed-form bias of the generalized Anscombe transform (utils/isp_algos.py:84-96)
__device__ __forceinline__ double close_form_bias_dev(double x, double sigma, double gain) {
    const double y = x / gain, sg = sigma / gain;
    const double yh = y + 0.375 + sg * sg;
    const double m1 = (y + sg * sg) / (yh * yh);
    const double m2 = y / (yh * yh * yh);
    const double q = y + sg * sg;
    const double m3 = (y + 3.0 * q * q) / (yh * yh * yh * yh);
    return 2.0 * sqrt(yh) * (-0.125 * m1 + 0.0625 * m2 - 0.0390625 * m3);
}

// 2-D BiasLUT semantics (flags & LUT_BIASLUT): inside the knots as lut_eval; beyond the last knot the reference's index
// position is clipped (constant = last ordinate) until it reaches x_len, from where get_bias_points' closed form takes over
__device__ __forceinline__ double biaslut_eval(const LutLds& L, int n, float xq, double gain, double sigma) {
    const double x = (double)xq, xl = L.x[n - 1];
    if (x > xl) {
        const double2 c = L.ab[n - 1];
        if (x - xl >= xl - L.x[n - 2]) return (double)(float)close_form_bias_dev(x, sigma, gain);
        return c.x + c.y * xl;                                   // the last ordinate
    }
    return lut_eval(L, n, xq);
}

// sqrt of a non-negative float64 to ~1e-15 relative: the float32 hardware estimate (1 ulp of float32) and one Newton step
// in float64; the full-precision float64 sqrt sequence costs four times as much and K1 rounds its result to float32 anyway
__device__ __forceinline__ double sqrt_newton(double a) {
    const float af = (float)a;
    if (!(af > 1e-30f)) return sqrt(a);                            // tiny / zero / huge arguments: the plain routine
    if (af > 1e30f) return sqrt(a);
    const float s0f = __builtin_amdgcn_sqrtf(af);
    const double s0 = (double)s0f;
    const double r = (double)(0.5f * __builtin_amdgcn_rcpf(s0f));  // 1 / (2 s0) to float32 accuracy
    const double e = fma(-s0, s0, a);                              // a - s0^2, exact to float64
    return fma(e, r, s0);
}

__global__ __launch_bounds__(256) void pack_vst_norm_kernel(const float* __restrict__ bayer, int H, int W,
                                                            float* __restrict__ out, int pad_l, int pad_t, int Hp,
                                                            int Wp, int mode, float scale_f, double gain, double sigma,
                                                            double lo, double hi, const double* __restrict__ lut_x,
                                                            const void* __restrict__ lut_y, int lut_n, int lut_flags,
                                                            unsigned int* __restrict__ img_max,
                                                            const double* __restrict__ prm, const void* __restrict__ lut_ws, int lut_cap,
                                                            int B) {
    // B frames of the same size that share every constant and the LUT (the 32 blocks of a SIDD image, YOND_SIDD.py:392-407):
    // bayer [B][H][W], out [B][Hp][Wp][4], img_max [B]; a workgroup takes a contiguous range of the B * Hp output rows
    extern __shared__ __attribute__((aligned(16))) unsigned char lut_raw[];
    __shared__ LutLds L;
    __shared__ float s_red[4];
    if (prm) {
        // the frame's constants from its parameter block (frame_chain.hip); a flagged frame is left to the host path
        const int fl = (int)prm[YOND_PRM_FLAGS];
        if (fl & (YOND_PRM_FLAG_BAD_ESTIMATE | YOND_PRM_FLAG_LUT_CAPACITY | YOND_PRM_FLAG_NO_FLAT_AREA)) return;
        gain = prm[YOND_PRM_GAIN]; sigma = prm[YOND_PRM_SIGMA]; lo = prm[YOND_PRM_LO]; hi = prm[YOND_PRM_HI];
        lut_n = lut_ws ? (int)prm[YOND_PRM_LUT_N] : 0;
        if (lut_n > lut_cap) return;                          // (cannot happen: the block's producer checked the same capacity)
    }
    if (threadIdx.x == 0) {
        L.ab = (double2*)lut_raw;
        L.x = lut_ws ? nullptr : (double*)(L.ab + lut_n);
    }
    __syncthreads();
    if (lut_ws) { if (lut_n > 0) lut_load(L, lut_ws, lut_n); }
    else if (lut_n > 0) lut_prepare(L, lut_x, lut_y, lut_n, lut_flags);
    const int h = H / 2, w = W / 2;
    const double c0 = 0.375 * gain * gain;       // (3/8)*gain**2
    const double s2 = sigma * sigma;
    const double two_over_gain = 2.0 / gain;
    const double inv_span = 1.0 / (hi - lo);
    float vmax = 0.0f;
    // the frame's maximum: per frame, flushed whenever the workgroup's row range moves on to the next frame
    auto flush_max = [&](int b) {
        if (!img_max || b < 0) return;
        const float m0 = wave_max(vmax);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m0;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
            atomicMax(img_max + b, __float_as_uint(m));                  // values are >= 0: uint order == float order
        }
    };
    // rows over the workgroups (a contiguous range each), columns over the threads: no 64-bit divisions per pixel
    const int rows_all = B * Hp, per = (rows_all + (int)gridDim.x - 1) / (int)gridDim.x;
    const int row_lo = blockIdx.x * per, row_hi = row_lo + per < rows_all ? row_lo + per : rows_all;
    int cur_b = -1;
    for (int row = row_lo; row < row_hi; ++row) {
    const int bi = row / Hp, yp = row - bi * Hp;
    if (bi != cur_b) { flush_max(cur_b); vmax = 0.0f; cur_b = bi; }
    const float* bayer_b = bayer + (size_t)bi * H * W;
    float* out_b = out + (size_t)bi * Hp * Wp * 4;
    for (int xp = threadIdx.x; xp < Wp; xp += 256) {
        const size_t p = (size_t)yp * Wp + xp;
        int sy = yp - pad_t, sx = xp - pad_l;
        if ((unsigned)sy >= (unsigned)h) sy = reflect101(sy, h);            // (uniform)
        if ((unsigned)sx >= (unsigned)w) sx = reflect101(sx, w);
        const f32x2 r0 = *(const f32x2*)(bayer_b + (size_t)(2 * sy) * W + 2 * sx);
        const f32x2 r1 = *(const f32x2*)(bayer_b + (size_t)(2 * sy + 1) * W + 2 * sx);
        float q[4] = {r0[0], r0[1], r1[0], r1[1]};
        f32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float u;
            if (mode == 0) {
                u = q[c];
            } else {
                const float x32 = q[c] * scale_f;                       // float32 * python float -> float32
                double fz = gain * (double)x32 + c0 + s2;               // gain*x + (3/8)gain^2 + sigma^2 (- gain*0)
                fz = fz > 0.0 ? fz : 0.0;
                double v = two_over_gain * sqrt_newton(fz);
                if (lut_n > 0) v -= (lut_flags & LUT_BIASLUT) ? biaslut_eval(L, lut_n, fmaxf(x32, 0.0f), gain, sigma)
                                                              : lut_eval(L, lut_n, fmaxf(x32, 0.0f));
                u = (float)((v - lo) * inv_span);                       // (v - lo) / (hi - lo) to one float64 ulp, then ONE rounding
            }
            u = fminf(fmaxf(u, 0.0f), 1.0f);
            o[c] = u;
            vmax = fmaxf(vmax, u);
        }
        *(f32x4*)(out_b + p * 4) = o;
    }
    }
    flush_max(cur_b);
}

// K1 for the frames of the device chain (constants from the parameter block, the table the chain prepared: <= 3 evenly spaced runs).
// The same function as pack_vst_norm_kernel with the affine tail folded into the coefficients,
//     u = (2/K sqrt(fz) - (a + b x) - lo) / (hi - lo)  =  A sqrt(fz) - (a' + b' x),   A = 2 / (K (hi - lo)), a' = (a + lo) / (hi - lo), b' = b / (hi - lo),
// and the runs held in registers: 11 float64 operations per element become 6, the interval lookup reads LDS once, and the kernel fits
// 64 registers (eight waves per SIMD: the loop is a chain of dependent conversions and float64 operations behind two loads).
// The float64 result moves by a few 1e-16 (another association of the same terms): after the one rounding to float32 a different value
// for about one element in 1e8.  A table that is not <= 3 runs is flagged (YOND_PRM_FLAG_LUT_CAPACITY: the host path takes the frame).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))
void pack_vst_chain_kernel(const float* __restrict__ bayer, int H, int W, float* __restrict__ out, int pad_l, int pad_t, int Hp, int Wp,
                           float scale_f, unsigned int* __restrict__ img_max, double* __restrict__ prm, const void* __restrict__ lut_ws,
                           int lut_cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lut_raw[];
    __shared__ LutLds L;
    __shared__ float s_red[4];
    const int fl = (int)prm[YOND_PRM_FLAGS];
    if (fl & (YOND_PRM_FLAG_BAD_ESTIMATE | YOND_PRM_FLAG_LUT_CAPACITY | YOND_PRM_FLAG_NO_FLAT_AREA)) return;
    const double gain = prm[YOND_PRM_GAIN], sigma = prm[YOND_PRM_SIGMA], lo = prm[YOND_PRM_LO], hi = prm[YOND_PRM_HI];
    int lut_n = (int)prm[YOND_PRM_LUT_N];
    if (lut_n > lut_cap) return;                              // (cannot happen: the block's producer checked the same capacity)
    if (threadIdx.x == 0) { L.ab = (double2*)lut_raw; L.x = nullptr; }
    __syncthreads();
    lut_load(L, lut_ws, lut_n);
    // the runs a query can land in: a repeated knot (get_bias concatenates its runs: 50 and 500 appear twice) opens a zero-width run that
    // lut_eval's count of run starts <= x always steps over -- drop those, keep (start, 1 / step, first knot, last knot) of the others
    float rx[3] = {0.f, INFINITY, INFINITY}, rv[3] = {0.f, 0.f, 0.f};
    int ra[3] = {0, 0, 0}, rb[3] = {0, 0, 0};
    int kept = 0;
    const int ns = L.nseg;
    for (int i = 0; i < ns; ++i) {
        if (i + 1 < ns && L.seg_x[i + 1] == L.seg_x[i]) continue;
        // (constant indices after unrolling: a run-time index would put the four little arrays into scratch memory -- 16 bytes per lane)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (kept == j) { rx[j] = L.seg_x[i]; rv[j] = L.seg_inv[i]; ra[j] = L.seg_i[i]; rb[j] = L.seg_i[i + 1]; }
        ++kept;
    }
    if (lut_n < 2 || ns < 1 || kept > 3) {
        if (blockIdx.x == 0 && threadIdx.x == 0) prm[YOND_PRM_FLAGS] = (double)(fl | YOND_PRM_FLAG_LUT_CAPACITY);
        return;
    }
    const int h = H / 2, w = W / 2;
    const double inv_span = 1.0 / (hi - lo);
    const double fA = 2.0 / gain * inv_span, fC = 0.375 * gain * gain + sigma * sigma;
    for (int i = threadIdx.x; i < lut_n; i += 256) {
        const double2 c = L.ab[i];
        L.ab[i] = make_double2((c.x + lo) * inv_span, c.y * inv_span);
    }
    // the kept runs in LDS, one 16-byte entry each: {start, 1 / step, first knot, last knot}; a run that is not there starts at +inf
    __shared__ f32x4 s_run[3];
    if (threadIdx.x == 0) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            f32x4 e;
            e[0] = rx[j]; e[1] = rv[j]; e[2] = __int_as_float(ra[j]); e[3] = __int_as_float(rb[j]);
            s_run[j] = e;
        }
    }
    const float fx1 = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(rx[1])));
    const float fx2 = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(rx[2])));
    const double2* tab = (const double2*)lut_raw;
    __syncthreads();
    const int nm1 = lut_n - 1;
    float vmax = 0.0f;
    const int per = (Hp + (int)gridDim.x - 1) / (int)gridDim.x;
    const int row_lo = blockIdx.x * per, row_hi = row_lo + per < Hp ? row_lo + per : Hp;
    for (int yp = row_lo; yp < row_hi; ++yp) {
        int sy = yp - pad_t;
        if ((unsigned)sy >= (unsigned)h) sy = reflect101(sy, h);                // (uniform)
        const float* row0 = bayer + (size_t)(2 * sy) * W;
        float* orow = out + (size_t)yp * Wp * 4;
        // two pixels per trip: four loads in flight ahead of ~90 dependent operations
        for (int xp = threadIdx.x; xp < Wp; xp += 512) {
            const int xp2 = xp + 256;
            const bool two = xp2 < Wp;
            int sx = xp - pad_l, sx2 = (two ? xp2 : xp) - pad_l;
            if ((unsigned)sx >= (unsigned)w) sx = reflect101(sx, w);
            if ((unsigned)sx2 >= (unsigned)w) sx2 = reflect101(sx2, w);
            const f32x2 r0 = *(const f32x2*)(row0 + 2 * sx), r1 = *(const f32x2*)(row0 + W + 2 * sx);
            const f32x2 r2 = *(const f32x2*)(row0 + 2 * sx2), r3 = *(const f32x2*)(row0 + W + 2 * sx2);
            const float q[8] = {r0[0], r0[1], r1[0], r1[1], r2[0], r2[1], r3[0], r3[1]};
            float o[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float x32 = q[c] * scale_f;                           // float32 * python float -> float32
                const double xd = (double)x32;
                const double fz = fmax(fma(gain, xd, fC), 0.0);
                // sqrt_newton without its branches: an argument below 1e-30 is taken as 1e-30 (the root, < 1e-15, times A ~ 1e-2 is far
                // below the float32 result's ulp either way)
                const float af = fmaxf((float)fz, 1e-30f);
                const float s0f = __builtin_amdgcn_sqrtf(af);
                const double s0 = (double)s0f;
                const double sq = fma(fma(-s0, s0, fz), (double)(0.5f * __builtin_amdgcn_rcpf(s0f)), s0);
                // the interval, as lut_eval finds it (searchsorted 'left' on evenly spaced runs)
                const float xq = fmaxf(x32, 0.0f);
                const f32x4 run = s_run[(xq >= fx1 ? 1 : 0) + (xq >= fx2 ? 1 : 0)];
                const float t = (xq - run[0]) * run[1];
                const int off = (int)fminf(fmaxf(ceilf(t), 0.0f), 16777216.0f);
                const int g = min(__float_as_int(run[2]) + off, __float_as_int(run[3]));
                const double2 cf = tab[max(1, min(g, nm1))];
                const double tl = fma(cf.y, fmax(xd, 0.0), cf.x);
                float u = (float)fma(fA, sq, -tl);
                u = fminf(fmaxf(u, 0.0f), 1.0f);
                o[c] = u;
                vmax = fmaxf(vmax, u);
            }
            *(f32x4*)(orow + (size_t)xp * 4) = f32x4{o[0], o[1], o[2], o[3]};
            if (two) *(f32x4*)(orow + (size_t)xp2 * 4) = f32x4{o[4], o[5], o[6], o[7]};
        }
    }
    if (img_max) {
        const float m0 = wave_max(vmax);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m0;
        __syncthreads();
        if (threadIdx.x == 0) atomicMax(img_max, __float_as_uint(fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]))));
    }
}

static int launch_pack_vst(const float* bayer, int H, int W, float* out, int pad_l, int pad_r, int pad_t, int pad_b, int mode,
                           double scale, double gain, double sigma, double lo, double hi, const double* lut_x, const void* lut_y,
                           int lut_n, int lut_flags, float* img_max, void* stream, int B = 1) {
    if (!bayer || !out || H < 2 || W < 2 || (H & 1) || (W & 1) || B < 1) return YOND_EINVAL;
    if (pad_l < 0 || pad_r < 0 || pad_t < 0 || pad_b < 0) return YOND_EINVAL;
    if (mode != 0 && mode != 1) return YOND_EINVAL;
    if (lut_n < 0 || lut_n > LUT_MAX || lut_n == 1 || (lut_n > 0 && (!lut_x || !lut_y))) return YOND_EINVAL;
    if (mode == 1 && !(gain > 0.0) ) return YOND_EINVAL;
    if (mode == 1 && !(hi > lo)) return YOND_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int Hp = H / 2 + pad_t + pad_b, Wp = W / 2 + pad_l + pad_r;
    if (img_max) {
        hipError_t e = hipMemsetAsync(img_max, 0, (size_t)B * sizeof(float), st);
        if (e != hipSuccess) return (int)e;
    }
    size_t nb = (size_t)Hp * B;
    if (nb > 1536) nb = 1536;                       // every workgroup prepares the LUT once: keep them few and long-lived
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)pack_vst_norm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LUT_MAX * LUT_BYTES_PER_KNOT);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    const int use_lut = mode == 1 ? lut_n : 0;
    hipLaunchKernelGGL(pack_vst_norm_kernel, dim3((unsigned)nb), dim3(256), (size_t)use_lut * LUT_BYTES_PER_KNOT, st, bayer, H, W, out,
                       pad_l, pad_t, Hp, Wp, mode, (float)scale, gain, sigma, lo, hi, lut_x, lut_y, use_lut, lut_flags, (unsigned int*)img_max,
                       (const double*)nullptr, (const void*)nullptr, 0, B);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

// ---- the frame's constants from a parameter block, the LUT as a prepared table (frame_chain.hip) ----
__global__ __launch_bounds__(256) void lut_table_kernel(const double* __restrict__ lut_x, const float* __restrict__ lut_y, int n,
                                                        const double* __restrict__ prm, void* __restrict__ ws, double* __restrict__ prm_rw) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lut_raw[];
    __shared__ LutLds L;
    LutHeader* hd = (LutHeader*)ws;
    if (prm) {
        const int fl = (int)prm[YOND_PRM_FLAGS];
        n = (fl & (YOND_PRM_FLAG_BAD_ESTIMATE | YOND_PRM_FLAG_LUT_CAPACITY)) ? 0 : (int)prm[YOND_PRM_LUT_N];
    }
    if (n < 2) { if (threadIdx.x == 0) { hd->n = 0; hd->nseg = 0; hd->nbreak = 0; } return; }
    if (threadIdx.x == 0) {
        L.ab = (double2*)lut_raw;
        L.x = (double*)(L.ab + n);
    }
    __syncthreads();
    lut_prepare(L, lut_x, lut_y, n, 0);
    lut_table_store(L, ws, n);
    if (threadIdx.x == 0) {
        // K1 keeps only the coefficients in LDS and finds the interval from the run table: a grid that is not a few evenly
        // spaced runs (never the case for yond_frame_params_f64's grids) goes back to the host path
        if (prm_rw && L.nseg == 0) prm_rw[YOND_PRM_FLAGS] = (double)((int)prm_rw[YOND_PRM_FLAGS] | YOND_PRM_FLAG_LUT_CAPACITY);
    }
}

extern "C" size_t yond_lut_ws_bytes(int lut_cap) {
    (void)lut_cap;                                   // (the table is laid out for the kernels' maximum, LUT_MAX knots)
    return sizeof(LutHeader) + (size_t)LUT_MAX * LUT_BYTES_PER_KNOT;
}

static int lut_smem_attr(const void* kern) {
    return (int)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, LUT_MAX * LUT_BYTES_PER_KNOT);
}

extern "C" int yond_lut_table_f64(const double* lut_x, const float* lut_y, int n, const double* prm, void* lut_ws, void* stream) {
    if (!lut_x || !lut_y || !lut_ws || (n < 0 && !prm) || n > LUT_MAX || n == 1) return YOND_EINVAL;
    static bool attr = false;
    if (!attr) {
        if (int e = lut_smem_attr((const void*)lut_table_kernel)) return e;
        attr = true;
    }
    hipLaunchKernelGGL(lut_table_kernel, dim3(1), dim3(256), (size_t)LUT_MAX * LUT_BYTES_PER_KNOT, (hipStream_t)stream, lut_x, lut_y,
                       n < 0 ? 0 : n, n < 0 ? prm : (const double*)nullptr, lut_ws, n < 0 ? (double*)prm : (double*)nullptr);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

extern "C" int yond_pack_vst_norm_dev_f32(const float* bayer, int H, int W, float* out, int pad_l, int pad_r, int pad_t, int pad_b,
                                          double scale, const double* prm, const void* lut_ws, int lut_cap, float* img_max, void* stream) {
    if (!bayer || !out || !prm || H < 2 || W < 2 || (H & 1) || (W & 1)) return YOND_EINVAL;
    if (lut_ws && (lut_cap < 2 || lut_cap > LUT_MAX)) return YOND_EINVAL;
    if (pad_l < 0 || pad_r < 0 || pad_t < 0 || pad_b < 0 || !(scale > 0.0)) return YOND_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int Hp = H / 2 + pad_t + pad_b, Wp = W / 2 + pad_l + pad_r;
    if (img_max) {
        hipError_t e = hipMemsetAsync(img_max, 0, sizeof(float), st);
        if (e != hipSuccess) return (int)e;
    }
    static bool attr = false;
    if (!attr) {
        if (int e = lut_smem_attr((const void*)pack_vst_norm_kernel)) return e;
        attr = true;
    }
    const unsigned nb = Hp < 1536 ? (unsigned)Hp : 1536u;       // (the table is copied, not derived, per workgroup)
    // (LDS: the coefficients of the caller's knot capacity: 1536 knots = 24 KB, six workgroups per CU)
    hipLaunchKernelGGL(pack_vst_norm_kernel, dim3(nb), dim3(256), lut_ws ? (size_t)lut_cap * sizeof(double2) : 0, st, bayer, H, W, out,
                       pad_l, pad_t, Hp, Wp, 1, (float)scale, 1.0, 0.0, 0.0, 1.0, (const double*)nullptr, (const void*)nullptr, 0, 0,
                       (unsigned int*)img_max, prm, lut_ws, lut_cap, 1);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

// ... and B frames that share the parameter block and the table (img_max [B])
extern "C" int yond_pack_vst_norm_batch_dev_f32(const float* bayer, int B, int H, int W, float* out, int pad_l, int pad_r, int pad_t, int pad_b,
                                          double scale, const double* prm, const void* lut_ws, int lut_cap, float* img_max, void* stream) {
    if (!bayer || !out || !prm || B < 1 || H < 2 || W < 2 || (H & 1) || (W & 1)) return YOND_EINVAL;
    if (lut_ws && (lut_cap < 2 || lut_cap > LUT_MAX)) return YOND_EINVAL;
    if (pad_l < 0 || pad_r < 0 || pad_t < 0 || pad_b < 0 || !(scale > 0.0)) return YOND_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int Hp = H / 2 + pad_t + pad_b, Wp = W / 2 + pad_l + pad_r;
    if (img_max) {
        hipError_t e = hipMemsetAsync(img_max, 0, (size_t)B * sizeof(float), st);
        if (e != hipSuccess) return (int)e;
    }
    static bool attr = false;
    if (!attr) {
        if (int e = lut_smem_attr((const void*)pack_vst_norm_kernel)) return e;
        attr = true;
    }
    const unsigned nb = (long long)Hp * B < 1536 ? (unsigned)(Hp * B) : 1536u;       // (the table is copied, not derived, per workgroup)
    // (LDS: the coefficients of the caller's knot capacity: 1536 knots = 24 KB, six workgroups per CU)
    hipLaunchKernelGGL(pack_vst_norm_kernel, dim3(nb), dim3(256), lut_ws ? (size_t)lut_cap * sizeof(double2) : 0, st, bayer, H, W, out,
                       pad_l, pad_t, Hp, Wp, 1, (float)scale, 1.0, 0.0, 0.0, 1.0, (const double*)nullptr, (const void*)nullptr, 0, 0,
                       (unsigned int*)img_max, prm, lut_ws, lut_cap, B);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

// The chain's own K1 (pack_vst_chain_kernel): as yond_pack_vst_norm_dev_f32 for a table yond_frame_chain_f64 prepared; prm is also
// WRITTEN (a table of another shape is flagged, nothing is computed).
extern "C" int yond_pack_vst_norm_chain_f32(const float* bayer, int H, int W, float* out, int pad_l, int pad_r, int pad_t, int pad_b,
                                            double scale, double* prm, const void* lut_ws, int lut_cap, float* img_max, void* stream) {
    if (!bayer || !out || !prm || !lut_ws || H < 2 || W < 2 || (H & 1) || (W & 1)) return YOND_EINVAL;
    if (lut_cap < 2 || lut_cap > LUT_MAX) return YOND_EINVAL;
    if (pad_l < 0 || pad_r < 0 || pad_t < 0 || pad_b < 0 || !(scale > 0.0)) return YOND_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int Hp = H / 2 + pad_t + pad_b, Wp = W / 2 + pad_l + pad_r;
    if (img_max) {
        hipError_t e = hipMemsetAsync(img_max, 0, sizeof(float), st);
        if (e != hipSuccess) return (int)e;
    }
    static bool attr = false;
    if (!attr) {
        if (int e = lut_smem_attr((const void*)pack_vst_chain_kernel)) return e;
        attr = true;
    }
    // three workgroups per CU, two rows each at the cfg-2 size: every workgroup copies and rescales the table once (measured, same box:
    // 256 workgroups 38.0 us, 512 30.2, 768 29.5, 1024 30.3, 1536 33.1)
    const long cap = yond_exp_long("YOND_K1_WGS", 768);
    const unsigned nb = Hp < (int)cap ? (unsigned)Hp : (unsigned)cap;
    hipLaunchKernelGGL(pack_vst_chain_kernel, dim3(nb), dim3(256), (size_t)lut_cap * sizeof(double2), st, bayer, H, W, out, pad_l, pad_t, Hp, Wp,
                       (float)scale, (unsigned int*)img_max, prm, lut_ws, lut_cap);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

extern "C" int yond_pack_vst_norm_f32(const float* bayer, int H, int W, float* out, int pad_l, int pad_r, int pad_t,
                                      int pad_b, int mode, double scale, double gain, double sigma, double lo,
                                      double hi, const double* lut_x, const float* lut_y, int lut_n, float* img_max,
                                      void* stream) {
    return launch_pack_vst(bayer, H, W, out, pad_l, pad_r, pad_t, pad_b, mode, scale, gain, sigma, lo, hi, lut_x, lut_y, lut_n, 0,
                           img_max, stream);
}

extern "C" int yond_pack_vst_norm_biaslut_f32(const float* bayer, int H, int W, float* out, int pad_l, int pad_r, int pad_t,
                                              int pad_b, double scale, double gain, double sigma, double lo, double hi,
                                              const double* lut_x, const double* lut_y, int lut_n, float* img_max, void* stream) {
    if (lut_n < 2) return YOND_EINVAL;
    return launch_pack_vst(bayer, H, W, out, pad_l, pad_r, pad_t, pad_b, 1, scale, gain, sigma, lo, hi, lut_x, lut_y, lut_n,
                           LUT_Y64 | LUT_BIASLUT, img_max, stream);
}

// B equally sized frames with shared constants and LUT in ONE launch (the 32 blocks of a SIDD image: 32 launches of ~15 us before,
// each workgroup of each preparing the LUT): bayer [B][H][W] -> out [B][Hp][Wp][4], img_max [B].  biaslut != 0: lut_y is float64
// and holds the merged row of the 2-D table (as yond_pack_vst_norm_biaslut_f32).
extern "C" int yond_pack_vst_norm_batch_f32(const float* bayer, int B, int H, int W, float* out, int pad_l, int pad_r, int pad_t, int pad_b,
                                            double scale, double gain, double sigma, double lo, double hi, const double* lut_x,
                                            const void* lut_y, int lut_n, int biaslut, float* img_max, void* stream) {
    if (biaslut && lut_n < 2) return YOND_EINVAL;
    return launch_pack_vst(bayer, H, W, out, pad_l, pad_r, pad_t, pad_b, 1, scale, gain, sigma, lo, hi, lut_x, lut_y, lut_n,
                           biaslut ? (LUT_Y64 | LUT_BIASLUT) : 0, img_max, stream, B);
}

// the LUT alone, per element (function seam: the interp1d object of get_bias / BiasLUT.get_lut called on an array)
__global__ __launch_bounds__(256) void bias_eval_kernel(const float* __restrict__ x, size_t n, const double* __restrict__ lut_x,
                                                        const void* __restrict__ lut_y, int lut_n, int lut_flags, double gain,
                                                        double sigma, double* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lut_raw[];
    __shared__ LutLds L;
    if (threadIdx.x == 0) {
        L.ab = (double2*)lut_raw;
        L.x = (double*)(L.ab + lut_n);
    }
    __syncthreads();
    lut_prepare(L, lut_x, lut_y, lut_n, lut_flags);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = (lut_flags & LUT_BIASLUT) ? biaslut_eval(L, lut_n, x[i], gain, sigma) : lut_eval(L, lut_n, x[i]);
}

extern "C" int yond_bias_eval_f32(const float* x, size_t n, const double* lut_x, const void* lut_y, int lut_n, int y_is_f64,
                                  int biaslut, double gain, double sigma, double* out, void* stream) {
    if (!x || !out || !lut_x || !lut_y || n == 0 || lut_n < 2 || lut_n > LUT_MAX) return YOND_EINVAL;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)bias_eval_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LUT_MAX * LUT_BYTES_PER_KNOT);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    size_t nb = (n + 255) / 256;
    if (nb > 512) nb = 512;
    hipLaunchKernelGGL(bias_eval_kernel, dim3((unsigned)nb), dim3(256), (size_t)lut_n * LUT_BYTES_PER_KNOT, (hipStream_t)stream, x, n,
                       lut_x, lut_y, lut_n, (y_is_f64 ? LUT_Y64 : 0) | (biaslut ? LUT_BIASLUT : 0), gain, sigma, out);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

__global__ __launch_bounds__(256) void denorm_ivst_unpack_kernel(const float* __restrict__ net_out, int Wp, int pad_t,
                                                                 int pad_l, int h, int w, float* __restrict__ bayer,
                                                                 int mode, double scale, double gain, double sigma,
                                                                 double lo, double hi, int clip01, const double* __restrict__ prm,
                                                                 int B, int Hp) {
    if (prm) {
        const int fl = (int)prm[YOND_PRM_FLAGS];
        if (fl & (YOND_PRM_FLAG_BAD_ESTIMATE | YOND_PRM_FLAG_LUT_CAPACITY | YOND_PRM_FLAG_NO_FLAT_AREA)) return;
        gain = prm[YOND_PRM_GAIN]; sigma = prm[YOND_PRM_SIGMA]; lo = prm[YOND_PRM_LO]; hi = prm[YOND_PRM_HI];
    }
    const double span = hi - lo;
    const double sg = sigma / gain;                 // inverse_VST: sigma = sigma / gain
    const double sg2 = sg * sg;
    const double r32 = sqrt(1.5);                   // (3/2)**0.5
    for (int row = blockIdx.x; row < B * h; row += gridDim.x) {
    const int bi = row / h, y = row - bi * h;                          // (B frames [Hp][Wp][4] -> [B][2h][2w], shared constants)
    const float* net_b = net_out + (size_t)bi * Hp * Wp * 4;
    float* bayer_b = bayer + (size_t)bi * (2 * h) * (2 * w);
    for (int x = threadIdx.x; x < w; x += 256) {
        const f32x4 v = *(const f32x4*)(net_b + ((size_t)(y + pad_t) * Wp + x + pad_l) * 4);
        float o[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float yc = fminf(fmaxf(v[c], 0.0f), 1.0f);
            if (mode == 0) { o[c] = yc; continue; }
            const double z = (double)yc * span + lo;
            double fz;
            if (mode == 2) {
                if (z > 0.0) {
                    const double iz = 1.0 / z;
                    fz = (z / 2) * (z / 2) + 0.25 * r32 * iz - 1.375 * (iz * iz) + 0.625 * r32 * (iz * iz * iz) - 0.125 - sg2;
                } else fz = 0.0;
            } else {
                fz = (z / 2) * (z / 2) - 0.375 - sg2;
            }
            fz = fz > 0.0 ? fz : 0.0;
            double r = fz * gain / scale;
            if (clip01) r = r < 0.0 ? 0.0 : (r > 1.0 ? 1.0 : r);
            o[c] = (float)r;
        }
        f32x2 r0 = {o[0], o[1]}, r1 = {o[2], o[3]};
        *(f32x2*)(bayer_b + (size_t)(2 * y) * (2 * w) + 2 * x) = r0;
        *(f32x2*)(bayer_b + (size_t)(2 * y + 1) * (2 * w) + 2 * x) = r1;
    }
    }
}

static int launch_ivst(const float* net_out, int B, int Hp, int Wp, int pad_t, int pad_l, int h, int w, float* bayer_out, int mode, double scale,
                       double gain, double sigma, double lo, double hi, int clip01, void* stream) {
    if (!net_out || !bayer_out || h <= 0 || w <= 0 || pad_t < 0 || pad_l < 0 || B < 1) return YOND_EINVAL;
    if (pad_t + h > Hp || pad_l + w > Wp) return YOND_EINVAL;
    if (mode < 0 || mode > 2) return YOND_EINVAL;
    if (mode != 0 && (!(gain > 0.0) || !(scale > 0.0))) return YOND_EINVAL;
    size_t nb = (size_t)h * B;
    if (nb > 256 * 16) nb = 256 * 16;
    hipLaunchKernelGGL(denorm_ivst_unpack_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, net_out, Wp, pad_t,
                       pad_l, h, w, bayer_out, mode, scale, gain, sigma, lo, hi, clip01, (const double*)nullptr, B, Hp);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

extern "C" int yond_denorm_ivst_unpack_f32(const float* net_out, int Hp, int Wp, int pad_t, int pad_l, int h, int w,
                                           float* bayer_out, int mode, double scale, double gain, double sigma,
                                           double lo, double hi, int clip01, void* stream) {
    return launch_ivst(net_out, 1, Hp, Wp, pad_t, pad_l, h, w, bayer_out, mode, scale, gain, sigma, lo, hi, clip01, stream);
}

// B equally sized frames with shared constants in one launch: net_out [B][Hp][Wp][4] -> bayer_out [B][2h][2w]
extern "C" int yond_denorm_ivst_unpack_batch_f32(const float* net_out, int B, int Hp, int Wp, int pad_t, int pad_l, int h, int w,
                                                 float* bayer_out, int mode, double scale, double gain, double sigma,
                                                 double lo, double hi, int clip01, void* stream) {
    return launch_ivst(net_out, B, Hp, Wp, pad_t, pad_l, h, w, bayer_out, mode, scale, gain, sigma, lo, hi, clip01, stream);
}

extern "C" int yond_denorm_ivst_unpack_dev_f32(const float* net_out, int Hp, int Wp, int pad_t, int pad_l, int h, int w,
                                               float* bayer_out, int mode, double scale, const double* prm, int clip01, void* stream) {
    if (!net_out || !bayer_out || !prm || h <= 0 || w <= 0 || pad_t < 0 || pad_l < 0) return YOND_EINVAL;
    if (pad_t + h > Hp || pad_l + w > Wp || (mode != 1 && mode != 2) || !(scale > 0.0)) return YOND_EINVAL;
    size_t nb = (size_t)h;
    if (nb > 256 * 16) nb = 256 * 16;
    hipLaunchKernelGGL(denorm_ivst_unpack_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, net_out, Wp, pad_t,
                       pad_l, h, w, bayer_out, mode, scale, 1.0, 0.0, 0.0, 1.0, clip01, prm, 1, Hp);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

// ... and B frames that share the parameter block (the 32 blocks of a SIDD image on the device chain)
extern "C" int yond_denorm_ivst_unpack_batch_dev_f32(const float* net_out, int B, int Hp, int Wp, int pad_t, int pad_l, int h, int w,
                                                     float* bayer_out, int mode, double scale, const double* prm, int clip01, void* stream) {
    if (!net_out || !bayer_out || !prm || B < 1 || h <= 0 || w <= 0 || pad_t < 0 || pad_l < 0) return YOND_EINVAL;
    if (pad_t + h > Hp || pad_l + w > Wp || (mode != 1 && mode != 2) || !(scale > 0.0)) return YOND_EINVAL;
    size_t nb = (size_t)h * B;
    if (nb > 256 * 16) nb = 256 * 16;
    hipLaunchKernelGGL(denorm_ivst_unpack_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, net_out, Wp, pad_t,
                       pad_l, h, w, bayer_out, mode, scale, 1.0, 0.0, 0.0, 1.0, clip01, prm, B, Hp);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

// ---- bit-exact pack / unpack (utils/isp_ops.py:57-63) ----
__global__ __launch_bounds__(256) void bayer2rggb_kernel(const float* __restrict__ bayer, int W, int w, size_t total,
                                                         float* __restrict__ rggb) {
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < total; p += (size_t)gridDim.x * 256) {
        const size_t y = p / w, x = p % w;
        const f32x2 r0 = *(const f32x2*)(bayer + (2 * y) * W + 2 * x);
        const f32x2 r1 = *(const f32x2*)(bayer + (2 * y + 1) * W + 2 * x);
        f32x4 o = {r0[0], r0[1], r1[0], r1[1]};
        *(f32x4*)(rggb + p * 4) = o;
    }
}

__global__ __launch_bounds__(256) void rggb2bayer_kernel(const float* __restrict__ rggb, int w, size_t total,
                                                         float* __restrict__ bayer) {
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < total; p += (size_t)gridDim.x * 256) {
        const size_t y = p / w, x = p % w;
        const f32x4 v = *(const f32x4*)(rggb + p * 4);
        f32x2 r0 = {v[0], v[1]}, r1 = {v[2], v[3]};
        *(f32x2*)(bayer + (2 * y) * (2 * (size_t)w) + 2 * x) = r0;
        *(f32x2*)(bayer + (2 * y + 1) * (2 * (size_t)w) + 2 * x) = r1;
    }
}

static unsigned stream_grid(size_t total) {
    size_t nb = (total + 255) / 256;
    if (nb > 256 * 16) nb = 256 * 16;
    if (nb < 1) nb = 1;
    return (unsigned)nb;
}

extern "C" int yond_bayer2rggb_f32(const float* bayer, int H, int W, float* rggb, void* stream) {
    if (!bayer || !rggb || H < 2 || W < 2 || (H & 1) || (W & 1)) return YOND_EINVAL;
    const size_t total = (size_t)(H / 2) * (W / 2);
    hipLaunchKernelGGL(bayer2rggb_kernel, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, bayer, W, W / 2, total, rggb);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

extern "C" int yond_rggb2bayer_f32(const float* rggb, int h, int w, float* bayer, void* stream) {
    if (!bayer || !rggb || h < 1 || w < 1) return YOND_EINVAL;
    const size_t total = (size_t)h * w;
    hipLaunchKernelGGL(rggb2bayer_kernel, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, rggb, w, total, bayer);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

// ---- NCHW(4) <-> NHWC4 for the archs plugin surface ----
__global__ __launch_bounds__(256) void nchw4_to_nhwc4_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                             size_t hw, size_t total) {
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < total; p += (size_t)gridDim.x * 256) {
        const size_t n = p / hw, i = p % hw;
        const float* s = src + n * 4 * hw + i;
        f32x4 o = {s[0], s[hw], s[2 * hw], s[3 * hw]};
        *(f32x4*)(dst + p * 4) = o;
    }
}

__global__ __launch_bounds__(256) void nhwc4_to_nchw4_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                             size_t hw, size_t total) {
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < total; p += (size_t)gridDim.x * 256) {
        const size_t n = p / hw, i = p % hw;
        const f32x4 v = *(const f32x4*)(src + p * 4);
        float* d = dst + n * 4 * hw + i;
        d[0] = v[0]; d[hw] = v[1]; d[2 * hw] = v[2]; d[3 * hw] = v[3];
    }
}

extern "C" int yond_nchw4_to_nhwc4_f32(const float* src, float* dst, int N, int H, int W, void* stream) {
    if (!src || !dst || N <= 0 || H <= 0 || W <= 0) return YOND_EINVAL;
    const size_t hw = (size_t)H * W, total = hw * N;
    hipLaunchKernelGGL(nchw4_to_nhwc4_kernel, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, src, dst, hw, total);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

extern "C" int yond_nhwc4_to_nchw4_f32(const float* src, float* dst, int N, int H, int W, void* stream) {
    if (!src || !dst || N <= 0 || H <= 0 || W <= 0) return YOND_EINVAL;
    const size_t hw = (size_t)H * W, total = hw * N;
    hipLaunchKernelGGL(nhwc4_to_nchw4_kernel, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, src, dst, hw, total);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

// ---- per-image maximum, two deterministic stages (archs/modules.py:18-19) ----
__global__ __launch_bounds__(256) void image_max_stage1(const float* __restrict__ x, size_t elems, float* __restrict__ partial) {
    __shared__ float s_red[4];
    const int n = blockIdx.y;
    const float* p = x + (size_t)n * elems;
    float m = -INFINITY;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < elems; i += (size_t)gridDim.x * 256) m = fmaxf(m, p[i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)n * gridDim.x + blockIdx.x] = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
}

__global__ __launch_bounds__(256) void image_max_stage2(const float* __restrict__ partial, int nb, float* __restrict__ out) {
    __shared__ float s_red[4];
    const int n = blockIdx.x;
    float m = -INFINITY;
    for (int i = threadIdx.x; i < nb; i += 256) m = fmaxf(m, partial[(size_t)n * nb + i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[n] = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
}

extern "C" int yond_image_max_f32(const float* x, int N, size_t elems, float* partial, float* out, void* stream) {
    if (!x || !partial || !out || N <= 0 || N > 65535 || elems == 0) return YOND_EINVAL;
    size_t nb = (elems + 256 * 16 - 1) / (256 * 16);
    if (nb > 256) nb = 256;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(image_max_stage1, dim3((unsigned)nb, N), dim3(256), 0, (hipStream_t)stream, x, elems, partial);
    YOND_LAUNCH_CHECK();
    hipLaunchKernelGGL(image_max_stage2, dim3(N), dim3(256), 0, (hipStream_t)stream, partial, (int)nb, out);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

// ---- stand-alone elementwise VST / inverse VST for the function seam (utils/isp_algos.py:5-33) ----
// The hot path uses the fused K1 / K4; these exist so that `VST(x, sigma, mu, gain)` / `inverse_VST(z, ...)`
// keep working on arbitrary arrays.  float32 in -> float64 out (NumPy promotes to float64 with np.float64
// parameters), float64 in -> float64 out for the inverse.
__global__ __launch_bounds__(256) void vst_elem_kernel(const float* __restrict__ x, size_t n, double sigma, double mu,
                                                       double gain, double* __restrict__ out) {
    const double c0 = 0.375 * gain * gain, s2 = sigma * sigma, gm = gain * mu, tg = 2.0 / gain;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        double fz = gain * (double)x[i] + c0 + s2 - gm;
        fz = fz > 0.0 ? fz : 0.0;
        out[i] = tg * sqrt(fz);
    }
}

__global__ __launch_bounds__(256) void ivst_elem_kernel(const double* __restrict__ z, size_t n, double sigma, double gain,
                                                        int exact, double* __restrict__ out) {
    const double sg = sigma / gain, sg2 = sg * sg, r32 = sqrt(1.5);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double v = z[i];
        double fz;
        if (exact) {
            if (v > 0.0) {
                const double iz = 1.0 / v;
                fz = (v / 2) * (v / 2) + 0.25 * r32 * iz - 1.375 * (iz * iz) + 0.625 * r32 * (iz * iz * iz) - 0.125 - sg2;
            } else fz = 0.0;
        } else {
            fz = (v / 2) * (v / 2) - 0.375 - sg2;
        }
        fz = fz > 0.0 ? fz : 0.0;
        out[i] = fz * gain;
    }
}

extern "C" int yond_vst_elem_f32(const float* x, size_t n, double sigma, double mu, double gain, double* out, void* stream) {
    if (!x || !out || n == 0 || !(gain > 0.0)) return YOND_EINVAL;
    hipLaunchKernelGGL(vst_elem_kernel, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream, x, n, sigma, mu, gain, out);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

extern "C" int yond_ivst_elem_f64(const double* z, size_t n, double sigma, double gain, int exact, double* out, void* stream) {
    if (!z || !out || n == 0 || !(gain > 0.0)) return YOND_EINVAL;
    hipLaunchKernelGGL(ivst_elem_kernel, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream, z, n, sigma, gain, exact, out);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

extern "C" int yond_abi_version(void) { return YOND_ABI_VERSION; }

// ---- rot90 of a stack of frames (np.rot90(x, k, axes=(-2, -1)): utils/sidd_utils.py:198-213 rot_bayer), bit exact ----
// dst[n][i][j] (H' x W' = W x H for odd k): k=1: src[j][W-1-i]; k=2: src[H-1-i][W-1-j]; k=3: src[H-1-j][i]
__global__ __launch_bounds__(256) void rot90_kernel(const float* __restrict__ src, int H, int W, int k, float* __restrict__ dst,
                                                    size_t per, size_t total) {
    const int Ho = (k & 1) ? W : H, Wo = (k & 1) ? H : W;
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < total; p += (size_t)gridDim.x * 256) {
        const size_t n = p / per;
        const int r = (int)(p % per), i = r / Wo, j = r % Wo;
        int sy, sx;
        if (k == 0) { sy = i; sx = j; }
        else if (k == 1) { sy = j; sx = W - 1 - i; }
        else if (k == 2) { sy = H - 1 - i; sx = W - 1 - j; }
        else { sy = H - 1 - j; sx = i; }
        (void)Ho;
        dst[p] = src[n * per + (size_t)sy * W + sx];
    }
}

extern "C" int yond_rot90_f32(const float* src, int N, int H, int W, int k, float* dst, void* stream) {
    if (!src || !dst || N <= 0 || H <= 0 || W <= 0) return YOND_EINVAL;
    k = ((k % 4) + 4) % 4;
    const size_t per = (size_t)H * W, total = per * N;
    hipLaunchKernelGGL(rot90_kernel, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, src, H, W, k, dst, per, total);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}
