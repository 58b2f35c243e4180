// K1 / K4 and the small layout kernels around the denoiser: pure streaming, HBM bound.
//   K1  Bayer -> pack -> *scale -> bias LUT -> VST -> normalise -> reflect pad -> clamp -> NHWC4 (+ image max)
//   K4  NHWC4 -> clamp -> crop -> de-normalise -> inverse VST -> unpack -> /scale -> clip -> Bayer
// One thread per packed pixel: it reads/writes one 2x2 Bayer quad (two 8-byte accesses on adjacent rows,
// coalesced across the wave) and one 16-byte NHWC4 slot.  Arithmetic follows the dtype staging of the
// reference under NumPy 2: float32 x*scale, float64 for everything up to the single rounding to float32
// (YOND_SIDD.py:251-269, 292-299; utils/isp_algos.py:5-33).
#include <stdlib.h>
#include "common.h"

#define LUT_MAX 4096
#define LUT_MAXSEG 8
#define LUT_BYTES_PER_KNOT 24

// The bias LUT (utils/isp_algos.py:103-108, 128: interp1d over knots that are runs of evenly spaced values, step 0.1 /
// 1 / 10) is a continuous piecewise-linear function, evaluated per pixel as a + b * x with the interval's coefficients
//   b_i = (y_i - y_{i-1}) / (x_i - x_{i-1})   (the float32 difference of the float32 ordinates, as interp1d forms it),
//   a_i = y_{i-1} - b_i x_{i-1}                (float64)
// from a 16-byte LDS entry.  The interval index comes from the run's spacing with one multiply; because the function is
// continuous, landing in the neighbouring interval when x sits within rounding distance of a knot changes the value by
// < 1e-12, far below the float32 rounding of K1's output -- so no search / repair against the knots is needed (the first
// version spent most of K1's time there, in float64 sqrt and in the float64 divide).  Knots that are not <= 8 evenly
// spaced runs fall back to bisection.
struct LutLds {
    double* x;                   // [n]  knots (bisection fallback)
    double2* ab;                 // [n]  coefficients of the interval that ENDS at knot i (i >= 1)
    float seg_x[LUT_MAXSEG], seg_inv[LUT_MAXSEG];
    int seg_i[LUT_MAXSEG + 1];
    int nseg;                    // 0: bisection
    int nbreak;
};

// flags: bit 0 -- ordinates are float64 (the 2-D BiasLUT's merged row) instead of float32 (get_bias' interp1d knots);
//        bit 1 -- BiasLUT semantics beyond the last knot (utils/isp_algos.py:188-194, 226-230): the last ordinate up to one
//                 more interval, Foi's closed form (float32-rounded, as the reference stores it) further out
#define LUT_Y64 1
#define LUT_BIASLUT 2
__device__ __forceinline__ void lut_prepare(LutLds& L, const double* __restrict__ lut_x, const void* __restrict__ lut_yv, int n,
                                            int flags) {
    const int tid = threadIdx.x;
    const float* lut_y = (const float*)lut_yv;
    const double* lut_y64 = (const double*)lut_yv;
    for (int i = tid; i < n; i += 256) L.x[i] = lut_x[i];
    if (tid == 0) { L.nbreak = 0; L.nseg = 0; }
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        if (i >= 1) {
            double dy, y0;
            if (flags & LUT_Y64) { dy = lut_y64[i] - lut_y64[i - 1]; y0 = lut_y64[i - 1]; }
            else { dy = (double)(lut_y[i] - lut_y[i - 1]); y0 = (double)lut_y[i - 1]; }   // float32 difference, as interp1d forms it
            double dx = L.x[i] - L.x[i - 1];
            if (dx == 0.0 && i >= 2) {
                // a repeated knot (get_bias concatenates its runs, so 50 and 500 appear twice): searchsorted('left') never
                // selects the empty interval -- a query equal to the knot belongs to the interval that ENDS at its first copy
                dx = L.x[i - 1] - L.x[i - 2];
                if (flags & LUT_Y64) { dy = lut_y64[i - 1] - lut_y64[i - 2]; y0 = lut_y64[i - 2]; }
                else { dy = (double)(lut_y[i - 1] - lut_y[i - 2]); y0 = (double)lut_y[i - 2]; }
                const double b = dy / dx;
                L.ab[i] = make_double2(y0 - b * L.x[i - 2], b);
            } else {
                const double b = dy / dx;
                L.ab[i] = make_double2(y0 - b * L.x[i - 1], b);
            }
        }
        if (i >= 1 && i + 1 < n) {
            const double d0 = L.x[i] - L.x[i - 1], d1 = L.x[i + 1] - L.x[i];
            if (fabs(d1 - d0) > 1e-3 * fabs(d0)) {                 // the spacing changes at knot i
                const int slot = atomicAdd(&L.nbreak, 1);
                if (slot < LUT_MAXSEG - 1) L.seg_i[slot + 1] = i;
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        const int nb = L.nbreak;
        if (nb <= LUT_MAXSEG - 1 && n >= 2) {
            L.seg_i[0] = 0;
            for (int a = 2; a <= nb; ++a) {                        // insertion sort of <= 7 break indices
                const int v = L.seg_i[a];
                int j = a - 1;
                while (j >= 1 && L.seg_i[j] > v) { L.seg_i[j + 1] = L.seg_i[j]; --j; }
                L.seg_i[j + 1] = v;
            }
            L.seg_i[nb + 1] = n - 1;
            bool ok = true;
            for (int sgm = 0; sgm <= nb; ++sgm) {
                const int i0 = L.seg_i[sgm];
                L.seg_x[sgm] = (float)L.x[i0];
                L.seg_inv[sgm] = (float)(1.0 / (L.x[i0 + 1] - L.x[i0]));
                ok = ok && ((double)L.seg_x[sgm] == L.x[i0]);      // run starts must be float32 values (0, 50, 500 are)
            }
            L.nseg = ok ? nb + 1 : 0;
        }
    }
    __syncthreads();
    // The index guess of lut_eval trusts that every run is EVENLY spaced.  The break detection above compares neighbouring
    // intervals with a relative tolerance, so a slowly drifting grid (a log grid of ratio < 1.001) would pass as one run:
    // check every knot against its run's ideal position x[i0] + (i - i0) * step and fall back to the bisection (exact for
    // any spacing) if one is off by more than 1e-3 of a step.  (float64 np.linspace knots are exact to ~1e-13 of a step;
    // the float32 ones NumPy 2 produces for a float32 maximum -- 500 ... ub in steps of ~9.8 -- to 6e-6 of a step; a query
    // that close to a knot may be evaluated on the neighbouring interval, which differs from the right one by the change of
    // slope times that distance: < 1e-9 on these tables.  A drifting grid is off by whole steps after a few hundred knots.)
    const int nseg = L.nseg;
    if (nseg > 0) {
        bool bad = false;
        for (int i = tid; i < n; i += 256) {
            int sgm = 0;
            for (int a = 1; a < nseg; ++a) sgm += (i > L.seg_i[a]) ? 1 : 0;
            const int i0 = L.seg_i[sgm];
            const double step = L.x[i0 + 1] - L.x[i0];
            // (a repeated knot -- get_bias' 50 and 500 -- opens a run: i0 is its second copy, the first copy closes the previous run)
            const double dev = fabs(L.x[i] - (L.x[i0] + (double)(i - i0) * step));
            bad = bad || !(dev <= 1e-3 * fabs(step));          // (the zero-width run between the two copies: dev = 0)
        }
        if (__syncthreads_or(bad ? 1 : 0)) {
            if (tid == 0) L.nseg = 0;
        }
        __syncthreads();
    }
}

// ---- the prepared table in global memory (yond_lut_table_f64): header, coefficients, knots ----
struct LutHeader {
    int n, nseg, nbreak, pad;
    float seg_x[LUT_MAXSEG], seg_inv[LUT_MAXSEG];
    int seg_i[LUT_MAXSEG + 1];
    int pad2[7];
};
static_assert(sizeof(LutHeader) == 144, "LutHeader layout");
__device__ __forceinline__ void lut_load(LutLds& L, const void* __restrict__ ws, int& n_out) {
    // plain copy of the image yond_lut_table_f64 stored: no arithmetic per workgroup
    const LutHeader* hd = (const LutHeader*)ws;
    const int n = hd->n;
    const double2* ab = (const double2*)((const char*)ws + sizeof(LutHeader));
    const double* x = (const double*)(ab + LUT_MAX);
    (void)x;                                          // (the knots themselves stay in global memory: the table's runs are even,
    for (int i = threadIdx.x; i < n; i += blockDim.x) L.ab[i] = ab[i];      //  yond_lut_table_f64 flags a grid that is not)
    if (threadIdx.x < LUT_MAXSEG) { L.seg_x[threadIdx.x] = hd->seg_x[threadIdx.x]; L.seg_inv[threadIdx.x] = hd->seg_inv[threadIdx.x]; }
    if (threadIdx.x <= LUT_MAXSEG) L.seg_i[threadIdx.x] = hd->seg_i[threadIdx.x];
    if (threadIdx.x == 0) { L.nseg = hd->nseg; L.nbreak = hd->nbreak; }
    n_out = n;
    __syncthreads();
}

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
    double2* ab = (double2*)((char*)ws + sizeof(LutHeader));
    double* x = (double*)(ab + LUT_MAX);
    for (int i = threadIdx.x; i < n; i += 256) { ab[i] = i >= 1 ? L.ab[i] : make_double2(0.0, 0.0); x[i] = L.x[i]; }
    if (threadIdx.x < LUT_MAXSEG) { hd->seg_x[threadIdx.x] = L.seg_x[threadIdx.x]; hd->seg_inv[threadIdx.x] = L.seg_inv[threadIdx.x]; }
    if (threadIdx.x <= LUT_MAXSEG) hd->seg_i[threadIdx.x] = L.seg_i[threadIdx.x];
    if (threadIdx.x == 0) {
        hd->n = n; hd->nseg = L.nseg; hd->nbreak = L.nbreak;
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
