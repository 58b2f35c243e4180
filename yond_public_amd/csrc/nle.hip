// K5-K7: the noise-level estimator's data passes (YOND_SIDD.py:13-115, utils/isp_algos.py:234-242, 345-365).
//   K5  box statistics: cv2.blur semantics (normalised k x k window, BORDER_REFLECT_101), window sums in
//       float64 (the float32 inputs make them exact), every intermediate rounded to float32 exactly where
//       NumPy / OpenCV round (no fused multiply-adds across those points).
//   K6  exact order statistics + np.percentile's linear interpolation: nle_select.hip.
//   K7  one pass over (lap, mean, var): occupancy of the 1/1000 mean bins per threshold bucket and the five
//       moment sums of the least-squares line per bucket.
// All of it is integer / streaming work bound by HBM and LDS, not MFMA.
#include "common.h"

// =====================================================================================================
// K5
// =====================================================================================================
#define BX_TH 32
#define BX_TW 64
#define BX_R 14
#define BX_RH (BX_TH + 2 * BX_R)
#define BX_RW (BX_TW + 2 * BX_R)

// Window sums of one quantity for this thread's 8 outputs: rows 8*(tid>>6) + i (i = 0..7), column tid&63.
// s_src: float [BX_RH][BX_RW] (tile with halo BX_R), s_h: double [BX_RH][BX_TW] scratch.
// Both passes slide the window (sum += entering - leaving) over runs of 8 outputs: 29+14 LDS reads per run
// instead of 8*29.  float32 data summed in float64: the window sums are exact (or differ from a direct sum
// by ~1e-16 relative for the squares), far below the float32 rounding that follows.
__device__ __forceinline__ void window_sums(const float* s_src, bool square, int k, double* s_h, double out[8]) {
    const int tid = threadIdx.x;
    const int rk = k / 2;
    const int off = BX_R - rk;                       // first tile row/col that the window of output 0 touches
    const int nrows = BX_TH + 2 * rk;
    __syncthreads();                                 // previous user of s_h is done
    for (int it = tid; it < nrows * (BX_TW / 8); it += 256) {
        const int row = it / (BX_TW / 8) + off, c0 = (it % (BX_TW / 8)) * 8;
        const float* p = s_src + row * BX_RW + c0 + off;
        double* o = s_h + row * BX_TW + c0;
        double s = 0.0;
        if (square) {
            for (int d = 0; d < k; ++d) { const float v = p[d]; s += (double)__fmul_rn(v, v); }
            o[0] = s;
#pragma unroll
            for (int j = 1; j < 8; ++j) {
                const float vn = p[j + k - 1], vo = p[j - 1];
                s += (double)__fmul_rn(vn, vn) - (double)__fmul_rn(vo, vo);
                o[j] = s;
            }
        } else {
            for (int d = 0; d < k; ++d) s += (double)p[d];
            o[0] = s;
#pragma unroll
            for (int j = 1; j < 8; ++j) { s += (double)p[j + k - 1] - (double)p[j - 1]; o[j] = s; }
        }
    }
    __syncthreads();
    const int col = tid & 63, rg = tid >> 6;
    const double* q = s_h + (rg * 8 + off) * BX_TW + col;
    double s = 0.0;
    for (int d = 0; d < k; ++d) s += q[d * BX_TW];
    out[0] = s;
#pragma unroll
    for (int i = 1; i < 8; ++i) { s += q[(i + k - 1) * BX_TW] - q[(i - 1) * BX_TW]; out[i] = s; }
}

__device__ __forceinline__ float blur_round(double s, int k) { return (float)(s * (1.0 / (double)(k * k))); }

// std = sqrt(max(B(x^2) - B(x)^2, 0)) in float32 steps (utils/isp_algos.py:236-241)
__device__ __forceinline__ float std_from(float b1, float b2) {
    const float d = __fsub_rn(b2, __fmul_rn(b1, b1));
    return __fsqrt_rn(fmaxf(d, 0.0f));
}

struct BoxSrc {
    const float* p;     // base pointer
    int bayer;          // 1: Bayer frame [2h][2w], plane = blockIdx.z ; 0: planar [4][h][w]
};

__device__ __forceinline__ void load_tile(float* s_src, BoxSrc src, int h, int w, int tile_w, int oy0, int ox0, int plane) {
    // reflect inside [0,h) x [bx0, bx0+bw): bw = tile_w (SIDD_256 re-tiling) or the whole width
    const int bw = tile_w > 0 ? tile_w : w;
    const int bx0 = tile_w > 0 ? (ox0 / tile_w) * tile_w : 0;
    const int dy = plane >> 1, dx = plane & 1;
    for (int it = threadIdx.x; it < BX_RH * BX_RW; it += 256) {
        const int ty = it / BX_RW, tx = it % BX_RW;
        const int gy = reflect101(oy0 - BX_R + ty, h);
        const int gx = bx0 + reflect101(ox0 - BX_R + tx - bx0, bw);
        float v;
        if (src.bayer) v = src.p[(size_t)(2 * gy + dy) * (2 * w) + 2 * gx + dx];
        else v = src.p[((size_t)plane * h + gy) * w + gx];
        s_src[it] = v;
    }
}

// mode 0: self stage 1 (mean, var, blur2 from the Bayer frame); 1: self stage 2 (lap from blur2);
// 2: collab (mean, var, lap from noisy + denoised Bayer frames)
template <int MODE>
__global__ __launch_bounds__(256) void box_stats_kernel(BoxSrc a, BoxSrc b, int h, int w, int k, int k2, int tile_w,
                                                        float* __restrict__ o0, float* __restrict__ o1, float* __restrict__ o2) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* s_h = (double*)smem_raw;                                   // [BX_RH][BX_TW]
    float* s_a = (float*)(smem_raw + sizeof(double) * BX_RH * BX_TW);  // [BX_RH][BX_RW]
    float* s_b = s_a + BX_RH * BX_RW;
    const int plane = blockIdx.z;
    const int ox0 = blockIdx.x * BX_TW, oy0 = blockIdx.y * BX_TH;
    load_tile(s_a, a, h, w, tile_w, oy0, ox0, plane);
    if (MODE == 2) load_tile(s_b, b, h, w, tile_w, oy0, ox0, plane);
    double q0[8], q1[8], q2[8], q3[8];
    if (MODE == 0) {
        window_sums(s_a, false, k, s_h, q0);
        window_sums(s_a, true, k, s_h, q1);
        window_sums(s_a, false, k2, s_h, q2);
    } else if (MODE == 1) {
        window_sums(s_a, false, k, s_h, q0);
        window_sums(s_a, true, k, s_h, q1);
    } else {
        window_sums(s_a, false, k, s_h, q0);
        window_sums(s_a, true, k, s_h, q1);
        window_sums(s_b, false, k, s_h, q2);
        window_sums(s_b, true, k, s_h, q3);
    }
    const int col = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int ox = ox0 + col;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int oy = oy0 + rg * 8 + i;
        if (oy >= h || ox >= w) continue;
        const size_t idx = ((size_t)plane * h + oy) * w + ox;
        if (MODE == 0) {
            const float m = blur_round(q0[i], k);
            const float sd = std_from(m, blur_round(q1[i], k));
            o0[idx] = m;
            o1[idx] = __fmul_rn(sd, sd);                              // var = lr_rggb_k**2 (YOND_SIDD.py:72)
            o2[idx] = blur_round(q2[i], k2);
        } else if (MODE == 1) {
            o0[idx] = std_from(blur_round(q0[i], k), blur_round(q1[i], k));
        } else {
            const float sl = std_from(blur_round(q0[i], k), blur_round(q1[i], k));
            const float mh = blur_round(q2[i], k);
            const float sh = std_from(mh, blur_round(q3[i], k));
            o0[idx] = mh;                                              // mean = blur(hr) (YOND_SIDD.py:97)
            o1[idx] = __fsub_rn(__fmul_rn(sl, sl), __fmul_rn(sh, sh));  // var = lr_k**2 - hr_k**2 (:96)
            o2[idx] = sh;                                              // img_lap = hr_k (:98)
        }
    }
}

static int box_args_ok(int h, int w, int k, int tile_w) {
    if (h < 1 || w < 1) return YOND_EINVAL;
    if (k < 1 || !(k & 1)) return YOND_EINVAL;
    if (k > 2 * BX_R + 1) return YOND_EUNSUPPORTED;
    if (tile_w < 0) return YOND_EINVAL;
    if (tile_w > 0 && (tile_w % BX_TW != 0 || w % tile_w != 0)) return YOND_EUNSUPPORTED;
    return YOND_OK;
}

template <int MODE>
static int launch_box(BoxSrc a, BoxSrc b, int h, int w, int k, int k2, int tile_w, float* o0, float* o1, float* o2, hipStream_t st) {
    const size_t smem = sizeof(double) * BX_RH * BX_TW + sizeof(float) * BX_RH * BX_RW * (MODE == 2 ? 2 : 1);
    static bool attr = false;
    auto kern = box_stats_kernel<MODE>;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    dim3 grid((w + BX_TW - 1) / BX_TW, (h + BX_TH - 1) / BX_TH, 4);
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, a, b, h, w, k, k2, tile_w, o0, o1, o2);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

extern "C" int yond_box_stats_self1_f32(const float* bayer, int H, int W, int k, int k2, int tile_w, float* mean,
                                        float* var, float* blur2, void* stream) {
    if (!bayer || !mean || !var || !blur2 || (H & 1) || (W & 1)) return YOND_EINVAL;
    int rc = box_args_ok(H / 2, W / 2, k, tile_w);
    if (rc) return rc;
    rc = box_args_ok(H / 2, W / 2, k2, tile_w);
    if (rc) return rc;
    BoxSrc a{bayer, 1}, b{nullptr, 0};
    return launch_box<0>(a, b, H / 2, W / 2, k, k2, tile_w, mean, var, blur2, (hipStream_t)stream);
}

extern "C" int yond_box_stats_self2_f32(const float* blur2, int h, int w, int k, int tile_w, float* lap, void* stream) {
    if (!blur2 || !lap) return YOND_EINVAL;
    const int rc = box_args_ok(h, w, k, tile_w);
    if (rc) return rc;
    BoxSrc a{blur2, 0}, b{nullptr, 0};
    return launch_box<1>(a, b, h, w, k, k, tile_w, lap, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int yond_box_stats_collab_f32(const float* bayer_lr, const float* bayer_hr, int H, int W, int k, int tile_w,
                                         float* mean, float* var, float* lap, void* stream) {
    if (!bayer_lr || !bayer_hr || !mean || !var || !lap || (H & 1) || (W & 1)) return YOND_EINVAL;
    const int rc = box_args_ok(H / 2, W / 2, k, tile_w);
    if (rc) return rc;
    BoxSrc a{bayer_lr, 1}, b{bayer_hr, 1};
    return launch_box<2>(a, b, H / 2, W / 2, k, k, tile_w, mean, var, lap, (hipStream_t)stream);
}

// =====================================================================================================
// K7: occupancy bitmap + bucketed moment sums
// =====================================================================================================
#define NLF_MAXT 32
#define NLF_BINS 1024        // 1001 used (np.bincount(minlength=nbins+1))

__global__ __launch_bounds__(256) void nlf_accumulate_kernel(const float* __restrict__ lap, const float* __restrict__ mean,
                                                             const float* __restrict__ var, size_t n,
                                                             const double* __restrict__ ths, int nt,
                                                             unsigned int* __restrict__ occ, double* __restrict__ mom) {
    __shared__ double s_ths[NLF_MAXT];
    __shared__ unsigned int s_occ[NLF_MAXT * (NLF_BINS / 32)];
    __shared__ double s_mom[(NLF_MAXT + 1) * 10];
    for (int i = threadIdx.x; i < nt; i += 256) s_ths[i] = ths[i];
    for (int i = threadIdx.x; i < NLF_MAXT * (NLF_BINS / 32); i += 256) s_occ[i] = 0;
    for (int i = threadIdx.x; i < (NLF_MAXT + 1) * 10; i += 256) s_mom[i] = 0.0;
    __syncthreads();
    // each thread walks a contiguous run of elements so that consecutive elements mostly share a bucket
    const size_t per = (n + (size_t)gridDim.x * 256 - 1) / ((size_t)gridDim.x * 256);
    const size_t gtid = (size_t)blockIdx.x * 256 + threadIdx.x;
    // interleave runs of 4 elements across the lanes of a wave: lane l handles elements base + 4*l .. +3
    // (coalesced 16-byte accesses), the wave then moves on by 256 elements
    (void)per; (void)gtid;
    int cur = -1;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;        // all pixels
    double b0 = 0, b1 = 0, b2 = 0, b3 = 0, b4 = 0;        // non-saturated pixels only
    auto flush = [&]() {
        if (cur >= 0) {
            double* m = s_mom + cur * 10;
            if (a0 != 0.0) { atomicAdd(m + 0, a0); atomicAdd(m + 1, a1); atomicAdd(m + 2, a2); atomicAdd(m + 3, a3); atomicAdd(m + 4, a4); }
            if (b0 != 0.0) { atomicAdd(m + 5, b0); atomicAdd(m + 6, b1); atomicAdd(m + 7, b2); atomicAdd(m + 8, b3); atomicAdd(m + 9, b4); }
        }
        a0 = a1 = a2 = a3 = a4 = 0.0;
        b0 = b1 = b2 = b3 = b4 = 0.0;
    };
    const size_t nvec = n / 4;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < nvec + 1; v += (size_t)gridDim.x * 256) {
        float l4[4], m4[4], v4[4];
        int cnt;
        if (v < nvec) {
            const f32x4 L = *(const f32x4*)(lap + v * 4), M = *(const f32x4*)(mean + v * 4), V = *(const f32x4*)(var + v * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { l4[e] = L[e]; m4[e] = M[e]; v4[e] = V[e]; }
            cnt = 4;
        } else {
            cnt = (int)(n - nvec * 4);                       // tail (n % 4 elements), handled by one thread
            for (int e = 0; e < 4; ++e) {
                const size_t i = nvec * 4 + e;
                l4[e] = i < n ? lap[i] : 0.f; m4[e] = i < n ? mean[i] : 0.f; v4[e] = i < n ? var[i] : 0.f;
            }
        }
        for (int e = 0; e < cnt; ++e) {
            const double ld = (double)l4[e];
            int i_le = 0;
            while (i_le < nt && !(ld <= s_ths[i_le])) ++i_le;   // first i with lap <= ths[i]   (YOND_SIDD.py:37)
            int i_lt = i_le;
            while (i_lt < nt && !(ld < s_ths[i_lt])) ++i_lt;    // first i with lap <  ths[i]   (YOND_SIDD.py:77)
            const float mf = m4[e];
            if (i_le < nt) {
                const int bin = (int)__fmul_rn(fminf(fmaxf(mf, 0.0f), 1.0f), 1000.0f);   // (mean.clip(0,1)*nbins).astype(int)
                const unsigned int bit = 1u << (bin & 31);
                unsigned int* w = &s_occ[i_le * (NLF_BINS / 32) + (bin >> 5)];
                if (!(*w & bit)) atomicOr(w, bit);
            }
            if (i_lt != cur) { flush(); cur = i_lt; }
            const double md = (double)mf, vd = (double)v4[e];
            a0 += 1.0; a1 += md; a2 += vd; a3 += md * md; a4 += md * vd;
            if (mf > 1e-4f && mf < 0.8f) { b0 += 1.0; b1 += md; b2 += vd; b3 += md * md; b4 += md * vd; }
        }
    }
    flush();
    __syncthreads();
    for (int i = threadIdx.x; i < nt * (NLF_BINS / 32); i += 256) {
        unsigned int wv = s_occ[i];
        const int t = i / (NLF_BINS / 32), wb = i % (NLF_BINS / 32);
        while (wv) {
            const int bpos = __ffs(wv) - 1;
            wv &= wv - 1;
            occ[(size_t)t * NLF_BINS + wb * 32 + bpos] = 1u;
        }
    }
    for (int i = threadIdx.x; i < (nt + 1) * 10; i += 256) {
        const double s = s_mom[i];
        if (s != 0.0) atomicAdd(mom + i, s);
    }
}

extern "C" size_t yond_nlf_ws_bytes(int nt) {
    (void)nt;
    return 64;
}

extern "C" int yond_nlf_accumulate_f32(const float* lap, const float* mean, const float* var, size_t n, const double* ths,
                                       int nt, uint32_t* occ, double* mom, void* ws, void* stream) {
    (void)ws;
    if (!lap || !mean || !var || !ths || !occ || !mom || n == 0 || nt <= 0 || nt > NLF_MAXT) return YOND_EINVAL;
    if (((uintptr_t)lap | (uintptr_t)mean | (uintptr_t)var) & 15) return YOND_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(occ, 0, sizeof(uint32_t) * nt * NLF_BINS, st);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(mom, 0, sizeof(double) * (nt + 1) * 10, st);
    if (e != hipSuccess) return (int)e;
    size_t nb = (n / 4 + 1 + 256 * 4 - 1) / (256 * 4);
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(nlf_accumulate_kernel, dim3((unsigned)nb), dim3(256), 0, st, lap, mean, var, n, ths, nt, occ, mom);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}
