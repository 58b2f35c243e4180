// K5-K7: the noise-level estimator's data passes (YOND_SIDD.py:13-115, utils/isp_algos.py:234-242, 345-365).
//   K5  box statistics: cv2.blur semantics (normalised k x k window, BORDER_REFLECT_101), window sums in
//       float64 (the float32 inputs make them exact), every intermediate rounded to float32 exactly where
//       NumPy / OpenCV round (no fused multiply-adds across those points).
//   K6  exact order statistics + np.percentile's linear interpolation: nle_select.hip.
//   K7  occupancy of the 1/1000 mean bins per threshold bucket (one sweep), score-3 selection on the device,
//       and the five moment sums of the least-squares line below the selected threshold (one sweep).
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
// K7: score-3 threshold selection and the moment sums of the least-squares line
// =====================================================================================================
// K7a  occupancy: one sweep over (lap, mean), seen as a [rows][width] matrix (the planar maps are [4*h][w]):
//      which of the 1/1000 mean bins are occupied among {lap <= ths[i]} for each of the nt thresholds.
//      A lane owns VEC adjacent columns and walks a segment of rows (adjacent lanes read adjacent 16-byte
//      vectors, NLF_BATCH rows ahead).  The thresholds live in scalar registers as float32
//      (x <= T  <=>  x <= floor32(T) for float32 x and float64 T) and the bucket index is a branch-free count
//      of compares; `mean` is a smooth map, so a lane remembers which buckets it has already marked for its
//      current bin and only new (bucket, bin) pairs reach the LDS bitmap (no-return atomic OR).
// K7s  score3 on the device (one wave): npeaks by prefix-OR + popcount, score = ths / (quants * npeaks),
//      first minimum over i >= 1 (YOND_SIDD.py:37-47) -> the selected threshold stays on the device.
// K7b  moments: one streaming sweep over (lap, mean, var), sums {n, Sm, Sv, Smm, Smv} over lap < th in
//      float64 registers (all pixels / 1e-4 < mean < 0.8 only), wave + workgroup reduction, 10 atomics per
//      workgroup.  (A single bucketed pass was tried first: on real frames lap changes bucket every 2-3
//      pixels, and the float64 LDS atomics of the per-bucket sums cost 3x the whole sweep.)
#define NLF_MAXT 32
#define NLF_BINS 1024        // 1001 used (np.bincount(minlength=nbins+1))
#define NLF_WORDS (NLF_BINS / 32)
#define NLF_SEG 16           // rows per lane segment
#define NLF_BATCH 4          // rows loaded ahead

template <int VEC, int MAXT>
__global__ __launch_bounds__(256) void nlf_occupancy_kernel(const float* __restrict__ lap, const float* __restrict__ mean,
                                                            int rows, int width, const double* __restrict__ ths, int nt,
                                                            unsigned int* __restrict__ occ) {
    __shared__ unsigned int s_occ[NLF_MAXT * NLF_WORDS];
    const int tid = threadIdx.x, lane = tid & 63;
    float te = INFINITY;                                       // padding: never below a finite value
    if (lane < nt) {
        const double T = ths[lane];
        const float f = (float)T;
        te = ((double)f > T) ? nextafterf(f, -INFINITY) : f;   // largest float32 <= T
    }
    float tle[MAXT];                                           // wave-uniform -> scalar registers
#pragma unroll
    for (int i = 0; i < MAXT; ++i) tle[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(te), i));
    for (int i = tid; i < NLF_MAXT * NLF_WORDS; i += 256) s_occ[i] = 0;
    __syncthreads();
    const int G = (width + VEC - 1) / VEC;
    const int nseg = (rows + NLF_SEG - 1) / NLF_SEG;
    const size_t nitems = (size_t)G * nseg;
    for (size_t item = (size_t)blockIdx.x * 256 + tid; item < nitems; item += (size_t)gridDim.x * 256) {
        const int sgm = (int)(item / G), g = (int)(item % G);
        const int r0 = sgm * NLF_SEG, r1 = min(rows, r0 + NLF_SEG);
        const int c0 = g * VEC;
        int cbin[VEC];
        unsigned int cmask[VEC];                               // buckets already marked for the column's current bin
#pragma unroll
        for (int e = 0; e < VEC; ++e) { cbin[e] = -1; cmask[e] = 0; }
        for (int rb = r0; rb < r1; rb += NLF_BATCH) {
            f32x4 Lb[NLF_BATCH], Mb[NLF_BATCH];
#pragma unroll
            for (int j = 0; j < NLF_BATCH; ++j) {
                const int r = min(rb + j, r1 - 1);
                const size_t base = (size_t)r * width + c0;
                if (VEC == 4) { Lb[j] = *(const f32x4*)(lap + base); Mb[j] = *(const f32x4*)(mean + base); }
                else { Lb[j][0] = lap[base]; Mb[j][0] = mean[base]; }
            }
#pragma unroll
            for (int j = 0; j < NLF_BATCH; ++j) {
                if (rb + j >= r1) break;
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float l = Lb[j][e], mf = Mb[j][e];
                    int ce = 0;
#pragma unroll
                    for (int i = 0; i < MAXT; ++i) ce += (l <= tle[i]) ? 1 : 0;
                    const int ile = min(MAXT - ce, nt);                  // first i with lap <= ths[i] (YOND_SIDD.py:37); NaN -> nt
                    const int bin = (int)__fmul_rn(fminf(fmaxf(mf, 0.0f), 1.0f), 1000.0f);   // (mean.clip(0,1)*nbins).astype(int)
                    if (bin != cbin[e]) { cbin[e] = bin; cmask[e] = 0; }
                    const unsigned int bit = ile < nt ? (1u << ile) : 0u;
                    if (bit & ~cmask[e]) {
                        cmask[e] |= bit;
                        atomicOr(&s_occ[ile * NLF_WORDS + (bin >> 5)], 1u << (bin & 31));
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < nt * NLF_WORDS; i += 256) {
        const unsigned int wv = s_occ[i];
        if (wv) atomicOr(occ + i, wv);
    }
}

struct ScoreArgs { double quants[NLF_MAXT]; };

__global__ __launch_bounds__(64) void nlf_score3_kernel(const unsigned int* __restrict__ occ, const double* __restrict__ ths,
                                                        ScoreArgs qa, int nt, double* __restrict__ sel, int* __restrict__ npeaks) {
    __shared__ unsigned int s_occ[NLF_MAXT * NLF_WORDS];
    const int lane = threadIdx.x;
    for (int i = lane; i < nt * NLF_WORDS; i += 64) s_occ[i] = occ[i];
    __syncthreads();
    // lane w < 32 owns bitmap word w: running OR over the buckets, popcount, sum over the words
    int np = 0;
    unsigned int acc = 0;
    for (int j = 0; j < nt; ++j) {
        if (lane < NLF_WORDS) acc |= s_occ[j * NLF_WORDS + lane];           // bins seen among lap <= ths[j]
        int c = lane < NLF_WORDS ? __popc(acc) : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
        if (lane == j) np = c;
    }
    double score = INFINITY;
    if (lane < nt) {
        npeaks[lane] = np;
        score = __ddiv_rn(ths[lane], __dmul_rn(qa.quants[lane], (double)np));   // YOND_SIDD.py:45
    }
    // first minimum over 1 <= i < nt (np.argmin(score[1:]) + 1)
    double best = (lane >= 1 && lane < nt) ? score : INFINITY;
    int bi = (lane >= 1 && lane < nt) ? lane : 0x7FFFFFFF;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (bi == 0x7FFFFFFF) bi = 0;                              // nt == 1 (or no finite score): first threshold
    if (lane == 0) {
        sel[0] = (double)bi;
        sel[1] = ths[bi];
        sel[2] = qa.quants[bi];
        sel[3] = best;
    }
}

#define MOM_UNROLL 4

__global__ __launch_bounds__(256) void nlf_moments_kernel(const float* __restrict__ lap, const float* __restrict__ mean,
                                                          const float* __restrict__ var, size_t n, int vec_ok,
                                                          const double* __restrict__ th_ptr, double* __restrict__ mom) {
    __shared__ double s_red[4][10];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double T = *th_ptr;
    const float f = (float)T;
    const float tl = ((double)f < T) ? nextafterf(f, INFINITY) : f;        // lap < T  <=>  lap < ceil32(T)
    double a[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) a[i] = 0.0;
    auto add = [&](float l, float mf, float vf) {
        if (l < tl) {                                                       // YOND_SIDD.py:77 / :105
            const double md = (double)mf, vd = (double)vf;
            const double mm = md * md, mv = md * vd;
            a[0] += 1.0; a[1] += md; a[2] += vd; a[3] += mm; a[4] += mv;
            if (mf > 1e-4f && mf < 0.8f) { a[5] += 1.0; a[6] += md; a[7] += vd; a[8] += mm; a[9] += mv; }   // utils/isp_algos.py:348
        }
    };
    const size_t nvec = vec_ok ? n / 4 : 0;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t v = (size_t)blockIdx.x * 256 + tid; v < nvec; v += stride * MOM_UNROLL) {
        f32x4 L[MOM_UNROLL], M[MOM_UNROLL], V[MOM_UNROLL];
#pragma unroll
        for (int u = 0; u < MOM_UNROLL; ++u) {
            const size_t vu = v + u * stride;
            const size_t idx = (vu < nvec ? vu : v) * 4;
            L[u] = *(const f32x4*)(lap + idx); M[u] = *(const f32x4*)(mean + idx); V[u] = *(const f32x4*)(var + idx);
        }
#pragma unroll
        for (int u = 0; u < MOM_UNROLL; ++u) {
            if (v + u * stride < nvec) {
#pragma unroll
                for (int e = 0; e < 4; ++e) add(L[u][e], M[u][e], V[u][e]);
            }
        }
    }
    for (size_t i = nvec * 4 + (size_t)blockIdx.x * 256 + tid; i < n; i += stride) add(lap[i], mean[i], var[i]);
#pragma unroll
    for (int i = 0; i < 10; ++i) a[i] = wave_sum(a[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 10; ++i) s_red[wave][i] = a[i];
    }
    __syncthreads();
    if (tid < 10) {
        const double t = (s_red[0][tid] + s_red[1][tid]) + (s_red[2][tid] + s_red[3][tid]);
        if (t != 0.0) atomicAdd(mom + tid, t);
    }
}

template <int VEC, int MAXT>
static void launch_occupancy(unsigned nb, hipStream_t st, const float* lap, const float* mean, int rows, int width,
                             const double* ths, int nt, uint32_t* occ) {
    hipLaunchKernelGGL((nlf_occupancy_kernel<VEC, MAXT>), dim3(nb), dim3(256), 0, st, lap, mean, rows, width, ths, nt, occ);
}

extern "C" int yond_nlf_occupancy_f32(const float* lap, const float* mean, size_t n, int width, const double* ths, int nt,
                                      uint32_t* occ, void* stream) {
    if (!lap || !mean || !ths || !occ || n == 0 || nt <= 0 || nt > NLF_MAXT) return YOND_EINVAL;
    if (width <= 0 || n % (size_t)width != 0 || n / (size_t)width > 0x7FFFFFFFull) return YOND_EINVAL;
    const int rows = (int)(n / (size_t)width);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(occ, 0, sizeof(uint32_t) * nt * NLF_WORDS, st);
    if (e != hipSuccess) return (int)e;
    const bool vec = (width % 4 == 0) && !(((uintptr_t)lap | (uintptr_t)mean) & 15);
    const int G = vec ? width / 4 : width;
    const size_t nitems = (size_t)G * ((rows + NLF_SEG - 1) / NLF_SEG);
    size_t nbs = (nitems + 255) / 256;
    if (nbs > 4096) nbs = 4096;
    if (nbs < 1) nbs = 1;
    const unsigned nb = (unsigned)nbs;
    if (vec) {
        if (nt <= 8) launch_occupancy<4, 8>(nb, st, lap, mean, rows, width, ths, nt, occ);
        else if (nt <= 24) launch_occupancy<4, 24>(nb, st, lap, mean, rows, width, ths, nt, occ);
        else launch_occupancy<4, 32>(nb, st, lap, mean, rows, width, ths, nt, occ);
    } else {
        if (nt <= 8) launch_occupancy<1, 8>(nb, st, lap, mean, rows, width, ths, nt, occ);
        else if (nt <= 24) launch_occupancy<1, 24>(nb, st, lap, mean, rows, width, ths, nt, occ);
        else launch_occupancy<1, 32>(nb, st, lap, mean, rows, width, ths, nt, occ);
    }
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

extern "C" int yond_nlf_score3_f64(const uint32_t* occ, const double* ths, const double* quants_host, int nt, double* sel,
                                   int32_t* npeaks, void* stream) {
    if (!occ || !ths || !quants_host || !sel || !npeaks || nt <= 0 || nt > NLF_MAXT) return YOND_EINVAL;
    ScoreArgs qa;
    for (int i = 0; i < NLF_MAXT; ++i) qa.quants[i] = i < nt ? quants_host[i] : 1.0;
    hipLaunchKernelGGL(nlf_score3_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, occ, ths, qa, nt, sel, npeaks);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

extern "C" int yond_nlf_moments_f32(const float* lap, const float* mean, const float* var, size_t n, const double* th,
                                    double* mom, void* stream) {
    if (!lap || !mean || !var || !th || !mom || n == 0) return YOND_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(mom, 0, sizeof(double) * 10, st);
    if (e != hipSuccess) return (int)e;
    const int vec_ok = !(((uintptr_t)lap | (uintptr_t)mean | (uintptr_t)var) & 15);
    size_t nb = (n / 4 + 256 * MOM_UNROLL - 1) / (256 * MOM_UNROLL);
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(nlf_moments_kernel, dim3((unsigned)nb), dim3(256), 0, st, lap, mean, var, n, vec_ok, th, mom);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}
