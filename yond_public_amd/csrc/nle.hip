// K5-K7: the noise-level estimator's data passes (YOND_SIDD.py:13-115, utils/isp_algos.py:234-242, 345-365).
//   K5  box statistics: cv2.blur semantics (normalised k x k window, BORDER_REFLECT_101), window sums in
//       float64, every intermediate rounded to float32 exactly where NumPy / OpenCV round (no fused
//       multiply-adds across those points).
//   K6  exact order statistics + np.percentile's linear interpolation: nle_select.hip.
//   K7  occupancy of the 1/1000 mean bins per threshold bucket (one sweep), score-3 selection on the device,
//       and the five moment sums of the least-squares line below the selected threshold (one sweep).
// All of it is integer / streaming work bound by HBM and LDS, not MFMA.
#include "nle_common.h"

// =====================================================================================================
// K5: box statistics, streamed down the rows
// =====================================================================================================
// A workgroup owns one plane, a strip of <= 256 - 2R "virtual" columns (outputs plus the reflected halo of
// R = k/2 on either side, one column per thread) and a segment of rows.  It walks down the rows:
//   vertical   every thread keeps the running k-row sums of its column in float64 registers
//              (S += entering - leaving; the leaving row is read again from L2) -- float32 data summed in
//              float64 is exact, so add/subtract leaves no drift;
//   horizontal the k-column window sum is a difference of two prefix sums across the strip: a DPP
//              inclusive scan inside each wave (row_shr 1/2/4/8, row_bcast 15/31), wave totals through LDS,
//              then out[c] = Q[c + R] - Q[c - R - 1].  The prefix over <= 256 columns costs a relative
//              error of ~1e-15 on the window sum, far below the float32 rounding that follows.
// BS_B rows are processed per barrier pair.  Every intermediate is rounded to float32 exactly where NumPy /
// OpenCV round (cv2.blur returns float32; stdfilt squares and subtracts in float32; no FMA contraction).
#define BS_T 256
#define BS_B 4
#define BS_MAXR 14
#define BS_MAXOH 256         // rows per segment (row table in LDS)

struct BoxSrc {
    const float* p;     // base pointer
    int bayer;          // 1: Bayer frame [2h][2w], plane = blockIdx.z ; 0: planar [4][h][w]
};

struct BoxGeom {
    int h, w, k, k2, tile_w;
    int ow_nom, nstrip, oh;     // outputs per strip, strips per (tile_w-wide) block, rows per segment
};

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_take(double v) {
    // lanes without a source (or masked rows) receive +0.0
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}

// inclusive prefix sums across the 64 lanes of N independent values, step by step over all of them so that the
// dependent DPP -> add chains of different values overlap
template <int N>
__device__ __forceinline__ void wave_incl_scan_f64(double (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_take<0x111, 0xf>(v[i]);       // row_shr:1
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_take<0x112, 0xf>(v[i]);       // row_shr:2
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_take<0x114, 0xf>(v[i]);       // row_shr:4
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_take<0x118, 0xf>(v[i]);       // row_shr:8
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_take<0x142, 0xa>(v[i]);       // row_bcast:15 into rows 1, 3
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_take<0x143, 0xc>(v[i]);       // row_bcast:31 into rows 2, 3
}

__device__ __forceinline__ float blur_round(double s, int k) { return (float)(s * (1.0 / (double)(k * k))); }

// std = sqrt(max(B(x^2) - B(x)^2, 0)) in float32 steps (utils/isp_algos.py:236-241)
__device__ __forceinline__ float std_from(float b1, float b2) {
    const float d = __fsub_rn(b2, __fmul_rn(b1, b1));
    return __fsqrt_rn(fmaxf(d, 0.0f));
}

// mode 0: self stage 1 (mean, var, blur2 from the Bayer frame); 1: self stage 2 (lap from blur2);
// 2: collab (mean, var, lap from noisy + denoised Bayer frames)
// STATS (MODE 0): stage 1 reads every pixel of the frame exactly once as an "own" pixel, so it also collects the frame
// maximum (lr.max() for the bias LUT grid, YOND_SIDD.py:256/393) into st->frame_max_key.  (Round 2 also tried the first
// sweep of the threshold selection inside the producers of the lap map: with ~1000 workgroups each flushing its LDS
// histogram the fold cost 66 us against 62 us for the stand-alone sweep of nle_fast.hip -- not kept.)
template <int MODE, bool STATS>
__global__ __launch_bounds__(BS_T) void box_stream_kernel(BoxSrc a, BoxSrc b, BoxGeom g, float* __restrict__ o0,
                                                         float* __restrict__ o1, float* __restrict__ o2, NleState* st) {
    static_assert(!STATS || MODE == 0, "only stage 1 collects the frame maximum");
    constexpr int NQ = MODE == 0 ? 3 : (MODE == 1 ? 2 : 4);      // running sums per column
    float fmax_ = -INFINITY;
    constexpr int NI = MODE == 2 ? 2 : 1;                        // input frames
    constexpr int NL = MODE == 0 ? 3 : 2;                        // loads per input and row: entering, leaving (k), leaving (k2)
    __shared__ double s_p[NQ][BS_B][BS_T];
    __shared__ double s_tot[NQ][BS_B][BS_T / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int plane = blockIdx.z;
    const int h = g.h, w = g.w, k = g.k, k2 = g.k2;
    const int R = k / 2, R2 = k2 / 2;
    // columns: reflect inside [bx0, bx0 + bw) -- bw = tile_w (SIDD_256 re-tiling) or the whole width
    const int bw = g.tile_w > 0 ? g.tile_w : w;
    const int blk = blockIdx.x / g.nstrip, strip = blockIdx.x % g.nstrip;
    const int bx0 = blk * bw;
    const int ox0 = bx0 + strip * g.ow_nom;
    const int ow = min(g.ow_nom, bx0 + bw - ox0);
    const bool live = tid < ow + 2 * R;
    const int rc = bx0 + reflect101(ox0 - R + tid - bx0, bw);
    const bool writer = tid >= R && tid < R + ow;
    const int ox = ox0 + tid - R;
    // rows
    const int oy0 = blockIdx.y * g.oh;
    const int ohe = min(g.oh, h - oy0);
    const int nsteps = ohe + 2 * R;
    const float* base[NI];
    size_t rstride[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const BoxSrc src = i == 0 ? a : b;
        if (src.bayer) { base[i] = src.p + (size_t)(plane >> 1) * (2 * w) + 2 * rc + (plane & 1); rstride[i] = (size_t)4 * w; }
        else { base[i] = src.p + (size_t)plane * h * w + rc; rstride[i] = (size_t)w; }
    }
    // image row of every local row l (reflect(oy0 - R + l)), once per workgroup
    __shared__ int s_row[BS_MAXOH + 2 * BS_MAXR];
    for (int l = tid; l < nsteps; l += BS_T) s_row[l] = reflect101(oy0 - R + l, h);
    __syncthreads();
    // value of local row l, l clamped into [0, nsteps).  Nothing depends on the loaded value until the batch is
    // consumed, so the loads of a batch stay in flight together.  Idle columns and rows past the end read valid
    // addresses: what they add only reaches prefix positions / rows that are never emitted.  Rows before the
    // start (the "leaving" row of the first k steps) are zeroed by a 0/1 factor when the batch is consumed.
    auto ld = [&](int i, int l) -> float {
        const int gy = s_row[min(max(l, 0), nsteps - 1)];
        return base[i][(size_t)gy * rstride[i]];
    };
    float cur[NI][NL][BS_B], nxt[NI][NL][BS_B];
    auto load_batch = [&](float (&dst)[NI][NL][BS_B], int l0) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
#pragma unroll
            for (int r = 0; r < BS_B; ++r) {
                dst[i][0][r] = ld(i, l0 + r);
                dst[i][1][r] = ld(i, l0 + r - k);
                if (NL == 3) dst[i][2][r] = ld(i, l0 + r - k2);
            }
        }
    };
    double S[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) S[q] = 0.0;
    load_batch(nxt, 0);
    for (int l0 = 0; l0 < nsteps; l0 += BS_B) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
#pragma unroll
            for (int j = 0; j < NL; ++j) {
#pragma unroll
                for (int r = 0; r < BS_B; ++r) cur[i][j][r] = nxt[i][j][r];
            }
        }
        if (l0 + BS_B < nsteps) load_batch(nxt, l0 + BS_B);
        double P[NQ * BS_B];
#pragma unroll
        for (int r = 0; r < BS_B; ++r) {
            const float mk = (l0 + r >= k) ? 1.0f : 0.0f, mk2 = (l0 + r >= k2) ? 1.0f : 0.0f;   // leaving row exists
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const float xn = cur[i][0][r], xo = __fmul_rn(cur[i][1][r], mk);
                S[2 * i] += (double)xn - (double)xo;
                S[2 * i + 1] += (double)__fmul_rn(xn, xn) - (double)__fmul_rn(xo, xo);
            }
            if (MODE == 0) S[2] += (double)cur[0][0][r] - (double)__fmul_rn(cur[0][2][r], mk2);
            if (STATS) fmax_ = fmaxf(fmax_, cur[0][0][r]);       // halo, idle columns and clamped rows are pixels of the frame too
#pragma unroll
            for (int q = 0; q < NQ; ++q) P[q * BS_B + r] = S[q];
        }
        // warm-up rows (no window complete yet) only feed the vertical sums: uniform skip of the horizontal pass
        if (l0 + BS_B <= (MODE == 0 ? R + R2 : 2 * R)) continue;
        wave_incl_scan_f64(P);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int r = 0; r < BS_B; ++r) {
                s_p[q][r][tid] = P[q * BS_B + r];
                if (lane == 63) s_tot[q][r][wave] = P[q * BS_B + r];
            }
        }
        __syncthreads();
        if (writer) {
#pragma unroll
            for (int r = 0; r < BS_B; ++r) {
                const int l = l0 + r;
                // window sum of quantity q with radius rad around this thread's column
                auto win = [&](int q, int rad) -> double {
                    // Q[hi] - Q[lo] with Q = in-wave prefix + totals of the waves before: hi and lo are at most
                    // one wave apart (2 * rad + 1 <= 29 < 64), so the totals cancel except for the wave of lo
                    const int hi = tid + rad, lo = tid - rad - 1;
                    double d = s_p[q][r][hi];
                    if (lo >= 0) {
                        d -= s_p[q][r][lo];
                        if ((hi >> 6) != (lo >> 6)) d += s_tot[q][r][lo >> 6];
                    }
                    return d;
                };
                const int oy = oy0 + l - 2 * R;                                   // k-window centred here is complete
                if (l >= 2 * R && l < nsteps) {
                    const size_t idx = ((size_t)plane * h + oy) * w + ox;
                    if (MODE == 0) {
                        const float m = blur_round(win(0, R), k);
                        const float sd = std_from(m, blur_round(win(1, R), k));
                        o0[idx] = m;
                        o1[idx] = __fmul_rn(sd, sd);                              // var = lr_rggb_k**2 (YOND_SIDD.py:72)
                    } else if (MODE == 1) {
                        o0[idx] = std_from(blur_round(win(0, R), k), blur_round(win(1, R), k));
                    } else {
                        const float sl = std_from(blur_round(win(0, R), k), blur_round(win(1, R), k));
                        const float mh = blur_round(win(2, R), k);
                        const float sh = std_from(mh, blur_round(win(3, R), k));
                        o0[idx] = mh;                                              // mean = blur(hr) (YOND_SIDD.py:97)
                        o1[idx] = __fsub_rn(__fmul_rn(sl, sl), __fmul_rn(sh, sh));  // var = lr_k**2 - hr_k**2 (:96)
                        o2[idx] = sh;                                              // img_lap = hr_k (:98)
                    }
                }
                if (MODE == 0) {
                    const int oy2 = oy0 + l - R - R2;                             // k2-window centred here is complete
                    if (l >= R + R2 && oy2 < oy0 + ohe)
                        o2[((size_t)plane * h + oy2) * w + ox] = blur_round(win(2, R2), k2);
                }
            }
        }
        __syncthreads();
    }
    if constexpr (STATS) {
        fmax_ = wave_max(fmax_);
        // ~4000 waves, one word: look first (a coherent load) -- after the first few arrivals nearly nobody has to write
        if (lane == 0 && fmax_ > -INFINITY) {
            const unsigned int key = f2key(fmax_);
            if (key > __hip_atomic_load(&st->frame_max_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&st->frame_max_key, key);
        }
    }
}

static int box_args_ok(int h, int w, int k, int tile_w) {
    if (h < 1 || w < 1) return YOND_EINVAL;
    if (k < 1 || !(k & 1)) return YOND_EINVAL;
    if (k > 2 * BS_MAXR + 1) return YOND_EUNSUPPORTED;
    if (tile_w < 0) return YOND_EINVAL;
    if (tile_w > 0 && w % tile_w != 0) return YOND_EUNSUPPORTED;
    return YOND_OK;
}

template <int MODE, bool STATS = false>
static int launch_box(BoxSrc a, BoxSrc b, int h, int w, int k, int k2, int tile_w, float* o0, float* o1, float* o2, hipStream_t st,
                      NleState* state = nullptr) {
    if (k2 > k) return YOND_EUNSUPPORTED;
    BoxGeom g;
    g.h = h; g.w = w; g.k = k; g.k2 = k2; g.tile_w = tile_w;
    const int bw = tile_w > 0 ? tile_w : w;
    const int nblk = w / bw;
    const int maxow = BS_T - 2 * (k / 2);
    g.nstrip = (bw + maxow - 1) / maxow;
    g.ow_nom = (bw + g.nstrip - 1) / g.nstrip;
    // row segments: about four workgroups per CU in one round; longer segments re-read fewer halo rows
    const long target = yond_exp_long("YOND_BOX_WGS", 1024);
    const long cols = 4L * nblk * g.nstrip;
    long nseg = (target + cols / 2) / cols;
    if (nseg < 1) nseg = 1;
    g.oh = (int)((h + nseg - 1) / nseg);
    if (g.oh < 16) g.oh = 16;
    if (g.oh > BS_MAXOH) g.oh = BS_MAXOH;
    if (g.oh > h) g.oh = h;
    const int nsy = (h + g.oh - 1) / g.oh;
    dim3 grid((unsigned)(nblk * g.nstrip), (unsigned)nsy, 4);
    hipLaunchKernelGGL((box_stream_kernel<MODE, STATS>), grid, dim3(BS_T), 0, st, a, b, g, o0, o1, o2, state);
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

// The default producers of the estimator: the kernels above (stage 1 also collects the frame maximum) followed by sweep 1 of
// the threshold selection (nle_fast.hip) on the workspace -- one call per frame.  `blur2` is a scratch plane [4][h][w].
extern "C" int yond_box_stats_self_stats_f32(const float* bayer, int H, int W, int k, int k2, int tile_w, float* mean, float* var,
                                             float* blur2, float* lap, const double* q_host, int nq, void* ws, void* stream) {
    if (!bayer || !mean || !var || !blur2 || !lap || !ws || ((uintptr_t)ws & 15) || (H & 1) || (W & 1)) return YOND_EINVAL;
    int rc = box_args_ok(H / 2, W / 2, k, tile_w);
    if (rc) return rc;
    rc = box_args_ok(H / 2, W / 2, k2, tile_w);
    if (rc) return rc;
    const int h = H / 2, w = W / 2;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(ws, 0, nf_state_bytes(), st);
    if (e != hipSuccess) return (int)e;
    BoxSrc s1{bayer, 1}, s2{blur2, 0}, nb{nullptr, 0};
    rc = launch_box<0, true>(s1, nb, h, w, k, k2, tile_w, mean, var, blur2, st, (NleState*)ws);
    if (rc) return rc;
    rc = launch_box<1>(s2, nb, h, w, k, k, tile_w, lap, nullptr, nullptr, st);
    if (rc) return rc;
    return nf_launch_stats(lap, mean, (size_t)4 * h * w, w, q_host, nq, ws, st, false);
}

extern "C" int yond_box_stats_collab_stats_f32(const float* bayer_lr, const float* bayer_hr, int H, int W, int k, int tile_w,
                                               float* mean, float* var, float* lap, const double* q_host, int nq, void* ws,
                                               void* stream) {
    if (!bayer_lr || !bayer_hr || !mean || !var || !lap || !ws || ((uintptr_t)ws & 15) || (H & 1) || (W & 1)) return YOND_EINVAL;
    int rc = box_args_ok(H / 2, W / 2, k, tile_w);
    if (rc) return rc;
    const int h = H / 2, w = W / 2;
    hipStream_t st = (hipStream_t)stream;
    BoxSrc s1{bayer_lr, 1}, s2{bayer_hr, 1};
    rc = launch_box<2>(s1, s2, h, w, k, k, tile_w, mean, var, lap, st);
    if (rc) return rc;
    return nf_launch_stats(lap, mean, (size_t)4 * h * w, w, q_host, nq, ws, st, true);
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
    // lap first: the selected pixels (a low percentile of a smooth map) cluster, so most waves find nothing below the
    // threshold in their 256 elements and never read mean / var there (144 MB -> lap + the selected neighbourhoods)
    for (size_t v = (size_t)blockIdx.x * 256 + tid; v < nvec; v += stride * MOM_UNROLL) {
        f32x4 L[MOM_UNROLL], M[MOM_UNROLL], V[MOM_UNROLL];
        bool take[MOM_UNROLL];
#pragma unroll
        for (int u = 0; u < MOM_UNROLL; ++u) {
            const size_t vu = v + u * stride;
            L[u] = *(const f32x4*)(lap + (vu < nvec ? vu : v) * 4);
        }
#pragma unroll
        for (int u = 0; u < MOM_UNROLL; ++u) {
            const bool mine = v + u * stride < nvec && (L[u][0] < tl || L[u][1] < tl || L[u][2] < tl || L[u][3] < tl);
            take[u] = __ballot(mine) != 0ull;                               // wave-uniform
            if (take[u]) {
                const size_t vu = v + u * stride;
                const size_t idx = (vu < nvec ? vu : v) * 4;
                M[u] = *(const f32x4*)(mean + idx); V[u] = *(const f32x4*)(var + idx);
            }
        }
#pragma unroll
        for (int u = 0; u < MOM_UNROLL; ++u) {
            if (take[u] && v + u * stride < nvec) {
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

static int launch_moments(const float* lap, const float* mean, const float* var, size_t n, const double* th, double* mom,
                          bool zero, hipStream_t st);

extern "C" int yond_nlf_moments_f32(const float* lap, const float* mean, const float* var, size_t n, const double* th,
                                    double* mom, void* stream) {
    if (!lap || !mean || !var || !th || !mom || n == 0) return YOND_EINVAL;
    return launch_moments(lap, mean, var, n, th, mom, true, (hipStream_t)stream);
}

// the same sweep on the workspace of the two-sweep selection (nle_fast.hip): threshold = sel[1], sums -> mom, which the
// workspace reset of sweep 1 has already zeroed (no memset launch of its own)
extern "C" int yond_nle_moments_f32(const float* lap, const float* mean, const float* var, size_t n, void* ws, void* stream) {
    if (!lap || !mean || !var || !ws || n == 0) return YOND_EINVAL;
    NleState* s = (NleState*)ws;
    return launch_moments(lap, mean, var, n, &s->sel[1], s->mom, false, (hipStream_t)stream);
}

static int launch_moments(const float* lap, const float* mean, const float* var, size_t n, const double* th, double* mom,
                          bool zero, hipStream_t st) {
    if (zero) {
        hipError_t e = hipMemsetAsync(mom, 0, sizeof(double) * 10, st);
        if (e != hipSuccess) return (int)e;
    }
    const int vec_ok = !(((uintptr_t)lap | (uintptr_t)mean | (uintptr_t)var) & 15);
    size_t nb = (n / 4 + 256 * MOM_UNROLL - 1) / (256 * MOM_UNROLL);
    const size_t cap = (size_t)yond_exp_long("YOND_MOM_WGS", 512);   // two workgroups per CU: 24.7 us (41 at 2048: ten same-line atomics per workgroup)
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(nlf_moments_kernel, dim3((unsigned)nb), dim3(256), 0, st, lap, mean, var, n, vec_ok, th, mom);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}
