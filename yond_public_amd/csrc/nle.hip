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
// K5: box statistics -- vertical running sums per column, horizontal sliding windows per chunk of columns
// =====================================================================================================
// A workgroup owns TWO planes (Bayer input: the two column phases dx = 0, 1 of one row parity, read together as one 8-byte
// load per packed pixel; planar input: planes 2z, 2z + 1), a strip of <= 256 - 2R "virtual" columns (outputs plus the
// reflected halo of R = k/2 on either side) and a segment of rows.  It walks down the rows BB = 4 at a time (two barriers per batch):
//   column phase   thread = virtual column: running k-row sums of its column in float64 registers (S += entering - leaving;
//                  float32 data summed in float64 is exact, so add / subtract leaves no drift; the leaving row comes back
//                  from L2), written to LDS;
//   task phase     thread = (plane, row of the batch, chunk of BX_C = 8 output columns): the k-column window sum of the
//                  chunk's first output is added up from LDS, the next seven SLIDE (+ entering column - leaving column: two
//                  LDS reads at immediate offsets and two float64 additions per sum and output), then the float32 finishing
//                  arithmetic exactly where NumPy / OpenCV round (cv2.blur returns float32; stdfilt squares and subtracts in
//                  float32; no FMA contraction) and two 16-byte stores per map: the lanes of a half-wave hold neighbouring
//                  chunks, so a store instruction covers 1 KB of one output row.
// LDS row of a quantity: one double per virtual column, one pad double per chunk (a half-wave's 32 chunk bases fall on 32
// different bank pairs).  Window sizes are template constants for the shipped k = 29 / k2 = 19; any other odd k <= 29 takes
// the same code with run-time loop bounds.  (Rounds 1-2 did the horizontal pass with wave-wide DPP prefix scans, one column
// per thread: ~18 vector instructions per sum and output against ~5 here, and 4-byte loads at an 8-byte stride.)
#define BX_T 256
#define BX_C 8
#define BX_MAXR 14
#define BX_VW (BX_T + BX_T / BX_C)      // doubles per LDS row

struct BoxSrc {
    const float* p;     // base pointer
    int bayer;          // 1: Bayer frame [2h][2w]; 0: planar [4][h][w]
};

struct BoxGeom {
    int h, w, k, k2, tile_w;
    int ow_nom, nstrip, oh;     // outputs per strip, strips per (tile_w-wide) block, rows per segment
};

__device__ __forceinline__ float blur_round(double s, int k) { return (float)(s * (1.0 / (double)(k * k))); }

// std = sqrt(max(B(x^2) - B(x)^2, 0)) in float32 steps (utils/isp_algos.py:236-241)
__device__ __forceinline__ float std_from(float b1, float b2) {
    const float d = __fsub_rn(b2, __fmul_rn(b1, b1));
    return __fsqrt_rn(fmaxf(d, 0.0f));
}

// eight consecutive outputs of one row: 16-byte stores where the row allows it
__device__ __forceinline__ void box_store8(float* __restrict__ p, const float (&v)[BX_C], int nvalid, bool vec_ok) {
    if (vec_ok && nvalid >= BX_C) {
        // (non-temporal: the maps stream out once and should not push the input rows -- which come back as "leaving" rows 19 / 29
        // steps later -- out of the XCD's L2)
        __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]}, (f32x4*)p);
        __builtin_nontemporal_store(f32x4{v[4], v[5], v[6], v[7]}, (f32x4*)(p + 4));
    } else {
#pragma unroll
        for (int i = 0; i < BX_C; ++i)
            if (i < nvalid) p[i] = v[i];
    }
}

// mode 0: self stage 1 (mean, var, blur2 from the Bayer frame); 1: self stage 2 (lap from blur2);
// 2: collab (mean, var, lap from noisy + denoised Bayer frames)
// STATS (MODE 0): stage 1 reads every pixel of the frame as an entering pixel, so it also collects the frame maximum
// (lr.max() for the bias LUT grid, YOND_SIDD.py:256/393) into st->frame_max_key.
// ROLES (experiment): 512 threads -- waves 0-3 only walk the columns (running sums of batch i into LDS buffer i & 1), waves 4-7 only
// run the task phase (window sums of batch i - 1 from the other buffer): the two phases of consecutive batches overlap inside one
// workgroup behind ONE barrier per batch, at twice the LDS (one workgroup per CU instead of two).
template <int MODE, int KT, int K2T, bool STATS, int BB, bool ROLES = false>
__global__ __launch_bounds__(ROLES ? 512 : BX_T, ROLES ? 1 : (MODE == 1 ? 4 : 2)) void box_slide_kernel(BoxSrc a, BoxSrc b, BoxGeom g, float* __restrict__ o0,
                                                        float* __restrict__ o1, float* __restrict__ o2, NleState* st) {
    static_assert(!STATS || MODE == 0, "only stage 1 collects the frame maximum");
    constexpr int NQ = MODE == 0 ? 3 : (MODE == 1 ? 2 : 4);      // running sums per column and plane
    constexpr int NI = MODE == 2 ? 2 : 1;                        // input frames
    constexpr int NL = MODE == 0 ? 3 : 2;                        // loads per input and row: entering, leaving (k), leaving (k2)
    extern __shared__ __attribute__((aligned(16))) double s_v[]; // [NQ][BB][2][BX_VW]   (ROLES: two of them)
    const int role = ROLES ? (int)(threadIdx.x >> 8) : -1;       // (wave-uniform) 0: column waves, 1: task waves
    const int tid = ROLES ? (int)(threadIdx.x & 255) : (int)threadIdx.x;
    const int h = g.h, w = g.w;
    const int k = KT > 0 ? KT : g.k, k2 = MODE == 0 ? (K2T > 0 ? K2T : g.k2) : k;
    const int R = k / 2, R2 = k2 / 2;
    // columns: reflect inside [bx0, bx0 + bw) -- bw = tile_w (SIDD_256 re-tiling) or the whole width
    const int bw = g.tile_w > 0 ? g.tile_w : w;
    const int blk = blockIdx.x / g.nstrip, strip = blockIdx.x % g.nstrip;
    const int bx0 = blk * bw;
    const int ox0 = bx0 + strip * g.ow_nom;
    const int ow = min(g.ow_nom, bx0 + bw - ox0);
    const int vc = min(tid, ow + 2 * R - 1);                     // idle columns repeat the last one (valid addresses, never read back)
    const int rc = bx0 + reflect101(ox0 - R + vc - bx0, bw);
    const int pv = tid + (tid >> 3);                             // the column's slot in an LDS row
    // rows
    const int oy0 = blockIdx.y * g.oh;
    const int ohe = min(g.oh, h - oy0);
    const int nsteps = ohe + 2 * R;
    const int z = blockIdx.z;                                    // Bayer: row parity dy; planar: plane pair
    // the two planes' values of image row gy at this thread's column
    auto ld2 = [&](int i, int gy) -> f32x2 {
        const BoxSrc src = i == 0 ? a : b;
        if (src.bayer) return *(const f32x2*)(src.p + (size_t)(2 * gy + z) * (size_t)(2 * w) + 2 * rc);
        const float* p0 = src.p + ((size_t)(2 * z) * h + gy) * (size_t)w + rc;
        return f32x2{p0[0], p0[(size_t)h * w]};
    };
    // Image rows: three cursors walk the reflected row index one step per local row -- entering row l, leaving rows l - k and
    // l - k2 -- as the PHASE i of BORDER_REFLECT_101's period 2 (h - 1): row = i < h ? i : period - i.  Uniform, i.e. scalar
    // arithmetic; a modulo per load (reflect101) was ~600 scalar instructions per batch and wave, and the CU has ONE scalar
    // unit.  Rows before the segment's start (the "leaving" row of the first k steps) are zeroed by a 0/1 factor when the batch
    // is consumed; rows past its end are rows of the image too and only reach outputs that are never emitted.
    const int period = 2 * (h - 1);
    auto phase0 = [&](int y) -> int {                            // once per cursor
        if (period == 0) return 0;
        int i = y % period;
        return i < 0 ? i + period : i;
    };
    int ph[3] = {phase0(oy0 - R), phase0(oy0 - R - k), phase0(oy0 - R - k2)};
    auto row_next = [&](int c) -> int {
        const int i = ph[c];
        const int y = i < h ? i : period - i;
        ph[c] = (i + 1 >= period) ? 0 : i + 1;
        return y;
    };
    f32x2 cur[NI][NL][BB], nxt[NI][NL][BB];
    const int kl_min = MODE == 0 ? k2 : k;                       // first local row with a leaving row
    const int first_out = MODE == 0 ? R + R2 : 2 * R;            // first local row that completes a window
    auto load_batch = [&](f32x2 (&dst)[NI][NL][BB], int l0) {  // called for l0 = 0, BB, 2 BB, ... in this order
        const bool leave = l0 + BB > kl_min;                   // uniform
#pragma unroll
        for (int r = 0; r < BB; ++r) {
            const int y0 = row_next(0), y1 = row_next(1), y2 = NL == 3 ? row_next(2) : 0;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                dst[i][0][r] = ld2(i, y0);
                if (leave) {
                    dst[i][1][r] = ld2(i, y1);
                    if (NL == 3) dst[i][2][r] = ld2(i, y2);
                }
            }
        }
    };
    double S[NQ][2];
#pragma unroll
    for (int q = 0; q < NQ; ++q) S[q][0] = S[q][1] = 0.0;
    float fmax_ = -INFINITY;
    // task phase: thread = (plane, row of the batch) x chunk; a half-wave holds the 32 chunk slots of one (plane, row)
    const int chunk = tid & 31, t_r = (tid >> 5) % BB, t_pl = (tid >> 5) / BB;
    const int c0 = chunk * BX_C;                                 // first output column of the chunk (strip-relative)
    const bool t_on = c0 < ow && (tid >> 5) < 2 * BB;
    const int nvalid = min(BX_C, ow - c0);
    const int vb_off = (t_r * 2 + t_pl) * BX_VW + chunk * (BX_C + 1);         // + q * BB * 2 * BX_VW ; virtual column c0 + d at d + (d >> 3)
    constexpr int QS = BB * 2 * BX_VW;
    const int t_plane = 2 * z + t_pl;                            // output plane: Bayer (dy = z, dx = t_pl) -> 2 dy + dx; planar: 2 z + t_pl
    const bool vec_ok = (w % 4 == 0) && (ox0 % 4 == 0) && !(((uintptr_t)o0 | (uintptr_t)o1 | (uintptr_t)o2) & 15);
    // the two phases of a batch (l0: its first local row; sv: the LDS buffer of its sums)
    auto col_phase = [&](int l0, double* __restrict__ sv) __attribute__((always_inline)) -> bool {      // true: a warm-up batch, nothing written
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NL; ++j)
#pragma unroll
                for (int r = 0; r < BB; ++r) cur[i][j][r] = nxt[i][j][r];
        if (l0 + BB < nsteps) load_batch(nxt, l0 + BB);
        const bool warm = l0 + BB <= kl_min && l0 + BB <= first_out;     // uniform: no leaving row, no finished window
        if (warm) {
#pragma unroll
            for (int r = 0; r < BB; ++r) {
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const f32x2 xn = cur[i][0][r];
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) {
                        S[2 * i][pl] += (double)xn[pl];
                        S[2 * i + 1][pl] += (double)__fmul_rn(xn[pl], xn[pl]);
                        if (MODE == 0) S[2][pl] += (double)xn[pl];
                    }
                    if (STATS) fmax_ = fmaxf(fmax_, fmaxf(xn[0], xn[1]));
                }
            }
            return true;
        }
#pragma unroll
        for (int r = 0; r < BB; ++r) {
            const float mk = (l0 + r >= k) ? 1.0f : 0.0f, mk2 = (l0 + r >= k2) ? 1.0f : 0.0f;   // leaving row exists
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const f32x2 xn = cur[i][0][r];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    const float xo = __fmul_rn(cur[i][1][r][pl], mk);
                    S[2 * i][pl] += (double)xn[pl] - (double)xo;
                    S[2 * i + 1][pl] += (double)__fmul_rn(xn[pl], xn[pl]) - (double)__fmul_rn(xo, xo);
                    if (MODE == 0) S[2][pl] += (double)xn[pl] - (double)__fmul_rn(cur[0][NL - 1][r][pl], mk2);
                }
                if (STATS) fmax_ = fmaxf(fmax_, fmaxf(xn[0], xn[1]));  // halo, idle columns and clamped rows are pixels of the frame too
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                sv[((q * BB + r) * 2 + 0) * BX_VW + pv] = S[q][0];
                sv[((q * BB + r) * 2 + 1) * BX_VW + pv] = S[q][1];
            }
        }
        return false;
    };
    auto task_phase = [&](int l0, const double* __restrict__ sv) __attribute__((always_inline)) {
        {
            const int l = l0 + t_r;
            const int oy = oy0 + l - 2 * R;                                       // k-window centred here is complete
            const bool emit = t_on && l >= 2 * R && l < nsteps;
            const int oy2 = oy0 + l - R - R2;                                     // MODE 0: k2-window centred here is complete
            const bool emit2 = MODE == 0 && t_on && l >= R + R2 && oy2 < oy0 + ohe;
            // window sums of quantity q over virtual columns [c0 + i + off, c0 + i + off + kk), i = 0 .. 7, by sliding; each sum
            // is handed to `use` as soon as it exists (no array of float64 sums stays live)
            auto slide = [&](int q, int off, int kk, auto&& use) {
                const double* v = sv + vb_off + q * QS;
                // (four partial sums: a chain of 29 dependent float64 additions is ~300 cycles a wave cannot hide;
                // compile-time bounds for the shipped k)
                double wsum = 0.0;
                if constexpr (KT > 0) {
                    double ps[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int d = off; d < off + kk; ++d) ps[(d - off) & 3] += v[d + (d >> 3)];
                    wsum = (ps[0] + ps[1]) + (ps[2] + ps[3]);
                } else {
                    for (int d = off; d < off + kk; ++d) wsum += v[d + (d >> 3)];
                }
                use(0, wsum);
#pragma unroll
                for (int i = 1; i < BX_C; ++i) {
                    const int de = off + kk + i - 1, dl = off + i - 1;
                    wsum += v[de + (de >> 3)] - v[dl + (dl >> 3)];
                    use(i, wsum);
                }
                // one window at a time: with the LDS reads of the next window hoisted above this point the kernel does not fit
                // the 128 registers of four waves per SIMD
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            };
            if (emit) {
                const size_t idx = ((size_t)t_plane * h + oy) * w + ox0 + c0;
                float r0[BX_C], r1[BX_C], r2[BX_C];
                if (MODE == 0) {
                    slide(0, 0, k, [&](int i, double ws) { r0[i] = blur_round(ws, k); });
                    slide(1, 0, k, [&](int i, double ws) {
                        const float sd = std_from(r0[i], blur_round(ws, k));
                        r1[i] = __fmul_rn(sd, sd);                                 // var = lr_rggb_k**2 (YOND_SIDD.py:72)
                    });
                    box_store8(o0 + idx, r0, nvalid, vec_ok);
                    box_store8(o1 + idx, r1, nvalid, vec_ok);
                } else if (MODE == 1) {
                    slide(0, 0, k, [&](int i, double ws) { r1[i] = blur_round(ws, k); });
                    slide(1, 0, k, [&](int i, double ws) { r0[i] = std_from(r1[i], blur_round(ws, k)); });
                    box_store8(o0 + idx, r0, nvalid, vec_ok);
                } else {
                    // the two frames one after the other (a loop the compiler may not unroll): noisy -> sl^2, denoised -> mh, sh
#pragma unroll 1
                    for (int pass = 0; pass < 2; ++pass) {
                        slide(2 * pass, 0, k, [&](int i, double ws) { r0[i] = blur_round(ws, k); });          // mean = blur(hr) (YOND_SIDD.py:97)
                        slide(2 * pass + 1, 0, k, [&](int i, double ws) {
                            const float sd = std_from(r0[i], blur_round(ws, k));
                            const float sq = __fmul_rn(sd, sd);
                            r1[i] = pass == 0 ? sq : __fsub_rn(r1[i], sq);                                     // var = lr_k**2 - hr_k**2 (:96)
                            r2[i] = sd;                                                                        // img_lap = hr_k (:98)
                        });
                    }
                    box_store8(o0 + idx, r0, nvalid, vec_ok);
                    box_store8(o1 + idx, r1, nvalid, vec_ok);
                    box_store8(o2 + idx, r2, nvalid, vec_ok);
                }
            }
            if (emit2) {
                float r2[BX_C];
                slide(2, R - R2, k2, [&](int i, double ws) { r2[i] = blur_round(ws, k2); });
                box_store8(o2 + ((size_t)t_plane * h + oy2) * w + ox0 + c0, r2, nvalid, vec_ok);
            }
        }
    };
    if constexpr (ROLES) {
        constexpr int BUF = NQ * BB * 2 * BX_VW;                 // doubles per buffer
        const int nbatch = (nsteps + BB - 1) / BB;
        if (role == 0) load_batch(nxt, 0);
        for (int i = 0; i <= nbatch; ++i) {
            if (role == 0) {
                if (i < nbatch) col_phase(i * BB, s_v + (i & 1) * BUF);
            } else if (i >= 1) {
                const int l0 = (i - 1) * BB;
                const bool warm = l0 + BB <= kl_min && l0 + BB <= first_out;
                if (!warm) task_phase(l0, s_v + ((i - 1) & 1) * BUF);
            }
            __syncthreads();
        }
    } else {
        load_batch(nxt, 0);
        for (int l0 = 0; l0 < nsteps; l0 += BB) {
            if (col_phase(l0, s_v)) continue;
            __syncthreads();
            task_phase(l0, s_v);
            __syncthreads();
        }
    }
    if constexpr (STATS) {
        if (ROLES && role != 0) return;                           // (the column waves saw the pixels)
        fmax_ = wave_max(fmax_);
        // a few thousand waves, one word: look first (a coherent load) -- after the first few arrivals nearly nobody has to write
        if ((tid & 63) == 0 && fmax_ > -INFINITY) {
            const unsigned int key = f2key(fmax_);
            if (key > __hip_atomic_load(&st->frame_max_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&st->frame_max_key, key);
        }
    }
}

static int box_args_ok(int h, int w, int k, int tile_w) {
    if (h < 1 || w < 1) return YOND_EINVAL;
    if (k < 1 || !(k & 1)) return YOND_EINVAL;
    if (k > 2 * BX_MAXR + 1) return YOND_EUNSUPPORTED;
    if (tile_w < 0) return YOND_EINVAL;
    if (tile_w > 0 && w % tile_w != 0) return YOND_EUNSUPPORTED;
    return YOND_OK;
}

template <int MODE, int KT, int K2T, bool STATS, int BB, bool ROLES = false>
static int launch_box_k(const BoxSrc& a, const BoxSrc& b, const BoxGeom& g, dim3 grid, float* o0, float* o1, float* o2, hipStream_t st,
                        NleState* state) {
    constexpr int NQ = MODE == 0 ? 3 : (MODE == 1 ? 2 : 4);
    constexpr size_t smem = (size_t)NQ * BB * 2 * BX_VW * sizeof(double) * (ROLES ? 2 : 1);
    static bool attr_set = false;
    auto kern = box_slide_kernel<MODE, KT, K2T, STATS, BB, ROLES>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(ROLES ? 512 : BX_T), smem, st, a, b, g, o0, o1, o2, state);
    YOND_LAUNCH_CHECK();
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
    const int maxow = ((BX_T - 2 * (k / 2)) / BX_C) * BX_C;      // whole chunks
    g.nstrip = (bw + maxow - 1) / maxow;
    g.ow_nom = (((bw + g.nstrip - 1) / g.nstrip) + BX_C - 1) / BX_C * BX_C;
    if (g.ow_nom > maxow) g.ow_nom = maxow;
    // row segments: one round of the workgroups the chip holds (two per CU); longer segments re-read fewer halo rows
    const bool roles = MODE != 2 && yond_exp_long("YOND_BOX_ROLES", 0) != 0;      // (experiment builds: the role-split form)
    const long target = yond_exp_long("YOND_BOX_WGS", roles ? (MODE == 1 ? 512 : 256) : (MODE == 1 ? 1024 : 512));
    const long cols = 2L * nblk * g.nstrip;
    long nseg = (target + cols / 2) / cols;
    if (nseg < 1) nseg = 1;
    g.oh = (int)((h + nseg - 1) / nseg);
    if (g.oh < 16) g.oh = 16;
    if (g.oh > h) g.oh = h;
    const int nsy = (h + g.oh - 1) / g.oh;
    dim3 grid((unsigned)(nblk * g.nstrip), (unsigned)nsy, 2);
    // rows per batch: 4 (LDS per workgroup 55 / 37 / 74 KB: two / four / two workgroups per CU).  Two rows per batch and more
    // workgroups measured slower (self stage 1: 101 us against 70): the task phase is latency bound per wave, and half the lanes idle
    constexpr int BB = 4;
#ifdef YOND_EXPERIMENTS
    if constexpr (MODE != 2)
        if (roles && k == 29 && (MODE != 0 || k2 == 19)) return launch_box_k<MODE, 29, 19, STATS, BB, true>(a, b, g, grid, o0, o1, o2, st, state);
#endif
    if (k == 29 && (MODE != 0 || k2 == 19)) return launch_box_k<MODE, 29, 19, STATS, BB>(a, b, g, grid, o0, o1, o2, st, state);
    return launch_box_k<MODE, 0, 0, STATS, BB>(a, b, g, grid, o0, o1, o2, st, state);
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
