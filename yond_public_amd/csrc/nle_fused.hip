// K5': the estimator's local statistics in ONE pass over the frame (YOND_SIDD.py:62-73 / 89-98, utils/isp_algos.py:234-242):
//   self    mean = B29(x), var = stdfilt(x, 29)^2, lap = stdfilt(B19(x), 29)          (k = 29, k2 = k//3*2+1 = 19)
//   collab  mean = B29(hr), var = stdfilt(lr, 29)^2 - stdfilt(hr, 29)^2, lap = stdfilt(hr, 29)
// plus, while the maps are produced, the level-1 histogram of lap, the per-mean-bin minimum of lap (nle_fast.hip: sweep 1
// of the threshold selection) and the frame maximum; the workgroup that finishes last resolves the level-1 bins of the
// percentile ranks.  The B19 map of the self mode never leaves the chip: its rows live in registers / LDS.
//
// cv2.blur semantics (normalised k x k window, BORDER_REFLECT_101), window sums in float64 (float32 data summed in
// float64 is exact, so any summation order gives the same bits), every intermediate rounded to float32 exactly where
// NumPy / OpenCV round; no FMA contraction.
//
// A 512-thread workgroup owns the two planes (dx = 0, 1: its two halves) of one Bayer row parity dy, a strip of <= 256
// virtual columns (outputs + the reflected halo, one column per thread) and a segment of rows, and walks down the rows
// four at a time with TWO barriers per batch:
//   column phase (thread = column)   vertical running sums S += entering - leaving (float64 registers; the leaving row
//       is read again from L2) -> LDS; for the self mode also the vertical sums of the B19 rows (entering from LDS, the
//       leaving one from a 29-deep register ring) -> LDS; stores of the finished output rows (coalesced), histogram /
//       minimum updates;
//   task phase (thread = (plane, row, chunk of 9..17 columns))   horizontal k-window sums by SLIDING along the chunk:
//       two LDS reads and two adds per sum and output after a k-read start -- the 64-lane float64 prefix scans of the
//       first version (6 DPP steps x 3 instructions per sum) were its whole cost (109 + 65 us); results -> LDS staging.
// The three stages (x -> B19 rows -> lap) of consecutive batches run in the same pair of phases (software pipeline).
#include "nle_common.h"

#define BF_T 256             // virtual columns per plane and workgroup
#define BF_B 4               // rows per batch
#define BF_MAXR 14
#define BF_RS 271            // LDS row stride of the vertical sums, in doubles (== 23 * 9 mod 32: chunks of 9 columns of
                             // consecutive rows continue one bank progression)
#define BF_MAXOH 160         // 2 planes * 210 columns * 160 rows < 65536: 16-bit histogram counters cannot overflow
#define BF_LA 9              // chunk length of the 29-window tasks (odd: conflict-free ds_read_b64)
#define BF_LB 17             // chunk length of the 19-window tasks

struct BfGeom {
    int h, w, k, k2, tile_w;
    int ow_nom, nstrip, oh;  // outputs per strip, strips per (tile_w-wide) block, rows per segment
    int W2;                  // Bayer row length (2w)
};

__device__ __forceinline__ float bf_blur_round(double s, double inv) { return (float)(s * inv); }
__device__ __forceinline__ float bf_std_from(float b1, float b2) {          // utils/isp_algos.py:236-241 in float32 steps
    const float d = __fsub_rn(b2, __fmul_rn(b1, b1));
    return __fsqrt_rn(fmaxf(d, 0.0f));
}

// Register rings indexed by a WAVE-UNIFORM position: the switch is a scalar branch and every case names a fixed register
// (a dynamically indexed private array would live in scratch memory).
#define BF_CASES16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define BF_CASES32(X) BF_CASES16(X) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31)
struct BfRing {                       // 29-deep ring of the B19 rows of this thread's column
    float r[32];
    __device__ __forceinline__ float swap(int pos, float in) {
        float out = 0.0f;
#define BF_CASE(i) case i: out = r[i]; r[i] = in; break;
        switch (pos) { BF_CASES32(BF_CASE) }
#undef BF_CASE
        return out;
    }
};
struct BfBins {                       // mean bins of the last 16 rows of this thread's column
    int r[16];
    __device__ __forceinline__ void set(int pos, int v) {
#define BF_CASE(i) case i: r[i] = v; break;
        switch (pos) { BF_CASES16(BF_CASE) }
#undef BF_CASE
    }
    __device__ __forceinline__ int get(int pos) const {
        int out = 0;
#define BF_CASE(i) case i: out = r[i]; break;
        switch (pos) { BF_CASES16(BF_CASE) }
#undef BF_CASE
        return out;
    }
};

// MODE 0: self (a = noisy Bayer frame), MODE 2: collab (a = noisy, b = denoised)
template <int MODE>
__global__ __launch_bounds__(512) void box_fused_kernel(const float* __restrict__ fa, const float* __restrict__ fb, BfGeom g,
                                                       float* __restrict__ o_mean, float* __restrict__ o_var,
                                                       float* __restrict__ o_lap, NleState* st, NfArgs args) {
    constexpr bool SELF = MODE == 0;
    constexpr int NQ1 = SELF ? 3 : 4;                     // vertical sums of stage 1
    constexpr int NQ = SELF ? 5 : 4;                      // all vertical sums
    constexpr int NI = SELF ? 1 : 2;                      // input frames
    constexpr int NL = SELF ? 3 : 2;                      // loads per input and row: entering, leaving (k), leaving (k2)
    extern __shared__ unsigned char s_raw[];
    double* s_v = (double*)s_raw;                                             // [NQ][2][BF_B][BF_RS]
    float* s_st = (float*)(s_raw + (size_t)NQ * 2 * BF_B * BF_RS * 8);        // staging [4][2][BF_B][BF_T]: mean, var, b19, lap
    unsigned int* s_h = (unsigned int*)(s_st + 4 * 2 * BF_B * BF_T);          // [NF_WIN_N / 2] two 16-bit counters per word
    unsigned int* s_mi = s_h + NF_WIN_N / 2;                                  // [NF_BINS]
    int* s_row = (int*)(s_mi + NF_BINS);                                      // [BF_MAXOH + 2 * (2 * BF_MAXR)]
    auto V = [&](int q, int half, int r) -> double* { return s_v + ((size_t)(q * 2 + half) * BF_B + r) * BF_RS; };
    auto ST = [&](int m, int half, int r) -> float* { return s_st + ((size_t)(m * 2 + half) * BF_B + r) * BF_T; };

    const int tid = threadIdx.x, half = tid >> 8, col = tid & 255;
    const int dy = blockIdx.z, plane = 2 * dy + half;
    const int h = g.h, w = g.w, k = g.k, k2 = g.k2;
    const int R = k / 2, R2 = SELF ? k2 / 2 : 0;
    const int HALO = SELF ? R + R2 : R;                                       // self: x -> b19 (R2) -> lap (R)
    const double inv_k = 1.0 / (double)(k * k), inv_k2 = 1.0 / (double)(k2 * k2);
    // columns: reflect inside [bx0, bx0 + bw) -- bw = tile_w (SIDD_256 re-tiling) or the whole width
    const int bw = g.tile_w > 0 ? g.tile_w : w;
    const int blk = blockIdx.x / g.nstrip, strip = blockIdx.x % g.nstrip;
    const int bx0 = blk * bw;
    const int ox0 = bx0 + strip * g.ow_nom;
    const int ow = min(g.ow_nom, bx0 + bw - ox0);
    const int rc = bx0 + reflect101(ox0 - HALO + col - bx0, bw);              // image column of this virtual column
    const bool writer = col >= HALO && col < HALO + ow;
    const int ox = ox0 + col - HALO;
    // rows
    const int oy0 = blockIdx.y * g.oh;
    const int ohe = min(g.oh, h - oy0);
    const int nsteps = ohe + 2 * HALO;
    for (int i = tid; i < NF_WIN_N / 2 + NF_BINS; i += 512) s_h[i] = 0;
    for (int l = tid; l < nsteps; l += 512) s_row[l] = reflect101(oy0 - HALO + l, h);
    __syncthreads();
    const float* base[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) base[i] = (i == 0 ? fa : fb) + (size_t)dy * g.W2 + 2 * rc + half;
    const size_t rstride = (size_t)2 * g.W2;
    auto ld = [&](int i, int l) -> float {
        const int gy = s_row[min(max(l, 0), nsteps - 1)];
        return base[i][(size_t)gy * rstride];
    };
    float cur[NI][NL][BF_B], nxt[NI][NL][BF_B];
    auto load_batch = [&](float (&dst)[NI][NL][BF_B], int l0) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
#pragma unroll
            for (int r = 0; r < BF_B; ++r) {
                dst[i][0][r] = ld(i, l0 + r);
                dst[i][1][r] = ld(i, l0 + r - k);
                if (NL == 3) dst[i][2][r] = ld(i, l0 + r - k2);
            }
        }
    };
    double S[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) S[q] = 0.0;
    BfRing ring;
#pragma unroll
    for (int i = 0; i < 32; ++i) ring.r[i] = 0.0f;
    BfBins rbin;                                                              // self: the lap row lags its mean row by R2 rows
#pragma unroll
    for (int i = 0; i < 16; ++i) rbin.r[i] = 0;
    unsigned int run_id = 0, run_cnt = 0, cmax = 0;
    int cbin = -1;
    float fmax_ = -INFINITY;
    auto hist_flush = [&]() {
        if (!run_cnt) return;
        const unsigned int wdw = run_id - NF_WIN_LO;
        if (wdw < NF_WIN_N) atomicAdd(&s_h[wdw >> 1], run_cnt << ((wdw & 1u) * 16));
        else atomicAdd(&st->hist1[run_id], run_cnt);
    };
    auto stats = [&](float lapv, int bin) {
        const unsigned int key = f2key(lapv);
        const unsigned int id = key >> 16;
        if (id != run_id) { hist_flush(); run_id = id; run_cnt = 0; }
        run_cnt += 1;
        if (bin != cbin) { cbin = bin; cmax = 0; }
        const unsigned int inv = ~key;
        if (inv > cmax) { cmax = inv; atomicMax(&s_mi[bin], inv); }
    };

    // ---- task tables (task phase): groups A (mean, var / collab: all three), C (lap, self only), B (b19, self only) ----
    const int nA = (ow + BF_LA - 1) / BF_LA;                                  // chunks per row of the 29-window groups
    const int nBc = SELF ? (ow + 2 * R + BF_LB - 1) / BF_LB : 0;              // chunks per row of the 19-window group
    const int tasksA = 2 * BF_B * nA;                                         // (half, row, chunk)
    const int baseC = (tasksA + 63) & ~63, tasksC = SELF ? tasksA : 0;
    const int baseB = SELF ? ((baseC + tasksC + 63) & ~63) : baseC, tasksB = 2 * BF_B * nBc;
    const int ntask_threads = baseB + tasksB;                                 // <= 512 for ow <= 210 (host checks)

    load_batch(nxt, 0);
    const int nbatch = (nsteps + BF_B - 1) / BF_B;
    // batch nb handles entering rows l0 .. l0 + 3.  Column phase of iteration nb: stage-1 sums of batch nb, stage-2 sums
    // of batch nb - 1 (its b19 rows were finished by the task phase of iteration nb - 1), stores of what the previous
    // task phase finished.  One extra iteration drains the pipeline.
    for (int nb = 0; nb <= nbatch + (SELF ? 1 : 0); ++nb) {
        const int l0 = nb * BF_B;
        // ================= column phase =================
        // (a) stores + statistics of the rows the previous task phase finished
        // (first the lap rows: they read the bins of mean rows stored up to the previous iteration -- the ring is 16 deep)
        if (SELF && nb > 1 && writer) {
            const int lp = l0 - 2 * BF_B;                                     // lap rows of the batch before the previous one
#pragma unroll
            for (int r = 0; r < BF_B; ++r) {
                const int l = lp + r;
                const int cl = l - R2 - R;                                    // lap row: x row l -> b19 row l - R2 -> lap row - R
                if (cl >= HALO && cl < HALO + ohe && l < nsteps) {
                    const size_t idx = ((size_t)plane * h + (oy0 + cl - HALO)) * w + ox;
                    const float lv = ST(3, half, r)[col];
                    o_lap[idx] = lv;
                    stats(lv, rbin.get(cl & 15));
                }
            }
        }
        if (nb > 0 && writer) {
            const int lp = l0 - BF_B;                                         // entering rows of the previous batch
#pragma unroll
            for (int r = 0; r < BF_B; ++r) {
                const int l = lp + r;
                if (SELF) {
                    const int cm = l - R;                                     // mean / var row (k-window centred there)
                    if (cm >= HALO && cm < HALO + ohe && l < nsteps) {
                        const size_t idx = ((size_t)plane * h + (oy0 + cm - HALO)) * w + ox;
                        const float m = ST(0, half, r)[col];
                        o_mean[idx] = m;
                        o_var[idx] = ST(1, half, r)[col];
                        rbin.set(cm & 15, (int)__fmul_rn(fminf(fmaxf(m, 0.0f), 1.0f), 1000.0f));   // (mean.clip(0,1)*nbins).astype(int)
                    }
                } else {
                    const int cm = l - R;
                    if (cm >= HALO && cm < HALO + ohe && l < nsteps) {
                        const size_t idx = ((size_t)plane * h + (oy0 + cm - HALO)) * w + ox;
                        const float m = ST(0, half, r)[col], lv = ST(3, half, r)[col];
                        o_mean[idx] = m;
                        o_var[idx] = ST(1, half, r)[col];
                        o_lap[idx] = lv;
                        stats(lv, (int)__fmul_rn(fminf(fmaxf(m, 0.0f), 1.0f), 1000.0f));
                    }
                }
            }
        }
        // (b) stage 2 (self): vertical sums of the b19 rows finished by the previous task phase
        if (SELF && nb > 0 && nb <= nbatch) {
            const int lp = l0 - BF_B;
#pragma unroll
            for (int r = 0; r < BF_B; ++r) {
                const int l = lp + r;
                const int j = l - R2;                                         // b19 row index (k2-window centred there)
                if (l >= 2 * R2 && l < nsteps) {                              // first complete k2-window: rows 0 .. 2 R2
                    const float bn = ST(2, half, r)[col];
                    const float bo = ring.swap(j % 29, bn);                  // b19 row j - 29 (0 until the ring has filled)
                    S[3] += (double)bn - (double)bo;
                    S[4] += (double)__fmul_rn(bn, bn) - (double)__fmul_rn(bo, bo);
                }
                V(3, half, r)[col] = S[3];
                V(4, half, r)[col] = S[4];
            }
        }
        // (c) stage 1: vertical sums of the entering rows of this batch
        if (nb < nbatch) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
#pragma unroll
                for (int j = 0; j < NL; ++j) {
#pragma unroll
                    for (int r = 0; r < BF_B; ++r) cur[i][j][r] = nxt[i][j][r];
                }
            }
            if (l0 + BF_B < nsteps) load_batch(nxt, l0 + BF_B);
#pragma unroll
            for (int r = 0; r < BF_B; ++r) {
                const int l = l0 + r;
                const float mk = (l >= k) ? 1.0f : 0.0f, mk2 = (l >= k2) ? 1.0f : 0.0f;       // leaving row exists
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const float xn = cur[i][0][r], xo = __fmul_rn(cur[i][1][r], mk);
                    S[2 * i] += (double)xn - (double)xo;
                    S[2 * i + 1] += (double)__fmul_rn(xn, xn) - (double)__fmul_rn(xo, xo);
                }
                if (SELF) S[2] += (double)cur[0][0][r] - (double)__fmul_rn(cur[0][2][r], mk2);
#pragma unroll
                for (int q = 0; q < NQ1; ++q) V(q, half, r)[col] = S[q];
                if (writer && l >= HALO && l < HALO + ohe) fmax_ = fmaxf(fmax_, cur[0][0][r]);   // the frame's own pixels
            }
        }
        __syncthreads();
        // ================= task phase =================
        if (tid < ntask_threads) {
            if (tid < tasksA && nb < nbatch) {
                // group A: 29-window sums of stage 1 -> mean, var (collab: + lap)
                const int t = tid;
                const int chunk = t % nA, rr = (t / nA) % BF_B, hf = t / (nA * BF_B);
                const int l = l0 + rr;
                const int cm = l - R;
                if (cm >= HALO && cm < HALO + ohe && l < nsteps) {
                    const int c0 = HALO + chunk * BF_LA, c1 = min(c0 + BF_LA, HALO + ow);
                    const double* v0 = V(0, hf, rr);
                    const double* v1 = V(1, hf, rr);
                    const double* v2 = V(SELF ? 0 : 2, hf, rr);
                    const double* v3 = V(SELF ? 1 : 3, hf, rr);
                    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
                    for (int i = c0 - R; i <= c0 + R; ++i) {
                        a0 += v0[i]; a1 += v1[i];
                        if (!SELF) { a2 += v2[i]; a3 += v3[i]; }
                    }
                    for (int c = c0; c < c1; ++c) {
                        if (c > c0) {
                            a0 += v0[c + R] - v0[c - R - 1]; a1 += v1[c + R] - v1[c - R - 1];
                            if (!SELF) { a2 += v2[c + R] - v2[c - R - 1]; a3 += v3[c + R] - v3[c - R - 1]; }
                        }
                        if (SELF) {
                            const float m = bf_blur_round(a0, inv_k);
                            const float sd = bf_std_from(m, bf_blur_round(a1, inv_k));
                            ST(0, hf, rr)[c] = m;
                            ST(1, hf, rr)[c] = __fmul_rn(sd, sd);                         // var = lr_rggb_k**2 (YOND_SIDD.py:72)
                        } else {
                            const float sl = bf_std_from(bf_blur_round(a0, inv_k), bf_blur_round(a1, inv_k));
                            const float mh = bf_blur_round(a2, inv_k);
                            const float sh = bf_std_from(mh, bf_blur_round(a3, inv_k));
                            ST(0, hf, rr)[c] = mh;                                        // mean = blur(hr) (YOND_SIDD.py:97)
                            ST(1, hf, rr)[c] = __fsub_rn(__fmul_rn(sl, sl), __fmul_rn(sh, sh));   // var = lr_k**2 - hr_k**2 (:96)
                            ST(3, hf, rr)[c] = sh;                                        // img_lap = hr_k (:98)
                        }
                    }
                }
            } else if (SELF && tid >= baseC && tid < baseC + tasksC && nb > 0 && nb <= nbatch) {
                // group C: 29-window sums of stage 2 (b19 rows of the previous batch) -> lap
                const int t = tid - baseC;
                const int chunk = t % nA, rr = (t / nA) % BF_B, hf = t / (nA * BF_B);
                const int l = l0 - BF_B + rr;
                const int cl = l - R2 - R;
                if (cl >= HALO && cl < HALO + ohe && l < nsteps) {
                    const int c0 = HALO + chunk * BF_LA, c1 = min(c0 + BF_LA, HALO + ow);
                    const double* v0 = V(3, hf, rr);
                    const double* v1 = V(4, hf, rr);
                    double a0 = 0.0, a1 = 0.0;
                    for (int i = c0 - R; i <= c0 + R; ++i) { a0 += v0[i]; a1 += v1[i]; }
                    for (int c = c0; c < c1; ++c) {
                        if (c > c0) { a0 += v0[c + R] - v0[c - R - 1]; a1 += v1[c + R] - v1[c - R - 1]; }
                        ST(3, hf, rr)[c] = bf_std_from(bf_blur_round(a0, inv_k), bf_blur_round(a1, inv_k));
                    }
                }
            } else if (SELF && tid >= baseB && nb < nbatch) {
                // group B: 19-window sums -> b19 rows (columns HALO - R .. HALO + ow + R)
                const int t = tid - baseB;
                const int chunk = t % nBc, rr = (t / nBc) % BF_B, hf = t / (nBc * BF_B);
                const int l = l0 + rr;
                if (l >= 2 * R2 && l < nsteps) {
                    const int c0 = HALO - R + chunk * BF_LB, c1 = min(c0 + BF_LB, HALO + ow + R);
                    const double* v0 = V(2, hf, rr);
                    double a0 = 0.0;
                    for (int i = c0 - R2; i <= c0 + R2; ++i) a0 += v0[i];
                    for (int c = c0; c < c1; ++c) {
                        if (c > c0) a0 += v0[c + R2] - v0[c - R2 - 1];
                        ST(2, hf, rr)[c] = bf_blur_round(a0, inv_k2);
                    }
                }
            }
        }
        __syncthreads();
    }
    hist_flush();
    // frame maximum (the estimator's caller needs lr.max() for the bias LUT grid, YOND_SIDD.py:256/393)
    fmax_ = wave_max(fmax_);
    if ((tid & 63) == 0 && fmax_ > -INFINITY) atomicMax(&st->frame_max_key, f2key(fmax_));
    __syncthreads();
    for (int i = tid; i < NF_WIN_N / 2; i += 512) {
        const unsigned int c = s_h[i];
        if (c & 0xFFFFu) atomicAdd(&st->hist1[NF_WIN_LO + 2 * i], c & 0xFFFFu);
        if (c >> 16) atomicAdd(&st->hist1[NF_WIN_LO + 2 * i + 1], c >> 16);
    }
    for (int i = tid; i < NF_BINS; i += 512) {
        const unsigned int v = s_mi[i];
        if (v) atomicMax(&st->maxinv[i], v);
    }
    const unsigned int nblocks = gridDim.x * gridDim.y * gridDim.z;
    if (nf_arrive_last(&st->ticket[0], nblocks)) nf_resolve1(st, args, (unsigned int*)s_raw);
}

template <int MODE>
static int launch_fused(const float* fa, const float* fb, int H, int W, int k, int k2, int tile_w, float* mean, float* var,
                        float* lap, const double* q_host, int nq, void* ws, hipStream_t st) {
    const int h = H / 2, w = W / 2;
    if (h < 1 || w < 1 || k < 1 || !(k & 1) || k2 < 1 || !(k2 & 1) || tile_w < 0) return YOND_EINVAL;
    if (k > 2 * BF_MAXR + 1 || k2 > k) return YOND_EUNSUPPORTED;
    if (tile_w > 0 && w % tile_w != 0) return YOND_EUNSUPPORTED;
    if (!ws || ((uintptr_t)ws & 15)) return YOND_EINVAL;
    const size_t n = (size_t)4 * h * w;
    if (n > 0xFFFFFFFFull) return YOND_EUNSUPPORTED;
    NfArgs a;
    int rc = nf_make_args(n, q_host, nq, &a);
    if (rc) return rc;
    BfGeom g;
    g.h = h; g.w = w; g.k = k; g.k2 = k2; g.tile_w = tile_w; g.W2 = W;
    const int halo = MODE == 0 ? k / 2 + k2 / 2 : k / 2;
    const int bw = tile_w > 0 ? tile_w : w;
    const int nblk = w / bw;
    const int maxow = BF_T - 2 * halo;
    g.nstrip = (bw + maxow - 1) / maxow;
    g.ow_nom = (bw + g.nstrip - 1) / g.nstrip;
    // row segments: one workgroup per CU in one round (the kernel needs most of the LDS); longer segments re-read
    // fewer halo rows
    long target = 240;
    if (const char* e = getenv("YOND_BOX_WGS")) target = atol(e);        // experiments only
    const long cols = 2L * nblk * g.nstrip;
    long nseg = (target + cols / 2) / cols;
    if (nseg < 1) nseg = 1;
    g.oh = (int)((h + nseg - 1) / nseg);
    if (g.oh < 16) g.oh = 16;
    if (g.oh > BF_MAXOH) g.oh = BF_MAXOH;
    if (g.oh > 65535 / (2 * g.ow_nom)) g.oh = 65535 / (2 * g.ow_nom);     // 16-bit LDS histogram counters: pixels per workgroup < 2^16
    if (g.oh > h) g.oh = h;
    const int nsy = (h + g.oh - 1) / g.oh;
    constexpr int NQ = MODE == 0 ? 5 : 4;
    const size_t lds = (size_t)NQ * 2 * BF_B * BF_RS * 8 + (size_t)4 * 2 * BF_B * BF_T * 4 + (NF_WIN_N / 2 + NF_BINS) * 4 +
                       (BF_MAXOH + 4 * BF_MAXR + 8) * 4;
    static bool attr[3] = {false, false, false};
    if (!attr[MODE]) {
        hipError_t e = hipFuncSetAttribute((const void*)box_fused_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr[MODE] = true;
    }
    hipError_t e = hipMemsetAsync(ws, 0, nf_state_bytes(), st);
    if (e != hipSuccess) return (int)e;
    dim3 grid((unsigned)(nblk * g.nstrip), (unsigned)nsy, 2);
    hipLaunchKernelGGL(box_fused_kernel<MODE>, grid, dim3(512), lds, st, fa, fb, g, mean, var, lap, (NleState*)ws, a);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

extern "C" int yond_box_stats_self_fused_f32(const float* bayer, int H, int W, int k, int k2, int tile_w, float* mean,
                                             float* var, float* lap, const double* q_host, int nq, void* ws, void* stream) {
    if (!bayer || !mean || !var || !lap || (H & 1) || (W & 1)) return YOND_EINVAL;
    return launch_fused<0>(bayer, nullptr, H, W, k, k2, tile_w, mean, var, lap, q_host, nq, ws, (hipStream_t)stream);
}

extern "C" int yond_box_stats_collab_fused_f32(const float* bayer_lr, const float* bayer_hr, int H, int W, int k, int tile_w,
                                               float* mean, float* var, float* lap, const double* q_host, int nq, void* ws,
                                               void* stream) {
    if (!bayer_lr || !bayer_hr || !mean || !var || !lap || (H & 1) || (W & 1)) return YOND_EINVAL;
    return launch_fused<2>(bayer_lr, bayer_hr, H, W, k, k, tile_w, mean, var, lap, q_host, nq, ws, (hipStream_t)stream);
}
