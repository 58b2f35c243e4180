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
//       leaving one from a 32-deep register ring) -> LDS; stores of the finished output rows (16 bytes per lane), one
//       LDS atomic per pixel for the level-1 histogram, read-then-atomic-max for the per-bin minimum;
//   task phase (thread = (plane, row, chunk of 9..17 columns))   horizontal k-window sums by SLIDING along the chunk:
//       two LDS reads and two adds per sum and output after a k-read start -- the 64-lane float64 prefix scans of the
//       first version (6 DPP steps x 3 instructions per sum) were its whole cost (109 + 65 us); results -> LDS staging.
// The three stages (x -> B19 rows -> lap) of consecutive batches run in the same pair of phases (software pipeline).
#ifdef YOND_EXPERIMENTS     // (built into experiment libraries only: see include/yond_hip_experiments.h)
#include "../../include/yond_hip_experiments.h"
#include "nle_common.h"

#define BF_T 256             // virtual columns per plane and workgroup
#define BF_B 4               // rows per batch (2, with two workgroups per CU, measured slower: 364 vs 297 us -- more barriers, spills)
#define BFW_LO 0xB080u       // LDS histogram window of this kernel: level-1 bins of 2^-30 <= lap < 4 (32 octaves x 128)
#define BFW_N 4096           //   exact zeros (flat clipped regions) count in a word of their own, anything else goes to global memory
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

// Dispatch on a WAVE-UNIFORM value: a scalar branch; every case sees its value as a constant, so the register rings
// below are indexed by constants (a dynamically indexed private array would live in scratch memory).
#define BF_CASES16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define BF_CASES32(X) BF_CASES16(X) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31)
#define BF_CASE(i) case i: f(IntC<i>{}); break;
template <class F>
__device__ __forceinline__ void uniform_switch16(int v, F&& f) { switch (v) { BF_CASES16(BF_CASE) } }
template <class F>
__device__ __forceinline__ void uniform_switch32(int v, F&& f) { switch (v) { BF_CASES32(BF_CASE) } }
#undef BF_CASE

template <int N>
struct BfPtrs { const double* p[N]; };

// MODE 0: self (a = noisy Bayer frame), MODE 2: collab (a = noisy, b = denoised)
template <int MODE, int K, int K2>
__global__ __launch_bounds__(512) void box_fused_kernel(const float* __restrict__ fa, const float* __restrict__ fb, BfGeom g,
                                                       float* __restrict__ o_mean, float* __restrict__ o_var,
                                                       float* __restrict__ o_lap, NleState* st, NfArgs args) {
    constexpr bool SELF = MODE == 0;
    constexpr int NQ1 = SELF ? 3 : 4;                     // vertical sums of stage 1
    constexpr int NQ = SELF ? 5 : 4;                      // all vertical sums
    constexpr int NI = SELF ? 1 : 2;                      // input frames
    constexpr int NL = SELF ? 3 : 2;                      // loads per input and row: entering, leaving (k), leaving (k2)
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];    // (unaligned, its float64 accesses crawl)
    double* s_v = (double*)s_raw;                                             // [NQ][2][BF_B][BF_RS]
    float* s_st = (float*)(s_raw + (size_t)NQ * 2 * BF_B * BF_RS * 8);        // staging [4][2][BF_B][BF_T]: mean, var, b19, lap
    unsigned int* s_h = (unsigned int*)(s_st + 4 * 2 * BF_B * BF_T);          // [BFW_N / 2] two 16-bit counters per word, [1] zeros
    unsigned int* s_mi = s_h + BFW_N / 2 + 4;                                 // [NF_BINS]
    auto V = [&](int q, int half, int r) -> double* { return s_v + ((size_t)(q * 2 + half) * BF_B + r) * BF_RS; };
    // staging rows are indexed from the first column that is written: outputs start at HALO (b19 rows: at HALO - R), so
    // that the 16-byte rows of the store phase are aligned
    auto ST = [&](int m, int half, int r) -> float* { return s_st + ((size_t)(m * 2 + half) * BF_B + r) * BF_T; };

    const int tid = threadIdx.x, half = tid >> 8, col = tid & 255, lane = tid & 63;
    const int dy = blockIdx.z, plane = 2 * dy + half;
    const int h = g.h, w = g.w;
    constexpr int k = K, k2 = K2;                                             // window sizes are template constants: the window
    constexpr int R = K / 2, R2 = SELF ? K2 / 2 : 0;                          // loops unroll into reads at immediate offsets
    constexpr int HALO = SELF ? R + R2 : R;                                   // self: x -> b19 (R2) -> lap (R)
    const double inv_k = 1.0 / (double)(k * k), inv_k2 = 1.0 / (double)(k2 * k2);
    // columns: reflect inside [bx0, bx0 + bw) -- bw = tile_w (SIDD_256 re-tiling) or the whole width
    const int bw = g.tile_w > 0 ? g.tile_w : w;
    const int blk = blockIdx.x / g.nstrip, strip = blockIdx.x % g.nstrip;
    const int bx0 = blk * bw;
    const int ox0 = bx0 + strip * g.ow_nom;
    const int ow = min(g.ow_nom, bx0 + bw - ox0);
    const int rc = bx0 + reflect101(ox0 - HALO + col - bx0, bw);              // image column of this virtual column
    const bool writer = col >= HALO && col < HALO + ow;
    const bool vec_ok = !((w | ox0 | ow) & 3);                                // 16-byte rows: stores as float4
    // rows
    const int oy0 = blockIdx.y * g.oh;
    const int ohe = min(g.oh, h - oy0);
    const int nsteps = ohe + 2 * HALO;
    for (int i = tid; i < BFW_N / 2 + 4 + NF_BINS; i += 512) s_h[i] = 0;
    const float* base[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) base[i] = (i == 0 ? fa : fb) + (size_t)dy * g.W2 + 2 * rc + half;
    const size_t rstride = (size_t)2 * g.W2;
    auto ld = [&](int i, int l) -> float {                                    // l is uniform: the row address is scalar arithmetic
        int gy = oy0 - HALO + min(max(l, 0), nsteps - 1);
        if (gy < 0 || gy >= h) {                                              // (uniform) one reflection without the modulo; else the general map
            const int g1 = gy < 0 ? -gy : 2 * (h - 1) - gy;
            gy = (g1 >= 0 && g1 < h) ? g1 : reflect101(gy, h);
        }
        return base[i][(size_t)gy * rstride];
    };
    float cur[NI][NL][BF_B], nxt[NI][NL][BF_B];
    auto load_batch = [&](float (&dst)[NI][NL][BF_B], int l0) {
        const int y0 = oy0 - HALO + l0;
        if (l0 >= k && l0 + BF_B <= nsteps && y0 - k >= 0 && y0 + BF_B <= h) {   // (uniform) interior batch: no clamp, no reflection
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const float* p = base[i] + (size_t)y0 * rstride;
#pragma unroll
                for (int r = 0; r < BF_B; ++r) {
                    dst[i][0][r] = p[(size_t)r * rstride];
                    dst[i][1][r] = (p - (size_t)k * rstride)[(size_t)r * rstride];
                    if (NL == 3) dst[i][2][r] = (p - (size_t)k2 * rstride)[(size_t)r * rstride];
                }
            }
            return;
        }
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
    float ring[32];                                                           // self: the last 32 b19 rows of this column (slot = row & 31)
#pragma unroll
    for (int i = 0; i < 32; ++i) ring[i] = 0.0f;
    int rbin[16];                                                             // self: mean bins of the last 16 rows (the lap row lags by R2)
#pragma unroll
    for (int i = 0; i < 16; ++i) rbin[i] = 0;
    float fmax_ = -INFINITY;
    unsigned int bmask_ = 0;                                                  // blocks of 4096 level-1 bins this lane added to in global memory
    // statistics of one finished row (all lanes of the wave take part): level-1 histogram with runs of equal bins across
    // the lanes (adjacent columns of the smooth map) merged into one LDS atomic; per mean bin the smallest lap
    // (the four rows of a batch go through each step together, so that their LDS round trips overlap)
    auto stats_rows = [&](const bool (&rowv)[BF_B], const float (&lapv)[BF_B], const int (&bin)[BF_B]) {
        unsigned int key[BF_B], known[BF_B];
        bool valid[BF_B];
#pragma unroll
        for (int r = 0; r < BF_B; ++r) {
            valid[r] = rowv[r] && writer;
            key[r] = f2key(lapv[r]);
            known[r] = s_mi[valid[r] ? bin[r] : 0];                           // what the workgroup already knows about the bin
        }
#pragma unroll
        for (int r = 0; r < BF_B; ++r) {
            if (valid[r]) {
                // one LDS atomic per pixel: neighbouring pixels hit neighbouring bins (two 16-bit counters per word: bins w
                // and w + 2048 share one, so neighbours sit in different words / banks); merging runs of equal bins across
                // the lanes first cost ~30 vector instructions per row, more than the few same-address replays it saved
                const unsigned int id = key[r] >> 16, wdw = id - BFW_LO;
                if (wdw < BFW_N) atomicAdd(&s_h[wdw & (BFW_N / 2 - 1)], 1u << ((wdw >> 11) * 16));
                else if (id == NF_WIN_LO) atomicAdd(&s_h[BFW_N / 2], 1u);                // lap == +0.0
                else { atomicAdd(&st->hist1[nf_hpos(id)], 1u); bmask_ |= 1u << (id >> 12); }
                const unsigned int inv = ~key[r];
                if (inv > known[r]) atomicMax(&s_mi[bin[r]], inv);
            }
        }
    };
    auto bin_of = [](float m) -> int { return (int)__fmul_rn(fminf(fmaxf(m, 0.0f), 1.0f), 1000.0f); };   // (mean.clip(0,1)*nbins).astype(int)

    // ---- task tables (task phase): groups A (mean, var / collab: all three), C (lap, self only), B (b19, self only) ----
    const int nA = (ow + BF_LA - 1) / BF_LA;                                  // chunks per row of the 29-window groups
    const int nBc = SELF ? (ow + 2 * R + BF_LB - 1) / BF_LB : 0;              // chunks per row of the 19-window group
    const int tasksA = 2 * BF_B * nA;                                         // (half, row, chunk)
    const int baseC = (tasksA + 63) & ~63, tasksC = SELF ? tasksA : 0;
    const int baseB = SELF ? ((baseC + tasksC + 63) & ~63) : baseC, tasksB = 2 * BF_B * nBc;
    const int ntask_threads = baseB + tasksB;                                 // <= 512 for ow <= 210 (host checks)
    const int n4 = (ow + 3) >> 2;
    // this thread's task of the task phase, decoded ONCE (the divisions by run-time chunk counts cost ~100 instructions)
    int tk_chunk, tk_rr, tk_hf;
    {
        const int grp = tid < tasksA ? 0 : ((SELF && tid >= baseC && tid < baseC + tasksC) ? 1 : 2);
        const int t = grp == 0 ? tid : (grp == 1 ? tid - baseC : tid - baseB);
        const int nch = grp == 2 ? max(nBc, 1) : max(nA, 1);
        tk_chunk = t % nch; tk_rr = (t / nch) % BF_B; tk_hf = t / (nch * BF_B);
    }

    __syncthreads();
    load_batch(nxt, 0);
    const int nbatch = (nsteps + BF_B - 1) / BF_B;
#ifdef BF_STAMPS
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#define BF_STAMP(i) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); tacc[i] += tn_ - tprev; tprev = tn_; }
#else
#define BF_STAMP(i)
#endif
    // Sliding k-window sums of one chunk: NS sums over rows v[0..NS), first output column c0 (index into the rows), LEN
    // outputs (only those below c1 are emitted), window radius rad = kk / 2.  All LDS reads are issued ahead of their use.
    auto slide = [&](auto ns_c, auto len_c, auto kk_c, auto vp, int c0, int c1, auto&& emit) {
        constexpr int NS = decltype(ns_c)::value, LEN = decltype(len_c)::value, KK = decltype(kk_c)::value, RAD = KK / 2;
        const double* v[NS];
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) v[s2] = vp.p[s2] + c0 - RAD;          // every read below is at an immediate offset
        double a[NS];
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) {                                     // window of the chunk's first output
            double t0 = 0.0, t1 = 0.0;                                        // (two chains; float64 sums of float32 data are exact)
#pragma unroll
            for (int i = 0; i + 1 < KK; i += 2) { t0 += v[s2][i]; t1 += v[s2][i + 1]; }
            a[s2] = (t0 + t1) + v[s2][KK - 1];
        }
        emit(c0, a);
#pragma unroll
        for (int u = 0; u < LEN - 1; ++u) {
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2) a[s2] += v[s2][KK + u] - v[s2][u];
            if (c0 + 1 + u < c1) emit(c0 + 1 + u, a);
        }
    };

    // batch nb handles entering rows l0 .. l0 + 3.  Column phase of iteration nb: stage-1 sums of batch nb, stage-2 sums
    // of batch nb - 1 (its b19 rows were finished by the task phase of iteration nb - 1), stores of what the previous
    // task phase finished.  One extra iteration drains the pipeline.
    for (int nb = 0; nb <= nbatch + (SELF ? 1 : 0); ++nb) {
        const int l0 = nb * BF_B;
        // ================= column phase =================
        // (c) stage 1: vertical sums of the entering rows of this batch (first: its loads are waited for before this
        // iteration's stores are queued behind them)
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
        BF_STAMP(0)
        // (b) stage 2 (self): vertical sums of the b19 rows finished by the previous task phase
        if (SELF && nb > 0 && nb <= nbatch) {
            const int lp = l0 - BF_B;
            const int j0 = lp - R2;                                           // b19 row of x row lp (k2-window centred there)
            const bool bcol = col >= HALO - R && col < HALO + ow + R;
            float bn[BF_B], bo[BF_B];
            bool val[BF_B];
#pragma unroll
            for (int r = 0; r < BF_B; ++r) {
                val[r] = lp + r >= 2 * R2 && lp + r < nsteps;                 // first complete k2-window: rows 0 .. 2 R2
                bn[r] = ST(2, half, r)[bcol ? col - (HALO - R) : 0];
                bo[r] = 0.0f;
            }
            // slot of b19 row j: j & 31; the row that leaves the 29-row window, j - 29, sits in slot (j + 3) & 31
            uniform_switch32((j0 + 64) & 31, [&](auto bc) {
                constexpr int B0 = decltype(bc)::value;
#pragma unroll
                for (int r = 0; r < BF_B; ++r) bo[r] = ring[(B0 + r + 3) & 31];
#pragma unroll
                for (int r = 0; r < BF_B; ++r) ring[(B0 + r) & 31] = val[r] ? bn[r] : ring[(B0 + r) & 31];
            });
#pragma unroll
            for (int r = 0; r < BF_B; ++r) {
                if (val[r]) {
                    S[3] += (double)bn[r] - (double)bo[r];
                    S[4] += (double)__fmul_rn(bn[r], bn[r]) - (double)__fmul_rn(bo[r], bo[r]);
                }
                V(3, half, r)[col] = S[3];
                V(4, half, r)[col] = S[4];
            }
        }
        BF_STAMP(1)
        // (a) statistics of the finished lap rows (thread = column): self: rows of batch nb - 2, collab: batch nb - 1
        {
            const int lp = l0 - (SELF ? 2 : 1) * BF_B;
            if (lp >= 0) {
                const int cl0 = lp - R2 - R;                                  // lap row of x row lp (x row l -> b19 row l - R2 -> lap row - R)
                int bins[BF_B];
#pragma unroll
                for (int r = 0; r < BF_B; ++r) bins[r] = 0;
                if (SELF) {
                    uniform_switch16(cl0 & 15, [&](auto bc) {
                        constexpr int B0 = decltype(bc)::value;
#pragma unroll
                        for (int r = 0; r < BF_B; ++r) bins[r] = rbin[(B0 + r) & 15];
                    });
                }
                bool rowv[BF_B];
                float lv[BF_B];
#pragma unroll
                for (int r = 0; r < BF_B; ++r) {
                    const int cl = cl0 + r;
                    rowv[r] = cl >= HALO && cl < HALO + ohe && lp + r < nsteps;          // uniform
                    lv[r] = ST(3, half, r)[writer ? col - HALO : 0];
                    if (!SELF) bins[r] = bin_of(ST(0, half, r)[writer ? col - HALO : 0]);
                }
                bool any = false;
#pragma unroll
                for (int r = 0; r < BF_B; ++r) any = any || rowv[r];
                if (any) stats_rows(rowv, lv, bins);
            }
        }
        // (a2) bins of the mean rows the previous task phase finished (self)
        if (SELF && nb > 0) {
            const int cm0 = l0 - BF_B - R;
            int nbins[BF_B];
            bool bval[BF_B];
#pragma unroll
            for (int r = 0; r < BF_B; ++r) {
                const int cm = cm0 + r;
                bval[r] = cm >= HALO && cm < HALO + ohe && l0 - BF_B + r < nsteps;
                nbins[r] = bin_of(ST(0, half, r)[writer ? col - HALO : 0]);
            }
            uniform_switch16(cm0 & 15, [&](auto bc) {
                constexpr int B0 = decltype(bc)::value;
#pragma unroll
                for (int r = 0; r < BF_B; ++r) rbin[(B0 + r) & 15] = bval[r] ? nbins[r] : rbin[(B0 + r) & 15];
            });
        }
        BF_STAMP(2)
        // (a3) stores of the finished rows, 16 bytes per lane (thread = (map, plane, row, 4 columns))
        {
            // one (map, plane, row) item per wave and trip -- decoded with shifts --, the lanes along the row
            const int per_row = vec_ok ? n4 : ow;
            for (int item = tid >> 6; item < 3 * 2 * BF_B; item += 8) {
                const int m = item / (2 * BF_B), hf = (item / BF_B) & 1, rr = item % BF_B;
                const bool lapmap = m == 2;
                const int lp = l0 - ((SELF && lapmap) ? 2 : 1) * BF_B;
                const int l = lp + rr;
                const int cr = l - R - ((SELF && lapmap) ? R2 : 0);           // finished row
                if (lp < 0 || l >= nsteps || cr < HALO || cr >= HALO + ohe) continue;                       // (uniform)
                float* op = (m == 0 ? o_mean : (m == 1 ? o_var : o_lap)) + ((size_t)(2 * dy + hf) * h + (oy0 + cr - HALO)) * w + ox0;
                const float* sp = ST(lapmap ? 3 : m, hf, rr);
                for (int j = lane; j < per_row; j += 64) {
                    if (vec_ok) *(f32x4*)(op + 4 * j) = *(const f32x4*)(sp + 4 * j);
                    else op[j] = sp[j];
                }
            }
        }
        BF_STAMP(3)
        __syncthreads();
        BF_STAMP(4)
        // ================= task phase =================
        if (tid < ntask_threads) {
            if (tid < tasksA && nb < nbatch) {
                // group A: 29-window sums of stage 1 -> mean, var (collab: + lap)
                const int chunk = tk_chunk, rr = tk_rr, hf = tk_hf;
                const int l = l0 + rr;
                const int cm = l - R;
                if (cm >= HALO && cm < HALO + ohe && l < nsteps) {
                    const int c0 = HALO + chunk * BF_LA, c1 = HALO + ow;
                    float* sm = ST(0, hf, rr) - HALO;
                    float* sv = ST(1, hf, rr) - HALO;
                    float* sl = ST(3, hf, rr) - HALO;
                    if (SELF) {
                        const BfPtrs<2> v = {{V(0, hf, rr), V(1, hf, rr)}};
                        slide(IntC<2>{}, IntC<BF_LA>{}, IntC<K>{}, v, c0, c1, [&](int c, const double (&a)[2]) {
                            const float m = bf_blur_round(a[0], inv_k);
                            const float sd = bf_std_from(m, bf_blur_round(a[1], inv_k));
                            sm[c] = m;
                            sv[c] = __fmul_rn(sd, sd);                                    // var = lr_rggb_k**2 (YOND_SIDD.py:72)
                        });
                    } else {
                        const BfPtrs<4> v = {{V(0, hf, rr), V(1, hf, rr), V(2, hf, rr), V(3, hf, rr)}};
                        slide(IntC<4>{}, IntC<BF_LA>{}, IntC<K>{}, v, c0, c1, [&](int c, const double (&a)[4]) {
                            const float s1 = bf_std_from(bf_blur_round(a[0], inv_k), bf_blur_round(a[1], inv_k));
                            const float mh = bf_blur_round(a[2], inv_k);
                            const float sh = bf_std_from(mh, bf_blur_round(a[3], inv_k));
                            sm[c] = mh;                                                   // mean = blur(hr) (YOND_SIDD.py:97)
                            sv[c] = __fsub_rn(__fmul_rn(s1, s1), __fmul_rn(sh, sh));      // var = lr_k**2 - hr_k**2 (:96)
                            sl[c] = sh;                                                   // img_lap = hr_k (:98)
                        });
                    }
                }
            } else if (SELF && tid >= baseC && tid < baseC + tasksC && nb > 0 && nb <= nbatch) {
                // group C: 29-window sums of stage 2 (b19 rows of the previous batch) -> lap
                const int chunk = tk_chunk, rr = tk_rr, hf = tk_hf;
                const int l = l0 - BF_B + rr;
                const int cl = l - R2 - R;
                if (cl >= HALO && cl < HALO + ohe && l < nsteps) {
                    const int c0 = HALO + chunk * BF_LA, c1 = HALO + ow;
                    float* sl = ST(3, hf, rr) - HALO;
                    const BfPtrs<2> v = {{V(3, hf, rr), V(4, hf, rr)}};
                    slide(IntC<2>{}, IntC<BF_LA>{}, IntC<K>{}, v, c0, c1, [&](int c, const double (&a)[2]) {
                        sl[c] = bf_std_from(bf_blur_round(a[0], inv_k), bf_blur_round(a[1], inv_k));
                    });
                }
            } else if (SELF && tid >= baseB && nb < nbatch) {
                // group B: 19-window sums -> b19 rows (columns HALO - R .. HALO + ow + R)
                const int chunk = tk_chunk, rr = tk_rr, hf = tk_hf;
                const int l = l0 + rr;
                if (l >= 2 * R2 && l < nsteps) {
                    const int c0 = HALO - R + chunk * BF_LB, c1 = HALO + ow + R;
                    float* sb = ST(2, hf, rr) - (HALO - R);
                    const BfPtrs<1> v = {{V(2, hf, rr)}};
                    slide(IntC<1>{}, IntC<BF_LB>{}, IntC<SELF ? K2 : 1>{}, v, c0, c1, [&](int c, const double (&a)[1]) {
                        sb[c] = bf_blur_round(a[0], inv_k2);
                    });
                }
            }
        }
        BF_STAMP(5)
        __syncthreads();
        BF_STAMP(6)
    }
#ifdef BF_STAMPS
    if (blockIdx.x == 1 && blockIdx.y == 1 && blockIdx.z == 0 && (tid & 63) == 0) {
        for (int i = 0; i < 7; ++i) st->dbg[(tid >> 6) * 8 + i] = tacc[i];
    }
#endif
    // frame maximum (the estimator's caller needs lr.max() for the bias LUT grid, YOND_SIDD.py:256/393)
    fmax_ = wave_max(fmax_);
    if ((tid & 63) == 0 && fmax_ > -INFINITY) atomicMax(&st->frame_max_key, f2key(fmax_));
    __syncthreads();
    for (int i = tid; i < BFW_N / 2; i += 512) {
        const unsigned int c = s_h[i];
        if (c & 0xFFFFu) { atomicAdd(&st->hist1[nf_hpos(BFW_LO + i)], c & 0xFFFFu); bmask_ |= 1u << ((BFW_LO + i) >> 12); }
        if (c >> 16) { atomicAdd(&st->hist1[nf_hpos(BFW_LO + BFW_N / 2 + i)], c >> 16); bmask_ |= 1u << ((BFW_LO + BFW_N / 2 + i) >> 12); }
    }
    if (tid == 0 && s_h[BFW_N / 2]) { atomicAdd(&st->hist1[nf_hpos(NF_WIN_LO)], s_h[BFW_N / 2]); bmask_ |= 1u << (NF_WIN_LO >> 12); }
    nf_mark_blocks(st, bmask_);
    for (int i = tid; i < NF_BINS; i += 512) {
        const unsigned int v = s_mi[i];
        if (v) atomicMax(&st->maxinv[i], v);
    }
    const unsigned int nblocks = gridDim.x * gridDim.y * gridDim.z;
    if (nf_arrive_last(&st->ticket[0], nblocks)) nf_resolve1(st, args, (unsigned int*)s_raw);
}

template <int MODE, int K, int K2>
static int launch_fused(const float* fa, const float* fb, int H, int W, int k, int k2, int tile_w, float* mean, float* var,
                        float* lap, const double* q_host, int nq, void* ws, hipStream_t st) {
    const int h = H / 2, w = W / 2;
    if (h < 1 || w < 1 || k < 1 || !(k & 1) || k2 < 1 || !(k2 & 1) || tile_w < 0) return YOND_EINVAL;
    if (k != K || k2 != K2) return YOND_EUNSUPPORTED;                    // the kernel is built for the estimator's windows (29, 19)
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
    if (!(bw & 3) && ((g.ow_nom + 3) & ~3) <= maxow) g.ow_nom = (g.ow_nom + 3) & ~3;     // 16-byte store rows
    // row segments: one workgroup per CU (132 KB of LDS) in one round; longer segments re-read fewer halo rows
    const long target = yond_exp_long("YOND_BOX_WGS", 240);
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
    const size_t lds = (size_t)NQ * 2 * BF_B * BF_RS * 8 + (size_t)4 * 2 * BF_B * BF_T * 4 + (BFW_N / 2 + 4 + NF_BINS) * 4;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)box_fused_kernel<MODE, K, K2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    hipError_t e = hipMemsetAsync(ws, 0, nf_state_bytes(), st);
    if (e != hipSuccess) return (int)e;
    dim3 grid((unsigned)(nblk * g.nstrip), (unsigned)nsy, 2);
    hipLaunchKernelGGL((box_fused_kernel<MODE, K, K2>), grid, dim3(512), lds, st, fa, fb, g, mean, var, lap, (NleState*)ws, a);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

extern "C" int yond_box_stats_self_fused_f32(const float* bayer, int H, int W, int k, int k2, int tile_w, float* mean,
                                             float* var, float* lap, const double* q_host, int nq, void* ws, void* stream) {
    if (!bayer || !mean || !var || !lap || (H & 1) || (W & 1)) return YOND_EINVAL;
    return launch_fused<0, 29, 19>(bayer, nullptr, H, W, k, k2, tile_w, mean, var, lap, q_host, nq, ws, (hipStream_t)stream);
}

extern "C" int yond_box_stats_collab_fused_f32(const float* bayer_lr, const float* bayer_hr, int H, int W, int k, int tile_w,
                                               float* mean, float* var, float* lap, const double* q_host, int nq, void* ws,
                                               void* stream) {
    if (!bayer_lr || !bayer_hr || !mean || !var || !lap || (H & 1) || (W & 1)) return YOND_EINVAL;
    return launch_fused<2, 29, 29>(bayer_lr, bayer_hr, H, W, k, k, tile_w, mean, var, lap, q_host, nq, ws, (hipStream_t)stream);
}

#endif  // YOND_EXPERIMENTS
