// N4: weight gradient of a 3x3 stride-1 convolution on the fp16 matrix cores with fp32-accurate SPLIT operands.
//
//   dW[tap][co][ci] = sum over pixels p of dY[p][co] * X[p + tap][ci]            (trainer_AWGN.py:116 loss.backward(): the
//                                                                                 weight gradient of every nn.Conv2d(c, c, 3, 1, 1))
// The fp32-input MFMA (train.hip, 157 TF/s peak) was 29 % of a training step.  Here the K axis of the MFMA is the PIXEL axis:
// v_mfma_f32_32x32x16_f16 with A = 32 output channels of dY x 16 pixels, B = the same 16 pixels (shifted by the tap) x 32 input
// channels of X.  The tensors are [N][H][W][C] float32, so an operand row (one channel, 16 consecutive pixels) is a strided
// gather: both tensors are staged through LDS as [pixel][32 channels] fp16 tiles and read back TRANSPOSED by
// ds_read_b64_tr_b16 (lane l receives channel l % 32 of pixels 8 (l / 32) + 0..3; two reads per operand half).
// Split operands as in conv_split_kernel.h: a = h + l 2^-11 with h = fp16(a), l = fp16((a - h) 2^11), product = h_a h_b +
// 2^-11 (h_a l_b + l_a h_b).  Unlike there, ALL THREE products go into ONE accumulator: dY is staged in three parts
// (P = fp16(2^11 a), h, l) and P h_x + h L_x + l h_x carries the common factor 2^11, removed when the tile is stored.  That
// halves the accumulator registers -- a wave holds all nine taps of a 32 x 32 tile (144 registers) -- at the price of
// |dY| < 32 (P must stay finite; TrainStep's loss scale keeps the back-propagated values O(1); a violation sets bit 0 of
// *status).  fp16 x fp16 products are exact in the fp32 accumulator; the dropped l_a l_b term is 2^-22 relative.
//
// Pixel stream: every image is padded to (H + 2) x (W + 2) (zero halo) and the images are laid end to end, the SAME
// coordinates for both tensors: stream index g = n Ls + (y + 1)(W + 2) + (x + 1).  Then the tap (ky, kx) of output pixel g
// reads X at g + (ky - 1)(W + 2) + (kx - 1) whatever the row, a zero of dY's halo kills every product that would cross an
// image edge, and a 16-pixel K step is 16 consecutive stream positions -- rows of any width fill the K steps (8 x 8 patches
// of the 512-channel level included; cost: (H + 2)(W + 2) / (H W) more K steps).
// Workgroup (512 threads, 8 waves) = PA x PB tiles of 32 x 32 (co, ci) x KS-way split of the step's K range; it walks a slice
// of the stream in steps of SK = 16 KS KPW pixels with ONE barrier per step: X lives in an LDS ring that is always `lead` =
// W + 3 pixels ahead of and behind the step's pixels, dY in two buffers; the global loads of step s + 1 are issued before the
// MFMAs of step s and split / written to LDS behind them.  The K axis (N Ls pixels) is cut into `nslices` slices; every
// workgroup stores its partial tiles to ws[slice][tap][Co][Ci], train.hip's wgrad_reduce_kernel adds the slices.
#include "common.h"

typedef short v4s __attribute__((__vector_size__(4 * sizeof(short))));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

struct WgsGeom {
    const float* x;            // [N][H][W][Ci]
    const float* dy;           // [N][H][W][Co]
    float* ws;                 // [nslices][9 Co Ci (+ Co: the slice's column sums of dY, with_bias)]
    int with_bias;
    int* status;
    int N, H, W, Ci, Co;
    int W2, H2, Ls;            // W + 2, H + 2, H2 * W2
    int total;                 // N * Ls
    int steps_per_slice, nslices;
    int ring;                  // pixels of the X ring (multiple of 4)
    int lead;                  // W + 3
    int abl;                   // experiments builds: timing-only ablations (1 no global loads, 2 no MFMAs, 4 no staging writes)
};

template <int PA, int PB, int KS, int KPW>
struct WgsCfg {
    static constexpr int NT = 512;
    static constexpr int SK = 16 * KS * KPW;                   // pixels per step
    static constexpr int NX = SK * PB * 8 / NT;                // 16-byte items of X a thread stages per step
    static constexpr int ND = SK * PA * 8 / NT;                // ... of dY
    static_assert(PA * PB * KS == 8, "eight waves");
    static_assert(SK * PB * 8 % NT == 0 && SK * PA * 8 % NT == 0, "whole items per thread");
    static constexpr size_t dy_bytes() { return (size_t)2 * PA * 3 * SK * 64; }
    static constexpr size_t x_bytes(int ring) { return (size_t)PB * 2 * ring * 64; }
};

// position of stream index g: image n, padded row r (0 .. H + 1), padded column c (0 .. W + 1)
struct StreamPos { int n, r, c; };
__device__ __forceinline__ StreamPos stream_pos(int g, int Ls, int W2) {
    StreamPos p;
    if (g < 0) { p.n = -1; p.r = 0; p.c = 0; return p; }
    p.n = g / Ls;
    const int i = g - p.n * Ls;
    p.r = i / W2;
    p.c = i - p.r * W2;
    return p;
}
__device__ __forceinline__ void stream_advance(StreamPos& p, int by, int W2, int H2) {
    if (p.n < 0) return;                                       // (only the first slice starts before the stream; handled by its caller)
    p.c += by;
    while (p.c >= W2) { p.c -= W2; ++p.r; }
    while (p.r >= H2) { p.r -= H2; ++p.n; }
}

template <int PA, int PB, int KS, int KPW>
__global__ __launch_bounds__(512) void wgrad_split_kernel(const WgsGeom g) {
    using C = WgsCfg<PA, PB, KS, KPW>;
    constexpr int SK = C::SK, NX = C::NX, ND = C::ND;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const xr = smem;                                      // [PB][2 parts][ring][32] halves
    char* const dyb = smem + C::x_bytes(g.ring);                // [2][PA][3 parts][SK][32] halves
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gb_n = (g.Ci / 32) / PB;
    const int group = blockIdx.x / g.nslices, slice = blockIdx.x - group * g.nslices;
    const int a0 = (group / gb_n) * PA, b0 = (group % gb_n) * PB;      // first co / ci tile of the workgroup
    const int pw = wave / KS, ks = wave - pw * KS;
    const int wa = pw / PB, wb = pw - wa * PB;                  // the wave's tile inside the workgroup's PA x PB
    const int G0 = slice * g.steps_per_slice * SK;              // first stream index of the slice
    int nsteps = g.steps_per_slice;
    {
        const int left = (g.total - G0 + SK - 1) / SK;
        if (nsteps > left) nsteps = left;
    }
    float amax = 0.0f;
    // with_bias: the bias gradient db[co] = sum over pixels of dY rides along -- every dY value passes through stage_dy exactly
    // once per output-channel group, so the workgroups of input-channel group 0 add up what they stage (a pass of its own over
    // dY cost 27 us per layer)
    const bool sum_dy = g.with_bias && (group % gb_n) == 0;
    f32x4 dsum[ND];
#pragma unroll
    for (int k = 0; k < ND; ++k) dsum[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    // ---- staging: one item = 4 consecutive channels of one pixel (16 bytes of float32 -> 8 bytes per fp16 part) ----
    // stream index of an item is (batch start) + po, po = it / (8 tiles); the item's position (n, r, c) advances by SK per step
    auto x_item = [&](int it, int& po, int& tile, int& c4) { c4 = it & 7; tile = (it >> 3) % PB; po = (it >> 3) / PB; };
    auto d_item = [&](int it, int& po, int& tile, int& c4) { c4 = it & 7; tile = (it >> 3) % PA; po = (it >> 3) / PA; };
    // (the load is unconditional -- a masked item reads element 0..3 -- and its value is NOT touched here: the zeroing of a
    // masked item happens where the value is consumed, behind the MFMAs, or the wave would wait for HBM in front of them)
    // (branch-free: a branch between two loads makes the compiler drain the first before it issues the second)
    auto px_ok = [&](const StreamPos& p) -> bool {
        return ((unsigned)p.n < (unsigned)g.N) & ((unsigned)(p.r - 1) < (unsigned)g.H) & ((unsigned)(p.c - 1) < (unsigned)g.W);
    };
    auto load_raw = [&](const float* base, int Cc, const StreamPos& p, int ch, bool ok) -> f32x4 {
        const size_t off = ok ? (((size_t)p.n * g.H + (p.r - 1)) * g.W + (p.c - 1)) * Cc + ch : 0;
        return *(const f32x4*)(base + off);
    };
    auto masked = [&](const f32x4& v, bool ok) -> f32x4 {
        const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
        return ok ? v : z;
    };
    auto load_px = [&](const float* base, int Cc, const StreamPos& p, int ch) -> f32x4 {
        const bool ok = px_ok(p);
        return masked(load_raw(base, Cc, p, ch, ok), ok);
    };
    auto split_x = [&](const f32x4& v, char* dst_h, char* dst_l) {
        const f16x4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        const f16x4 l = {(_Float16)((v[0] - (float)h[0]) * 2048.0f), (_Float16)((v[1] - (float)h[1]) * 2048.0f),
                         (_Float16)((v[2] - (float)h[2]) * 2048.0f), (_Float16)((v[3] - (float)h[3]) * 2048.0f)};
        *(f16x4*)dst_h = h;
        *(f16x4*)dst_l = l;
    };
    auto x_dst = [&](int tile, int part, int pos, int c4) -> char* { return xr + (unsigned)(((tile * 2 + part) * g.ring + pos) * 64 + c4 * 8); };
    auto d_dst = [&](int buf, int tile, int part, int o, int c4) -> char* {
        return dyb + (unsigned)((((buf * PA + tile) * 3 + part) * SK + o) * 64 + c4 * 8);
    };
    auto stage_dy = [&](const f32x4& v, int buf, int tile, int o, int c4, f32x4& sum, bool count) {
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
        if (count) { sum[0] += v[0]; sum[1] += v[1]; sum[2] += v[2]; sum[3] += v[3]; }
        const f16x4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        const f16x4 P = {(_Float16)(v[0] * 2048.0f), (_Float16)(v[1] * 2048.0f), (_Float16)(v[2] * 2048.0f), (_Float16)(v[3] * 2048.0f)};
        const f16x4 l = {(_Float16)((v[0] - (float)h[0]) * 2048.0f), (_Float16)((v[1] - (float)h[1]) * 2048.0f),
                         (_Float16)((v[2] - (float)h[2]) * 2048.0f), (_Float16)((v[3] - (float)h[3]) * 2048.0f)};
        *(f16x4*)d_dst(buf, tile, 0, o, c4) = P;
        *(f16x4*)d_dst(buf, tile, 1, o, c4) = h;
        *(f16x4*)d_dst(buf, tile, 2, o, c4) = l;
    };

    // ---- prologue: X [G0 - lead, G0 + SK + lead) into ring positions 0 ..; dY [G0, G0 + SK) into buffer 0 ----
    const int xs0 = G0 - g.lead;                                 // stream index at ring position 0
    for (int base = 0; base < SK + 2 * g.lead; base += SK) {
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            int po, tile, c4;
            x_item(tid + k * C::NT, po, tile, c4);
            const int rel = base + po;
            if (rel < SK + 2 * g.lead) {
                const StreamPos p = stream_pos(xs0 + rel, g.Ls, g.W2);
                const f32x4 v = load_px(g.x, g.Ci, p, (b0 + tile) * 32 + c4 * 4);
                split_x(v, x_dst(tile, 0, rel, c4), x_dst(tile, 1, rel, c4));
            }
        }
    }
    StreamPos xp[NX], dp[ND];
    int xpos[NX];                                                // ring position the item's next pixel goes to
#pragma unroll
    for (int k = 0; k < NX; ++k) {
        int po, tile, c4;
        x_item(tid + k * C::NT, po, tile, c4);
        xp[k] = stream_pos(G0 + SK + g.lead + po, g.Ls, g.W2);
        int ps = SK + 2 * g.lead + po;
        if (ps >= g.ring) ps -= g.ring;
        xpos[k] = ps;
    }
#pragma unroll
    for (int k = 0; k < ND; ++k) {
        int po, tile, c4;
        d_item(tid + k * C::NT, po, tile, c4);
        dp[k] = stream_pos(G0 + po, g.Ls, g.W2);
        const f32x4 v = load_px(g.dy, g.Co, dp[k], (a0 + tile) * 32 + c4 * 4);
        stage_dy(v, 0, tile, po, c4, dsum[k], true);
        stream_advance(dp[k], SK, g.W2, g.H2);
    }
    __syncthreads();

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    // fragment addressing (ds_read_b64_tr_b16): group gq = lane / 16 reads channels 16 (gq & 1) .. of pixels 8 (gq >> 1) + q,
    // lane 4 q + p of the group supplies the address of row (pixel) q, channels 4 p .. 4 p + 3
    const int gq = lane >> 4, fq = (lane >> 2) & 3, fp = lane & 3;
    const int frag_px = 8 * (gq >> 1) + fq;                      // + 4 for the second read
    const int frag_ch = (16 * (gq & 1) + 4 * fp) * 2;            // bytes
    typedef __attribute__((address_space(3))) v4s* lds_v4s;
    auto rd = [&](const char* p) -> v4s { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)p); };
    union Frag { v4s s[2]; f16x8 h; };

    // Step s (one barrier): the MFMAs of batch s; the staging of batch s + 1 (X: the SK pixels in front of the ring; dY: the other
    // buffer) from registers that were loaded during step s - 1; the loads of batch s + 2 into those registers.  The two waves
    // of a SIMD take these in OPPOSITE order -- waves 4-7 stage and load first and multiply last, waves 0-3 multiply first --
    // so that one wave's vector / memory work runs under its partner's MFMAs instead of both waves' at the same time.
    const bool wave_hi = wave >= 4;
    f32x4 vx[NX], vd[ND];
    bool okx[NX], okd[ND];
    auto issue_loads = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            int po, tile, c4;
            x_item(tid + k * C::NT, po, tile, c4);
            okx[k] = px_ok(xp[k]);
            if (!(g.abl & 1)) vx[k] = load_raw(g.x, g.Ci, xp[k], (b0 + tile) * 32 + c4 * 4, okx[k]);
            else vx[k] = f32x4{0.5f, 0.25f, -0.5f, 1.0f};
        }
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            int po, tile, c4;
            d_item(tid + k * C::NT, po, tile, c4);
            okd[k] = px_ok(dp[k]);
            if (!(g.abl & 1)) vd[k] = load_raw(g.dy, g.Co, dp[k], (a0 + tile) * 32 + c4 * 4, okd[k]);
            else vd[k] = f32x4{0.5f, 0.25f, -0.5f, 1.0f};
        }
    };
    issue_loads();                                               // batch 1 (staged during step 0)
    int wbase = 0;                                               // ring position of stream index g_s - lead
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        auto stage_and_load = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < NX; ++k) {
                int po, tile, c4;
                x_item(tid + k * C::NT, po, tile, c4);
                if (!(g.abl & 4)) split_x(masked(vx[k], okx[k]), x_dst(tile, 0, xpos[k], c4), x_dst(tile, 1, xpos[k], c4));
                xpos[k] += SK;
                xpos[k] -= xpos[k] >= g.ring ? g.ring : 0;
            }
#pragma unroll
            for (int k = 0; k < ND; ++k) {
                int po, tile, c4;
                d_item(tid + k * C::NT, po, tile, c4);
                // (the batch staged by the slice's last step belongs to the next slice: not this one's to count)
                if (!(g.abl & 4)) stage_dy(masked(vd[k], okd[k]), buf ^ 1, tile, po, c4, dsum[k], s + 1 < nsteps);
            }
            // (the cursors behind the consumers: their carry loops are branches)
#pragma unroll
            for (int k = 0; k < NX; ++k) stream_advance(xp[k], SK, g.W2, g.H2);
#pragma unroll
            for (int k = 0; k < ND; ++k) stream_advance(dp[k], SK, g.W2, g.H2);
            issue_loads();
        };
        if (wave_hi) stage_and_load();
        // ---- the step's products ----
        // Software pipeline over (K step, kernel row): the fragments of the NEXT kernel row are requested before the nine MFMAs
        // of the current row are issued, and those nine go tap-interleaved (three accumulators in turn), so neither an LDS
        // latency nor a dependent accumulator sits in front of an MFMA.  (Scheduling fences keep the reads where they are.)
        // (register budget: one set of dY fragments, two of the h halves of X -- the next kernel row's are requested a row ahead --
        // and ONE of the l halves, requested at the top of their own row: the three h_a h_x MFMAs in front of their first use
        // cover the LDS latency)
        Frag A[1][3], Bh[2][3], Bl[1][3];
        const unsigned abase = (unsigned)((((buf * PA + wa) * 3) * SK + frag_px) * 64) + frag_ch;
        auto load_A = [&](int kp, int set) {
            const int o = 16 * (ks * KPW + kp);                  // first pixel of the wave's K step inside the batch
#pragma unroll
            for (int part = 0; part < 3; ++part) {
                const char* p = dyb + abase + (unsigned)((part * SK + o) * 64);
                A[set][part].s[0] = rd(p);
                A[set][part].s[1] = rd(p + 4 * 64);
            }
        };
        const char* const ph = xr + (unsigned)((wb * 2 + 0) * g.ring * 64) + frag_ch;
        const char* const pl = xr + (unsigned)((wb * 2 + 1) * g.ring * 64) + frag_ch;
        auto load_B = [&](int kp, int ky, int set, bool low) {
            const int o = 16 * (ks * KPW + kp);
            // ring position of the K step's first pixel for tap (ky, 0): stream index g_s + o + (ky - 1) W2 - 1
            const int rel0 = wbase + g.lead + o + (ky - 1) * g.W2 - 1 + frag_px;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int rel = rel0 + kx, rel1 = rel + 4;
                const unsigned r0 = (unsigned)(rel >= g.ring ? rel - g.ring : rel) * 64u;
                const unsigned r1 = (unsigned)(rel1 >= g.ring ? rel1 - g.ring : rel1) * 64u;
                if (low) {
                    Bl[0][kx].s[0] = rd(pl + r0);
                    Bl[0][kx].s[1] = rd(pl + r1);
                } else {
                    Bh[set][kx].s[0] = rd(ph + r0);
                    Bh[set][kx].s[1] = rd(ph + r1);
                }
            }
        };
        if (!(g.abl & 2)) {
        load_A(0, 0);
        load_B(0, 0, 0, false);
        static_for<0, KPW * 3>([&](auto ic) {
            constexpr int i = decltype(ic)::value, kp = i / 3, ky = i % 3;
            load_B(kp, ky, 0, true);
            if constexpr (i + 1 < KPW * 3) load_B((i + 1) / 3, (i + 1) % 3, (i + 1) & 1, false);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int part = 0; part < 3; ++part)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    // part 0: 2^11 h_a h_x ; 1: h_a (2^11 l_x) ; 2: (2^11 l_a) h_x
                    acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0][part].h, part == 1 ? Bl[0][kx].h : Bh[i & 1][kx].h,
                                                                             acc[ky * 3 + kx], 0, 0, 0);
            if constexpr (ky == 2 && i + 1 < KPW * 3) load_A(kp + 1, 0);      // (behind the K step's last MFMAs: its readers)
            __builtin_amdgcn_sched_barrier(0);
        });
        }
        if (!wave_hi) stage_and_load();
        wbase += SK;
        if (wbase >= g.ring) wbase -= g.ring;
        __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // (the loads of the batches past the end)
    if (g.status && !(amax < 31.9f)) atomicOr(g.status, 1);      // a P part (2^11 dy) left fp16's range

    // ---- the workgroup's partial tiles -> ws[slice][tap][Co][Ci] (x 2^-11: the products' common factor) ----
    // Accumulator layout: lane l holds column (ci) l % 32, rows (co) (r & 3) + 8 (r >> 2) + 4 (l / 32).  Stored from there a
    // wave-instruction would move 2 x 128 bytes, and 144 of them per wave made the write-out longer than the main loop (store
    // issue, not bandwidth).  So, tap by tap, every wave puts its tile into LDS (the rings are dead now), the KS waves of a
    // tile are added up there, and each thread stores 16 bytes: a wave-instruction covers eight whole 128-byte rows.
    const size_t slice_floats = (size_t)9 * g.Co * g.Ci + (g.with_bias ? g.Co : 0);
    float* const wsl = g.ws + (size_t)slice * slice_floats;
    const int li = lane & 31, lk = lane >> 5;
    float* const red = (float*)smem;                             // [2 buffers][8 waves][32 co][32 ci]
    constexpr int NP = PA * PB;
    __syncthreads();
    if (g.with_bias) {                                           // (uniform over the workgroup: the barriers below are safe)
        // the threads' column sums -> LDS, item-major (item it holds channels (tile, c4) of pixel it / (8 PA)) -> one thread per
        // channel adds its column -> the slice's db block.  (LDS float atomics here cost 50 us per launch.)
        float* cs = red;                                         // [ND * NT items][4]; the tile buffers are not in use yet
#pragma unroll
        for (int k = 0; k < ND; ++k) *(f32x4*)(cs + (size_t)(tid + k * C::NT) * 4) = dsum[k];
        __syncthreads();
        if (sum_dy && tid < PA * 32) {
            const int tile = tid >> 5, c4 = (tid >> 2) & 7, e = tid & 3;
            float a = 0.0f;
            for (int po = 0; po < SK; ++po) a += cs[(size_t)(((po * PA + tile) << 3) + c4) * 4 + e];
            wsl[(size_t)9 * g.Co * g.Ci + a0 * 32 + tid] = a;
        }
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float* rb = red + (t & 1) * 8 * 1024;
#pragma unroll
        for (int r = 0; r < 16; ++r) rb[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * 32 + li] = acc[t][r];
        __syncthreads();                                         // (two buffers: the next tap's writes need no second barrier)
        for (int i = tid; i < NP * 256; i += C::NT) {
            const int c4 = i & 7, row = (i >> 3) & 31, p = i >> 8;
            f32x4 v = *(const f32x4*)(rb + ((p * KS) * 32 + row) * 32 + c4 * 4);
#pragma unroll
            for (int k2 = 1; k2 < KS; ++k2) {
                const f32x4 u = *(const f32x4*)(rb + ((p * KS + k2) * 32 + row) * 32 + c4 * 4);
                v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3];
            }
            v[0] *= 1.0f / 2048.0f; v[1] *= 1.0f / 2048.0f; v[2] *= 1.0f / 2048.0f; v[3] *= 1.0f / 2048.0f;
            const int pa = p / PB, pb = p - pa * PB;
            *(f32x4*)(wsl + ((size_t)t * g.Co + (a0 + pa) * 32 + row) * g.Ci + (b0 + pb) * 32 + c4 * 4) = v;
        }
    }
}

// ---- host side ----
struct WgsPlan { int cfg; int nslices, steps_per_slice, ring; size_t lds; };   // cfg 0: not supported

template <int PA, int PB, int KS, int KPW>
static bool wgs_try(int N, int H, int W, int Ci, int Co, int cfg, WgsPlan& p) {
    using C = WgsCfg<PA, PB, KS, KPW>;
    const int Ta = Co / 32, Tb = Ci / 32;
    if (Ta % PA || Tb % PB) return false;
    const int lead = W + 3;
    const int ring = (2 * C::SK + 2 * lead + 3) / 4 * 4;
    const size_t lds = C::x_bytes(ring) + C::dy_bytes();
    if (lds > 160 * 1024) return false;
    const long long total = (long long)N * (H + 2) * (W + 2);
    if (total >= 0x7fffffffLL / 2) return false;
    const long long steps = (total + C::SK - 1) / C::SK;
    const long long groups = (long long)(Ta / PA) * (Tb / PB);
    // about one workgroup per CU and round (256 CUs), at least 4 steps per slice (ring warm-up: 2 lead + SK pixels per slice)
    long long ns = 256 / groups;
    if (ns < 1) ns = 1;
    if (ns > steps / 4) ns = steps / 4 > 0 ? steps / 4 : 1;
    const long long sps = (steps + ns - 1) / ns;
    ns = (steps + sps - 1) / sps;
    p.cfg = cfg;
    p.nslices = (int)ns;
    p.steps_per_slice = (int)sps;
    p.ring = ring;
    p.lds = lds < 64 * 1024 ? 64 * 1024 : lds;                   // (the write-out stages the tiles in 64 KB of it)
    return true;
}

static WgsPlan wgs_plan(int N, int H, int W, int Ci, int Co) {
    WgsPlan p{0, 0, 0, 0, 0};
    if (N <= 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0 || Ci % 32 || Co % 32) return p;
    if ((long long)N * H * W * (Ci > Co ? Ci : Co) >= 0x7fffffffLL) return p;
    const int pairs = (Ci / 32) * (Co / 32);
    const long force = yond_exp_long("YOND_WGS_CFG", 0);      // (experiment builds: try one configuration first)
    if (force == 2 && wgs_try<2, 2, 2, 1>(N, H, W, Ci, Co, 2, p)) return p;
    if (force == 4 && wgs_try<2, 1, 4, 1>(N, H, W, Ci, Co, 4, p)) return p;
    if (force == 5 && wgs_try<1, 2, 4, 1>(N, H, W, Ci, Co, 5, p)) return p;
    if (force == 1 && wgs_try<1, 1, 8, 1>(N, H, W, Ci, Co, 1, p)) return p;
    if (pairs == 1 && wgs_try<1, 1, 8, 1>(N, H, W, Ci, Co, 1, p)) return p;
    // 2 x 2 tile pairs per workgroup up to 256 x 256 channels: half the slice workspace of 2 x 4 (36 instead of 72 MB written and re-read by the
    // reduction) outweighs its extra staging per MFMA -- 128 ch 194 vs 178 TF/s, 256 ch 187 vs 176; at 512 ch 2 x 4 wins (153 vs 149)
    if (pairs <= 64 && wgs_try<2, 2, 2, 1>(N, H, W, Ci, Co, 2, p)) return p;
    if (wgs_try<2, 4, 1, 2>(N, H, W, Ci, Co, 3, p)) return p;
    if (wgs_try<2, 2, 2, 1>(N, H, W, Ci, Co, 2, p)) return p;
    if (wgs_try<2, 1, 4, 1>(N, H, W, Ci, Co, 4, p)) return p;
    if (wgs_try<1, 2, 4, 1>(N, H, W, Ci, Co, 5, p)) return p;
    if (wgs_try<1, 1, 8, 1>(N, H, W, Ci, Co, 1, p)) return p;
    p.cfg = 0;
    return p;
}

// bytes of workspace yond_conv_wgrad_split_f32 needs (0: the layer does not fit this kernel -- the caller keeps yond_conv_wgrad_ws_f32)
extern "C" size_t yond_conv_wgrad_split_ws_bytes(int N, int H, int W, int Cin, int Cout) {
    const WgsPlan p = wgs_plan(N, H, W, Cin, Cout);
    return p.cfg ? (size_t)p.nslices * ((size_t)9 * Cout * Cin + Cout) * sizeof(float) : 0;
}

template <int PA, int PB, int KS, int KPW>
static int wgs_launch(const WgsGeom& g, const WgsPlan& p, hipStream_t st) {
    auto kern = wgrad_split_kernel<PA, PB, KS, KPW>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int groups = (g.Co / 32 / PA) * (g.Ci / 32 / PB);
    hipLaunchKernelGGL(kern, dim3((unsigned)(groups * p.nslices)), dim3(512), p.lds, st, g);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

int yond_wgrad_reduce_launch(const float* ws, int nchunk, size_t n, float* dw, hipStream_t st, int oihw_taps, size_t cc);     // train.hip

// dw[9][Cout][Cin] of a 3x3 stride-1 pad-1 convolution from x [N][H][W][Cin], dy [N][H][W][Cout] (float32, channels multiples of 32)
// with_bias != 0: dw has 9 Cout Cin + Cout floats, the last Cout = db[co] = the sum over all pixels of dy (the bias gradient);
// with_bias & 2: the weight gradient is written in OIHW order, dw[co][ci][tap] (an nn.Conv2d weight's), instead of [tap][co][ci]
extern "C" int yond_conv_wgrad_split_f32(const float* x, const float* dy, int N, int H, int W, int Cin, int Cout, float* dw, int with_bias,
                                         float* ws, size_t ws_bytes, int* status, void* stream) {
    const int oihw = (with_bias & 2) ? 9 : 0;
    with_bias &= 1;
    if (!x || !dy || !dw || !ws) return YOND_EINVAL;
    const WgsPlan p = wgs_plan(N, H, W, Cin, Cout);
    if (!p.cfg) return YOND_EUNSUPPORTED;
    const size_t slice_floats = (size_t)9 * Cout * Cin + (with_bias ? Cout : 0);
    if (ws_bytes < (size_t)p.nslices * slice_floats * sizeof(float)) return YOND_EINVAL;
    WgsGeom g;
    g.x = x; g.dy = dy; g.ws = ws; g.status = status; g.with_bias = with_bias ? 1 : 0;
    g.N = N; g.H = H; g.W = W; g.Ci = Cin; g.Co = Cout;
    g.W2 = W + 2; g.H2 = H + 2; g.Ls = g.W2 * g.H2;
    g.total = N * g.Ls;
    g.steps_per_slice = p.steps_per_slice; g.nslices = p.nslices;
    g.ring = p.ring; g.lead = W + 3;
    g.abl = (int)yond_exp_long("YOND_WGS_ABL", 0);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    switch (p.cfg) {
        case 1: rc = wgs_launch<1, 1, 8, 1>(g, p, st); break;
        case 2: rc = wgs_launch<2, 2, 2, 1>(g, p, st); break;
        case 3: rc = wgs_launch<2, 4, 1, 2>(g, p, st); break;
        case 4: rc = wgs_launch<2, 1, 4, 1>(g, p, st); break;
        default: rc = wgs_launch<1, 2, 4, 1>(g, p, st); break;
    }
    if (rc != YOND_OK) return rc;
    return yond_wgrad_reduce_launch(ws, p.nslices, slice_floats, dw, st, oihw, (size_t)Cout * Cin);
}
