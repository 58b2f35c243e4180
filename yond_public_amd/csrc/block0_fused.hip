// K2f: one whole LEVEL-0 residual block of the guided U-Net in ONE launch (archs/modules.py:186-196 at nf = 32):
//
//        out = conv2( SiLU( FiLM1( conv1( SiLU(x) ) ) ) ) * s2 + t2 + x            32 -> 32 -> 32 channels, 3x3, stride 1
//
// Why ("Level 0: the bytes"): at level 0 the two convolutions of a block are separate launches that move
// x (388 MB at 1504 x 2016) -> tmp (388 MB written, 388 MB read back) -> out (388 MB) + the residual x again (388 MB): 1.94 GB for
// 112 GFLOP, and the conv2 launch already runs at the HBM ceiling.  Here tmp never leaves the chip: a workgroup computes conv1 on
// a (TH + 2) x 34 region around its TH x 32 output tile, writes SiLU(FiLM(.)) in split halves into LDS and runs conv2 from there.
// HBM traffic: x once (+ its halo, + the residual read, both mostly L2 hits) and out once.
//
// Arithmetic: the split-operand products of conv_split_kernel.h -- a = h + l 2^-11 with h = fp16(a), l = fp16((a - h) 2^11); a
// product block = three fp16 MFMAs, h_w h_x into one accumulator, h_w l_x + l_w h_x into a second one folded in as acc + acc2 2^-11.
// The MFMA here is v_mfma_f32_16x16x32_f16: K = 32 is ALL input channels of a level-0 layer, so a tap is one K step and there is no
// channel-chunk loop, no weight streaming and no per-step barrier: both layers' weights (2 x 36 KB) stay in LDS for the whole
// kernel.  Operand map (cdna_hip_programming.md section 3): lane l supplies A[row l & 15][k = 8 (l >> 4) + j] and
// B[k = 8 (l >> 4) + j][col l & 15], and owns D[row 4 (l >> 4) + i][col l & 15].  A = weights (row = output channel), B = pixels.
//
// Tile: TH = 12 rows x 32 pixels.  LDS (148 KB):
//   image   8 planes [channel group kg 0..3][part h, l][16 x 36 pixels] of 16-byte units (8 channels): first SiLU(x) of the
//           (TH + 4) x 36 input region, then -- the SAME bytes, after a barrier -- the (TH + 2) x 36 region of conv1's output
//           ("mid"; both regions use the pitch 36 so that conv1 walks mid as a FLAT pixel index: 32 work items of 16
//           consecutive pixels, 4 per wave, no ragged row ends);
//   weights [layer][tap][part][kg][32 output channels] x 16 bytes.
// Per tile: conv1 (32 items x 9 taps x 6 MFMAs) | barrier | epilogue 1 -> mid | barrier | conv2 (24 items: 12 rows x 2 halves, 3 per
// wave) with the NEXT tile's input loads and this tile's residual loads in flight | epilogue 2 -> HBM | barrier | stage the next input.
#ifdef YOND_EXPERIMENTS     // (a measured no-go, built into experiment libraries only: include/yond_hip_experiments.h)
#include "common.h"
#include "../../include/yond_hip_experiments.h"

typedef _Float16 b0_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 b0_f16x4 __attribute__((ext_vector_type(4)));

#ifndef B0_DBG
#define B0_DBG 0                   // 1 (tools/b0_dbg.py): cycle stamps of workgroup 0, waves 0 and 4, at the phase boundaries of its first tiles
#endif
#if B0_DBG
static __device__ unsigned long long g_b0_dbg[2][32][12];
extern "C" int yond_block0_debug_read(unsigned long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_b0_dbg), sizeof(g_b0_dbg)); }
#define BDBG(slot)                                                                                    \
    do {                                                                                              \
        if (blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0 && dbg_tile < 32)                \
            g_b0_dbg[wave >> 2][dbg_tile][slot] = __builtin_readcyclecounter();                       \
    } while (0)
#else
#define BDBG(slot) do {} while (0)
#endif

namespace {

constexpr int B0_TH = 12, B0_TW = 32, B0_P = 36;                 // output tile; pitch of both LDS regions
constexpr int B0_IH = B0_TH + 4, B0_MH = B0_TH + 2;              // rows of the input region / of mid
constexpr int B0_PU = B0_IH * B0_P;                              // 576 units per plane (a multiple of 16: plane strides of 256 B keep fragment reads conflict-free)
constexpr int B0_IMG_BYTES = 8 * B0_PU * 16 + 256;               // + the over-read of conv1's last (masked) lanes
constexpr int B0_W_BYTES = 9 * 2 * 4 * 32 * 16;                  // 36,864 per layer
constexpr int B0_FILM_OFF = B0_IMG_BYTES + 2 * B0_W_BYTES;        // 4 x 32 floats: s1, t1, s2, t2 of the tile's image
constexpr int B0_SMEM = B0_FILM_OFF + 512;
constexpr int B0_NIN = (B0_PU * 8) / 512;                        // 9 staged 4-channel items per thread
constexpr int B0_MID_PIX = B0_MH * B0_P;                         // 504 flat mid pixels (32 items x 16 = 512 lanes: the last 8 masked)

__device__ __forceinline__ float b0_silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f)); }

// Pure vector arithmetic has no place of its own in the compiler's instruction order: instruction selection lets it sink to its first
// use -- for the staging / epilogue values below that is the LDS store BEHIND the next barrier, where every wave of the CU would do
// it at once with no MFMA to hide under.  An empty volatile asm that "modifies" the value is ordered with the scheduling fences and the
// barriers, so the arithmetic that feeds it stays in front of it: between the MFMAs of the tap where the source put it.
__device__ __forceinline__ void b0_pin(f32x4& v) {
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
}

struct B0Tile { int n, oy0, ox0; };

// IN_NHWC: x is [N][H][W][32] float32 (the decoder's block); else planes of 4 channels [N][8][H*W][4] (the encoder's block).
// O4: the block's output feeds only the network's 1x1 output projection (+ global residual, de-normalisation): computed here, the
// 32-channel tensor is never stored; else the output goes to split planes (raw), what the stride-2 layer and the decoder read.
template <bool IN_NHWC, bool O4>
__global__ __launch_bounds__(512) void block0_fused_kernel(const YondBlock0Desc d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* img = smem;
    char* wl = smem + B0_IMG_BYTES;
    float* film = (float*)(smem + B0_FILM_OFF);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, kg = lane >> 4;
    const int HW = d.H * d.W;
    const int ntx = (d.W + B0_TW - 1) / B0_TW, nty = (d.H + B0_TH - 1) / B0_TH;
    const int total = ntx * nty * d.N;
    const int G = gridDim.x;
    const int lslot = (G % 8 == 0) ? (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : blockIdx.x;   // XCD-contiguous runs of tiles
    auto tile_at = [&](int t) {
        B0Tile T;
        const int tx = t % ntx;
        int b = t / ntx;
        const int ty = b % nty;
        T.n = b / nty;
        T.oy0 = ty * B0_TH;
        T.ox0 = tx * B0_TW;
        return T;
    };

    // ---- both layers' weights: global -> LDS once ----
    {
        const f32x4* g1 = (const f32x4*)d.w1;
        const f32x4* g2 = (const f32x4*)d.w2;
#pragma unroll
        for (int k = 0; k < B0_W_BYTES / 16 / 512; ++k) {
            const int i = tid + k * 512;
            *(f32x4*)(wl + i * 16) = g1[i];
            *(f32x4*)(wl + B0_W_BYTES + i * 16) = g2[i];
        }
        static_assert(B0_W_BYTES % (16 * 512) == 0 || true, "");
        for (int i = (B0_W_BYTES / 16 / 512) * 512 + tid; i < B0_W_BYTES / 16; i += 512) {
            *(f32x4*)(wl + i * 16) = g1[i];
            *(f32x4*)(wl + B0_W_BYTES + i * 16) = g2[i];
        }
    }

    // ---- the thread's staging items: (4-channel group g, region pixel) ----
    // items 0..7: pixel `tid` of the region, group k (ONE pixel geometry per thread for eight loads; the 64 lanes of a wave take 64
    // consecutive pixels of a group: contiguous along a region row in planes of 4, conflict-free 8-byte LDS writes); item 8: pixel
    // 512 + (tid & 63), group tid >> 6 (the region has 576 = 512 + 64 pixels)
    const int pixA = tid, pixB = 512 + (tid & 63), gB = tid >> 6;
    float amax = 0.0f;
    f32x4 vin[B0_NIN];
    bool okA = false, okB = false;
    auto load_input = [&](const B0Tile& T) {
        // (the pixel geometry is recomputed per tile from an opaque copy of the thread index: a handful of integer operations;
        // kept across the tile loop it costs registers the MFMA stretches need, and the compiler's answer is scratch memory)
        int tq = tid;
        asm volatile("" : "+v"(tq));
        const int pA = tq, pB = 512 + (tq & 63);
        const int iyA = pA / B0_P, ixA = pA - iyA * B0_P, iyB = pB / B0_P, ixB = pB - iyB * B0_P;
        const int gyA = T.oy0 - 2 + iyA, gxA = T.ox0 - 2 + ixA, gyB = T.oy0 - 2 + iyB, gxB = T.ox0 - 2 + ixB;
        okA = gyA >= 0 && gyA < d.H && gxA >= 0 && gxA < d.W;
        okB = gyB >= 0 && gyB < d.H && gxB >= 0 && gxB < d.W;
        const int gpA = okA ? gyA * d.W + gxA : 0, gpB = okB ? gyB * d.W + gxB : 0;       // (outside the image: a valid address, zeroed at the write)
        // (a wave-uniform 64-bit base per load + a 32-bit byte offset per lane: one address register per lane instead of two)
        const char* xn = (const char*)d.x + (size_t)T.n * (size_t)HW * 128;
        const unsigned hw16 = (unsigned)HW * 16u;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if constexpr (IN_NHWC) vin[k] = *(const f32x4*)(xn + 16 * k + (unsigned)gpA * 128u);
            else vin[k] = *(const f32x4*)(xn + (size_t)k * hw16 + (unsigned)gpA * 16u);
        }
        if constexpr (IN_NHWC) vin[8] = *(const f32x4*)(xn + ((unsigned)gpB * 128u + 16u * (unsigned)gB));
        else vin[8] = *(const f32x4*)(xn + ((size_t)gB * hw16 + (size_t)((unsigned)gpB * 16u)));
    };
    // staging in two steps: `convert` (SiLU, zero padding, split: vector ALU only, the four floats of an item replaced in place by
    // its packed (h, l) halves) is placed between the MFMAs of conv2, `write_input` (two 8-byte LDS stores per item) behind the barrier
    auto convert = [&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        const bool ok = k < 8 ? okA : okB;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ok ? b0_silu(vin[k][e]) : 0.0f;          // conv zero padding
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
        const b0_f16x4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        const b0_f16x4 l = {(_Float16)((v[0] - (float)h[0]) * 2048.0f), (_Float16)((v[1] - (float)h[1]) * 2048.0f),
                            (_Float16)((v[2] - (float)h[2]) * 2048.0f), (_Float16)((v[3] - (float)h[3]) * 2048.0f)};
        const f32x2 hb = __builtin_bit_cast(f32x2, h), lb = __builtin_bit_cast(f32x2, l);
        vin[k] = f32x4{hb[0], hb[1], lb[0], lb[1]};
        b0_pin(vin[k]);
    };
    auto write_input = [&]() {
#pragma unroll
        for (int k = 0; k < B0_NIN; ++k) {
            const int g = k < 8 ? k : gB, pix = k < 8 ? pixA : pixB;
            char* sp = img + (((g >> 1) * 2) * B0_PU + pix) * 16 + (g & 1) * 8;       // the h half; the l half one plane further
            *(f32x2*)sp = f32x2{vin[k][0], vin[k][1]};
            *(f32x2*)(sp + B0_PU * 16) = f32x2{vin[k][2], vin[k][3]};
        }
    };

    // ---- fragment addresses ----
    // pixel fragment: plane (kg, part), unit = flat pixel index; weight fragment: ((tap, part), kg) x 32 channels
    const char* xfrag = img + ((kg * 2) * B0_PU + c) * 16;
    char* mid_base = img + (((kg >> 1) * 2) * B0_PU + 16 * wave + c) * 16 + (kg & 1) * 8;
    const char* wfrag = wl + (kg * 32 + c) * 16;
    // conv1: item j of this wave covers mid pixels 16 (wave + 8 j) .. + 15; conv2: item j = (row m, half hx) = wave + 8 j

    B0Tile T = tile_at(lslot < total ? lslot : 0);
    int t_cur = lslot;
    if (t_cur < total) {
        load_input(T);
        static_for<0, B0_NIN>([&](auto kc) { convert(kc); });
        write_input();
    }
    int dbg_tile = 0;
    (void)dbg_tile;
    for (; t_cur < total; t_cur += G) {
        BDBG(0);
        if (wave < 4 && lane < 32) {                                   // the tile's FiLM vectors (per image): fetched from LDS where they are used
            const int wv = __builtin_amdgcn_readfirstlane(wave);       // (wave-uniform choice of the vector: scalar registers)
            const float* src = wv == 0 ? d.s1 : wv == 1 ? d.t1 : wv == 2 ? d.s2 : d.t2;
            film[wv * 32 + lane] = src ? src[(d.ebatch ? T.n * 32 : 0) + lane] : ((wv & 1) ? 0.0f : 1.0f);
        }
        __syncthreads();                                               // the input region (and, the first time, the weights) is in LDS
        BDBG(1);
        // ================= conv1 over the flat mid region, in two halves of two items =================
        // The epilogue of conv1 (fold, FiLM, SiLU, zero padding, split: ~270 vector-ALU cycles per item and 16-channel block) is vector
        // work that nothing else of the workgroup could hide once all waves sit between the two barriers.  So the wave's four items run as
        // two halves: the first half's epilogue ARITHMETIC rides between the MFMAs of the second half (its results wait in registers as
        // packed halves), and only the second half's is left for the stretch between the barriers.  Price: the weight fragments of a tap
        // are read from LDS once per half.
        f32x4 a0[2][2], a1[2][2];                                      // the half in flight
        f32x4 mq[4][2];                                                // finished items: (h0h1, h2h3, l0l1, l2l3) as raw bits
        auto epi1_piece = [&](int j, int mb, const f32x4& p0, const f32x4& p1) __attribute__((always_inline)) {
            int lq = lane;                                             // (opaque: see load_input)
            asm volatile("" : "+v"(lq));
            const int o = 16 * (wave + 8 * j) + (lq & 15);             // the lane's flat mid pixel of item j
            const int my = o / B0_P, mx = o - my * B0_P;
            const int gy = T.oy0 - 1 + my, gx = T.ox0 - 1 + mx;
            const bool ok = o < B0_MID_PIX && mx < B0_TW + 2 && gy >= 0 && gy < d.H && gx >= 0 && gx < d.W;
            const f32x4 es = *(const f32x4*)(film + mb * 16 + 4 * kg), et = *(const f32x4*)(film + 32 + mb * 16 + 4 * kg);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float y = fmaf(p1[e], 1.0f / 2048.0f, p0[e]);
                y = b0_silu(fmaf(y, es[e], et[e]));
                v[e] = ok ? y : 0.0f;                                  // outside the image: conv2's zero padding
            }
            amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
            const b0_f16x4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            const b0_f16x4 l = {(_Float16)((v[0] - (float)h[0]) * 2048.0f), (_Float16)((v[1] - (float)h[1]) * 2048.0f),
                                (_Float16)((v[2] - (float)h[2]) * 2048.0f), (_Float16)((v[3] - (float)h[3]) * 2048.0f)};
            const f32x2 hb = __builtin_bit_cast(f32x2, h), lb = __builtin_bit_cast(f32x2, l);
            mq[j][mb] = f32x4{hb[0], hb[1], lb[0], lb[1]};
            b0_pin(mq[j][mb]);
        };
        f32x4 h0[2][2], h1[2][2];                                      // the first half's accumulators while the second half runs
        static_for<0, 2>([&](auto hc) __attribute__((always_inline)) {
            constexpr int hf = decltype(hc)::value;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    a0[j][mb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                    a1[j][mb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                }
            // (ONE set of weight fragments: the next tap's are requested behind the last MFMA that reads this tap's -- their latency is
            // the sibling wave's turn on the matrix pipe; a second set costs 16 registers the kernel does not have)
            b0_f16x8 wf[2][2];
            auto load_w = [&](int tap) __attribute__((always_inline)) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int p = 0; p < 2; ++p) wf[mb][p] = *(const b0_f16x8*)(wfrag + (((tap * 2 + p) * 4) * 32 + mb * 16) * 16);
            };
            load_w(0);
            static_for<0, 9>([&](auto tc) __attribute__((always_inline)) {
                constexpr int tap = decltype(tc)::value;
                constexpr int dy = tap / 3, dx = tap - dy * 3;
                b0_f16x8 xf[2][2];
                auto load_x = [&](int j, int s_) __attribute__((always_inline)) {
                    const int po = (16 * (wave + 8 * (2 * hf + j)) + dy * B0_P + dx) * 16;
                    xf[s_][0] = *(const b0_f16x8*)(xfrag + po);
                    xf[s_][1] = *(const b0_f16x8*)(xfrag + po + B0_PU * 16);
                };
                load_x(0, 0);
                load_x(1, 1);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const b0_f16x8 xh = xf[j][0], xl = xf[j][1];
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) a0[j][mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[mb][0], xh, a0[j][mb], 0, 0, 0);
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) a1[j][mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[mb][0], xl, a1[j][mb], 0, 0, 0);
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) a1[j][mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[mb][1], xh, a1[j][mb], 0, 0, 0);
                }
                if constexpr (tap < 8) load_w(tap + 1);
                // second half: one (item, block) piece of the first half's epilogue per two taps
                if constexpr (hf == 1 && (tap & 1) == 1) {
                    constexpr int pc = tap / 2;                          // 0..3
                    epi1_piece(pc >> 1, pc & 1, h0[pc >> 1][pc & 1], h1[pc >> 1][pc & 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (hf == 0) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) {
                        h0[j][mb] = a0[j][mb];
                        h1[j][mb] = a1[j][mb];
                    }
            }
        });
        BDBG(2);
        __syncthreads();                                               // every wave has read the input region: mid may overwrite it
        BDBG(3);
        // ================= epilogue 1: SiLU(FiLM(conv1)) -> split halves -> mid (zero outside the image: conv2's padding) =================
        auto mid_write = [&](int j, int mb) __attribute__((always_inline)) {
            // channels mb 16 + 4 kg .. + 3: plane (channel / 8) = 2 mb + (kg >> 1), second half of the unit for odd kg
            // (ONE lane-dependent base, the item and the block as constants: immediate offsets of the stores)
            char* mp = mid_base + (mb * 4 * B0_PU + 128 * j) * 16;
            *(f32x2*)mp = f32x2{mq[j][mb][0], mq[j][mb][1]};
            *(f32x2*)(mp + B0_PU * 16) = f32x2{mq[j][mb][2], mq[j][mb][3]};
        };
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) mid_write(j, mb);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                epi1_piece(2 + j, mb, a0[j][mb], a1[j][mb]);
                mid_write(2 + j, mb);
            }
        BDBG(4);
        // ---- memory operations that land under conv2: the next tile's input, this tile's residual ----
        const int t_next = t_cur + G;
        const B0Tile Tn = tile_at(t_next < total ? t_next : t_cur);
        load_input(Tn);                                                // (the last tile loads itself again: harmless, never written)
        f32x4 rr[3][2];
        float xq[O4 ? 3 : 1];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int it = wave + 8 * j;
            const int oy = T.oy0 + (it >> 1), ox = T.ox0 + (it & 1) * 16 + c;
            const bool ok = oy < d.H && ox < d.W;
            const unsigned gp = ok ? (unsigned)(oy * d.W + ox) : 0u;
            const char* xn = (const char*)d.x + (size_t)T.n * (size_t)HW * 128;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                if constexpr (IN_NHWC) rr[j][mb] = *(const f32x4*)(xn + (gp * 128u + (unsigned)(mb * 64 + 16 * kg)));
                else rr[j][mb] = *(const f32x4*)(xn + (size_t)(mb * 4) * ((size_t)HW * 16) + ((size_t)kg * ((size_t)HW * 16) + (size_t)(gp * 16u)));
            }
            if constexpr (O4) xq[j] = d.out4_x ? d.out4_x[((long long)T.n * HW + (ok ? oy * d.W + ox : 0)) * 4 + kg] : 0.0f;
        }
        BDBG(5);
        __syncthreads();                                               // mid is complete
        BDBG(6);
        // ================= conv2: 12 rows x 2 halves of 16 pixels =================
        f32x4 b0a[3][2], b1a[3][2];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                b0a[j][mb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                b1a[j][mb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            }
        {
            b0_f16x8 wf[2][2];
            auto load_w = [&](int tap) __attribute__((always_inline)) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int p = 0; p < 2; ++p) wf[mb][p] = *(const b0_f16x8*)(wfrag + B0_W_BYTES + (((tap * 2 + p) * 4) * 32 + mb * 16) * 16);
            };
            load_w(0);
            static_for<0, 9>([&](auto tc) __attribute__((always_inline)) {
                constexpr int tap = decltype(tc)::value;
                constexpr int dy = tap / 3, dx = tap - dy * 3;
                b0_f16x8 xf[2][2];
                auto load_x = [&](int j, int s_) __attribute__((always_inline)) {
                    const int it = wave + 8 * j;
                    const int po = (((it >> 1) + dy) * B0_P + (it & 1) * 16 + dx) * 16;
                    xf[s_][0] = *(const b0_f16x8*)(xfrag + po);
                    xf[s_][1] = *(const b0_f16x8*)(xfrag + po + B0_PU * 16);
                };
                load_x(0, 0);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if (j < 2) load_x(j + 1, (j + 1) & 1);
                    const b0_f16x8 xh = xf[j & 1][0], xl = xf[j & 1][1];
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) b0a[j][mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[mb][0], xh, b0a[j][mb], 0, 0, 0);
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) b1a[j][mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[mb][0], xl, b1a[j][mb], 0, 0, 0);
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) b1a[j][mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[mb][1], xh, b1a[j][mb], 0, 0, 0);
                }
                if constexpr (tap < 8) load_w(tap + 1);
                // the next tile's input arrives during the first taps; its SiLU + split (vector ALU only) rides between the MFMAs of the
                // later ones: items 0-1 in tap 4, 2-3 in tap 5, 4-5 in tap 6, 6-7 in tap 7, 8 in tap 8
                if constexpr (tap >= 4) {
                    convert(IntC<2 * (tap - 4)>{});
                    if constexpr (tap < 8) convert(IntC<2 * (tap - 4) + 1>{});
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        BDBG(7);
        // ================= epilogue 2: FiLM2 / bias + residual -> HBM =================
        {
            // (the tile's coordinates made opaque here: the store addresses are formed now, not carried in registers through conv2)
            int e_n = T.n, e_oy0 = T.oy0, e_ox0 = T.ox0;
            asm volatile("" : "+s"(e_n), "+s"(e_oy0), "+s"(e_ox0));
            const int PSo = YOND_SP_PLANE_UNITS(d.H, d.W);
            f32x4 es[2], et[2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                es[mb] = *(const f32x4*)(film + 64 + mb * 16 + 4 * kg);
                et[mb] = *(const f32x4*)(film + 96 + mb * 16 + 4 * kg);
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int it = wave + 8 * j;
                const int oy = e_oy0 + (it >> 1), ox = e_ox0 + (it & 1) * 16 + c;
                const bool ok = oy < d.H && ox < d.W;
                f32x4 v[2];
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float y = fmaf(b1a[j][mb][e], 1.0f / 2048.0f, b0a[j][mb][e]);
                        v[mb][e] = fmaf(y, es[mb][e], et[mb][e]) + rr[j][mb][e];
                    }
                if constexpr (O4) {
                    // 1x1 projection 32 -> 4 (archs/Unet.py:463-468): the lane's 8 channels, then the pixel's four lanes (kg) by xor
                    // shuffles over lanes 16 and 32 apart; lane kg finishes output component kg: + bias + x / ub, * ub
                    float o[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float t = 0.0f;
#pragma unroll
                        for (int mb = 0; mb < 2; ++mb) {
                            const f32x4 wv = *(const f32x4*)(d.out4_w + q * 32 + mb * 16 + 4 * kg);
#pragma unroll
                            for (int e = 0; e < 4; ++e) t = fmaf(v[mb][e], wv[e], t);
                        }
                        t += __shfl_xor(t, 16);
                        t += __shfl_xor(t, 32);
                        o[q] = t;
                    }
                    float t = kg == 0 ? o[0] : kg == 1 ? o[1] : kg == 2 ? o[2] : o[3];
                    const float ubv = d.out4_ub ? d.out4_ub[e_n] : 1.0f;
                    if (d.out4_b) t += d.out4_b[kg];
                    if (d.out4_x) t += d.out4_ub ? xq[j] / ubv : xq[j];
                    if (d.out4_ub) t *= ubv;
                    if (ok) d.out4_dst[((long long)e_n * HW + oy * d.W + ox) * 4 + kg] = t;
                } else {
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) {
                        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[mb][0]), fabsf(v[mb][1])), fmaxf(fabsf(v[mb][2]), fabsf(v[mb][3]))));
                        const b0_f16x4 h = {(_Float16)v[mb][0], (_Float16)v[mb][1], (_Float16)v[mb][2], (_Float16)v[mb][3]};
                        const b0_f16x4 l = {(_Float16)((v[mb][0] - (float)h[0]) * 2048.0f), (_Float16)((v[mb][1] - (float)h[1]) * 2048.0f),
                                            (_Float16)((v[mb][2] - (float)h[2]) * 2048.0f), (_Float16)((v[mb][3] - (float)h[3]) * 2048.0f)};
                        // split planes [n][C/16 = 2][channel half][part][PSo]: channels mb 16 + 4 kg ..: chunk mb, half kg >> 1
                        char* pp = (char*)d.dst + (size_t)((e_n * 2 + mb) * 4) * (size_t)PSo * 16 + ((size_t)((kg >> 1) * 2) * (size_t)PSo * 16 + (size_t)((unsigned)(oy * d.W + ox) * 16u + (unsigned)((kg & 1) * 8)));
                        if (ok) {
                            *(b0_f16x4*)pp = h;
                            *(b0_f16x4*)(pp + (size_t)PSo * 16) = l;
                        }
                    }
                }
            }
        }
        BDBG(8);
        __syncthreads();                                               // every wave has read mid: the next input may overwrite it
        BDBG(9);
        write_input();                                                 // (the last tile writes its own input again: nobody reads it)
        BDBG(10);
        T = Tn;
        ++dbg_tile;
    }
    if (d.status && !(amax <= 65504.0f)) atomicOr(d.status, 1);      // a staged value left fp16's range (or is NaN: the comparison fails)
}

}  // namespace

// Host side: OIHW float32 [32][32][3][3] -> the kernel's LDS image [tap][part h, l][k-group of 8 input channels][32 output channels] x 8 halves.
extern "C" int yond_pack_block0_weight_f32(const float* w_oihw, int cout, int cin, void* out) {
    if (!w_oihw || !out || cout < 1 || cout > 32 || cin < 1 || cin > 32) return YOND_EINVAL;
    _Float16* o = (_Float16*)out;
    for (int tap = 0; tap < 9; ++tap)
        for (int part = 0; part < 2; ++part)
            for (int kgp = 0; kgp < 4; ++kgp)
                for (int co = 0; co < 32; ++co)
                    for (int j = 0; j < 8; ++j) {
                        const int ci = 8 * kgp + j;
                        const float w = (co < cout && ci < cin) ? w_oihw[((size_t)co * cin + ci) * 9 + tap] : 0.0f;
                        if (!(fabsf(w) <= 65504.0f)) return YOND_EUNSUPPORTED;
                        const _Float16 h = (_Float16)w;
                        const _Float16 l = (_Float16)((w - (float)h) * 2048.0f);
                        o[((((size_t)tap * 2 + part) * 4 + kgp) * 32 + co) * 8 + j] = part ? l : h;
                    }
    return YOND_OK;
}

extern "C" int yond_block0_fused_f32(const YondBlock0Desc* dp, void* stream) {
    if (!dp) return YOND_EINVAL;
    const YondBlock0Desc& d = *dp;
    if (!d.x || !d.w1 || !d.w2 || d.N < 1 || d.H < 1 || d.W < 1) return YOND_EINVAL;
    if (d.in_fmt != YOND_FMT_PLANES4 && d.in_fmt != YOND_FMT_NHWC_F32) return YOND_EINVAL;
    const bool o4 = d.out4_dst != nullptr;
    if (o4 ? !d.out4_w : !d.dst) return YOND_EINVAL;
    if (((uintptr_t)d.x | (uintptr_t)d.w1 | (uintptr_t)d.w2 | (uintptr_t)d.dst) & 15) return YOND_EINVAL;
    if ((double)d.N * d.H * d.W * 32.0 >= 2147483648.0 * 4.0) return YOND_EUNSUPPORTED;     // (64-bit offsets inside; a sanity bound)
    if ((double)YOND_SP_PLANE_UNITS(d.H, d.W) * 16.0 * 8.0 * d.N >= 9.0e18) return YOND_EUNSUPPORTED;
    const int ntx = (d.W + B0_TW - 1) / B0_TW, nty = (d.H + B0_TH - 1) / B0_TH;
    const long long total = (long long)ntx * nty * d.N;
    if (total >= 0x7fffffffLL) return YOND_EUNSUPPORTED;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return YOND_EINVAL;
        ncu = p.multiProcessorCount;
    }
    const int grid = (int)(total < ncu ? total : ncu);
    hipStream_t st = (hipStream_t)stream;
#define B0_LAUNCH(NH, OF)                                                                                                   \
    do {                                                                                                                    \
        static bool attr = false;                                                                                           \
        if (!attr) {                                                                                                        \
            hipError_t e = hipFuncSetAttribute((const void*)block0_fused_kernel<NH, OF>, hipFuncAttributeMaxDynamicSharedMemorySize, B0_SMEM); \
            if (e != hipSuccess) return (int)e;                                                                             \
            attr = true;                                                                                                    \
        }                                                                                                                   \
        hipLaunchKernelGGL((block0_fused_kernel<NH, OF>), dim3(grid), dim3(512), B0_SMEM, st, d);                           \
    } while (0)
    if (d.in_fmt == YOND_FMT_NHWC_F32) {
        if (o4) B0_LAUNCH(true, true);
        else B0_LAUNCH(true, false);
    } else {
        if (o4) B0_LAUNCH(false, true);
        else B0_LAUNCH(false, false);
    }
#undef B0_LAUNCH
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}
#endif  // YOND_EXPERIMENTS
