// K2w: 3x3 stride-1 convolutions by Winograd F(2x2, 3x3) on the fp32 matrix cores (v_mfma_f32_16x16x4_f32).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A      per 2x2 output patch, 4x4 input patch d, 3x3 filter g
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
//
// 16 multiplications per patch and channel pair instead of 36: the sixteen "planes" (xi, nu) of the transformed
// domain are sixteen independent GEMMs  M_p[cout][patch] = sum_cin U_p[cout][cin] * V_p[cin][patch]  that run
// on the MFMA unit; the transforms are additions on the vector ALU.  fp32 throughout (products and sums are the
// MFMA's fp32 FMAs); the result differs from a direct convolution only by the rounding order (a few 1e-7
// relative, tests/test_hip_conv.py).
//
// Persistent workgroups (one per CU, 512 threads = two waves per SIMD) walk (tile, 8-channel chunk) steps; a tile is
// 8 x 32 output pixels = 4 x 16 patches x 64 output channels; wave (mb, cq) owns patches [32 mb, 32 mb + 32) x
// channels [16 cq, 16 cq + 16) of ALL sixteen planes as 16 x 2 accumulators of v_mfma_f32_16x16x4_f32 (128
// registers), so the output transform is lane-local: lane = patch (lane & 15), its 4 registers = 4 consecutive
// channels (weights are the A operand as in conv.hip).  The two waves of a SIMD run the two halves of a step --
// the 64 MFMAs, and everything else -- in opposite order, so the matrix pipe always has one of them.
// Per step, in one instruction stream with ONE barrier at its end:
//     64 MFMAs of step s on V[s&1], U[s&1]
//     || input transform raw[(s+1)&1] -> V[(s+1)&1]   || (SiLU +) ds_write of the registers loaded a step ago -> raw[s&1]
//     || issue of the global loads of step s+3         || LDS-DMA of the transformed weights U of step s+1
// (U is packed in LDS order by yond_pack_conv_wino_weight_f32.)
//
// LDS images:
//   raw [2][kh][10 rows][parity][17][4]  kh = 4-channel half of the chunk; even / odd columns apart, so the sixteen
//                                      lanes of a ds_read_b128 (consecutive patches, column stride 2) are contiguous
//   V   [2][16 planes][kh][64 patches][4]      U   [2][16 planes][kh][TN][4]
// MFMA kk (0, 1) of a chunk takes channel 2*kq + kk from lane group kq = lane>>4, i.e. float2 (kq & 1) of the
// 16-byte slot of half kh = kq >> 1 (any pairing works as long as U and V agree).
#include "common.h"

#ifndef WINO_ABL
#define WINO_ABL 0      // timing-only ablations: 1 no input loads, 2 no weight DMA, 4 no epilogue, 8 no transform, 16 no MFMA
#endif

// Tile shapes: <TN 64, TH 8, KC 8, VB 2>   8 x 32 px x 64 channels, 8-channel chunks  (layers with Cout % 64 == 0)
//              <TN 32, TH 16, KC 8, VB 1> 16 x 32 px x 32 channels, 8-channel chunks (the 32-channel level-0 layers: all eight
//                                         waves still own a full 32-patch x 16-channel x 16-plane block).  Its V image is
//                                         64 KB, so there is ONE of it: MFMA phase | barrier | transform phase.  With fp32
//                                         MFMA and VALU serial anyway this costs only the latency overlap, and the steps are
//                                         twice as long as with 4-channel chunks (the per-step fixed cost was the problem)
//              <TN 32, TH 16, KC 4, VB 2> the same tile with 4-channel chunks and two V images (kept for comparison)
template <int TN, int TH_, int KC_, int VB_>
struct WinoCfg {
    static constexpr int VB = VB_;                               // V buffers: 2 = transform of step s+1 runs beside the MFMAs
                                                                 // of step s; 1 = MFMA phase | barrier | transform phase
    static constexpr int KC = KC_;
    static constexpr int KH = KC / 4;                            // 16-byte channel groups per chunk (1 or 2)
    static constexpr int TH = TH_, TW = 32;
    static constexpr int NPR = TH / 2;                           // patch rows (4 or 8), 16 patches each
    static constexpr int NP = NPR * (TW / 2);                    // patches per tile (64 or 128)
    static constexpr int MB = NP / 32;                           // 32-patch blocks (2 or 4)
    static constexpr int CQ = TN / 16;                           // 16-channel quarters (4 or 2); MB * CQ = 8 waves
    static constexpr int IH = TH + 2, IW = TW + 2, HALF = IW / 2;
    static constexpr int RAW_FLOATS = KH * IH * 2 * HALF * 4;
    static constexpr int V_FLOATS = 16 * KH * NP * 4;
    static constexpr int U_FLOATS = 16 * KH * TN * 4;
    static constexpr int NITEM = IH * IW * KH;                   // (pixel, kh) 16-byte items
    static constexpr int NT = 512;                               // threads per workgroup
    static constexpr int NIN = (NITEM + NT - 1) / NT;
    static constexpr int NUT = U_FLOATS / 4 / NT;                // LDS-DMA instructions per thread and weight slice
    static constexpr int RAWB_FLOATS = RAW_FLOATS + 4;           // + a dummy slot for items past the end of the tile
    static constexpr int TPT = KH * NP * 4 / NT;                 // transform tasks per thread (1 or 2)
    static constexpr int SMEM_BYTES = (VB * V_FLOATS + 2 * U_FLOATS + 2 * RAWB_FLOATS) * 4;
    static_assert(MB * CQ == 8 && KH * NP * 4 == TPT * NT && NUT >= 1 && U_FLOATS / 4 % NT == 0, "wave / task maps below");
};

__device__ __forceinline__ float wino_silu(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f));
}

// wait for this wave's LDS traffic only (not for global loads still in flight), then the workgroup barrier
__device__ __forceinline__ void barrier_lds_only() { if (!(WINO_ABL & 32)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// ... and for all but the N most recent vector-memory operations (the loads issued last stay in flight)
template <int N>
__device__ __forceinline__ void barrier_lds_keep_loads() { if (!(WINO_ABL & 32)) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory"); }

template <int TN, int TH, int KC, int VB, bool PRE>
__global__ __launch_bounds__(512) void conv_wino_kernel(const YondConvDesc d) {
    using C = WinoCfg<TN, TH, KC, VB>;
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_v = smem;
    float* s_u = smem + VB * C::V_FLOATS;
    float* s_raw = s_u + 2 * C::U_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lj = lane & 15, kq = lane >> 4;
    const int mb = wave % C::MB, cq = wave / C::MB;

    const int nct = d.Cout / TN;
    const bool computes = d.Cout > 0;                        // always true; opaque to the compiler (keeps the MFMA stretch
                                                             // a block of its own: merged with its neighbours it spilled)
    const int ntx = (d.Wo + 31) / 32, nty = (d.Ho + TH - 1) / TH;
    const int tiles_per_img = nct * ntx * nty;
    const int total = tiles_per_img * d.N;
    const int G = gridDim.x;
    const int lslot = (G % 8 == 0) ? (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : blockIdx.x;   // XCD-contiguous runs
    const int Cin = d.C0 + d.C1;
    const int nchunk = Cin / KC;
    const int my_kh = C::KH == 2 ? (tid & 1) : 0;

    int raw_lds[C::NIN];
#pragma unroll
    for (int k = 0; k < C::NIN; ++k) {
        const int it = tid + k * C::NT;
        const int pix = it / C::KH;
        const int py = pix / C::IW, px = pix % C::IW;
        // items past the end of the tile go to a dummy slot behind the image (never read): no branch in the step body
        raw_lds[k] = it < C::NITEM ? ((((my_kh * C::IH + py) * 2 + (px & 1)) * C::HALF + (px >> 1)) * 4) : C::RAW_FLOATS;
    }

    struct Tile {
        int ct, n, ox0, oy0;
        int goff[C::NIN];
    };
    auto decode = [&](int t, Tile& T) {
        const int n = t / tiles_per_img;
        int b = t - n * tiles_per_img;
        T.n = n;
        T.ct = b % nct;
        b /= nct;
        const int tx = b % ntx, ty = b / ntx;
        T.ox0 = tx * 32;
        T.oy0 = ty * TH;
#pragma unroll
        for (int k = 0; k < C::NIN; ++k) {
            const int it = tid + k * C::NT;
            const int pix = it / C::KH;
            const int py = pix / C::IW, px = pix % C::IW;
            const int gy = T.oy0 - 1 + py, gx = T.ox0 - 1 + px;
            T.goff[k] = (it < C::NITEM && gy >= 0 && gy < d.H && gx >= 0 && gx < d.W) ? ((n * d.H + gy) * d.W + gx) : -1;
        }
    };

    // Two register sets for the raw loads: set s & 1 is written out as raw(s+2) in step s and refilled at once with
    // the loads of step s+4, so a load has two steps to arrive.
    f32x4 vin[2][C::NIN];
    if (WINO_ABL & 1) {
#pragma unroll
        for (int k = 0; k < C::NIN; ++k) { const f32x4 z = {0.5f, 0.5f, 0.5f, 0.5f}; vin[0][k] = z; vin[1][k] = z; }
    }
    unsigned vin_ok[2] = {0, 0};
    auto issue_loads = [&](auto pc, const Tile& T, int ch) {
        constexpr int P = decltype(pc)::value;
        const int c0 = ch * KC;
        const float* src;
        int Cs, cc;
        if (c0 < d.C0) { src = d.src0; Cs = d.C0; cc = c0; }
        else { src = d.src1; Cs = d.C1; cc = c0 - d.C0; }
        vin_ok[P] = 0;
#pragma unroll
        for (int k = 0; k < C::NIN; ++k) {
            const bool ok = T.goff[k] >= 0;                    // outside the image: read pixel 0, zeroed at the LDS write
            if (!(WINO_ABL & 1)) vin[P][k] = *(const f32x4*)(src + (size_t)(ok ? T.goff[k] : 0) * Cs + cc + my_kh * 4);
            vin_ok[P] |= (ok ? 1u : 0u) << k;
        }
    };
    // The weight slice goes global -> LDS by LDS-DMA, written as inline assembly: the compiler orders every later
    // LDS access behind a builtin DMA with s_waitcnt vmcnt(0), i.e. it would expose the whole L2 latency once per
    // step.  Completion is awaited explicitly by the end-of-step barrier (barrier_lds_keep_loads).
    auto issue_weights = [&](const Tile& T, int ch, float* ubuf) {
        const float* wsrc = d.wpk + ((size_t)T.ct * nchunk + ch) * C::U_FLOATS;        // wave-uniform
        const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)ubuf;
#pragma unroll
        for (int k = 0; k < C::NUT; ++k) {
            const int it = tid + k * C::NT;
            const unsigned lds_wave = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(it - lane) * 16u);
            const unsigned voff = (unsigned)it * 16u;
            if (!(WINO_ABL & 2))
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_wave), "v"(voff), "s"(wsrc) : "memory");
        }
    };
    auto write_raw = [&](auto pc, float* rawbuf) {
        constexpr int P = decltype(pc)::value;
#pragma unroll
        for (int k = 0; k < C::NIN; ++k) {
            f32x4 v = vin[P][k];
            if (PRE) { v[0] = wino_silu(v[0]); v[1] = wino_silu(v[1]); v[2] = wino_silu(v[2]); v[3] = wino_silu(v[3]); }
            const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
            *(f32x4*)(rawbuf + raw_lds[k]) = ((vin_ok[P] >> k) & 1u) ? v : z;                     // conv zero padding
        }
    };
    // input transform: (patch, kh, channel) scalar tasks, TPT per thread -- task i of wave w is "virtual wave" w + 8 i: patch
    // row (w + 8 i) % NPR of half kh = (w + 8 i) / NPR; lane = 4 * patch column + channel: every LDS access is 256 contiguous
    // bytes per wave
    const int t_pc = lane >> 2, t_e = lane & 3;
    constexpr int PLV = C::KH * C::NP * 4;                     // floats per plane of V
    constexpr int PLU = C::KH * TN * 4;                        // floats per plane of U
    float traw[C::TPT][4][4];
    auto transform_load = [&](const float* rawbuf) {
#pragma unroll
        for (int i = 0; i < C::TPT; ++i) {
            const int vw = wave + 8 * i;
            const int t_kh = vw / C::NPR, t_pr = vw % C::NPR;
            const float* src = rawbuf + (((t_kh * C::IH + 2 * t_pr) * 2) * C::HALF + t_pc) * 4 + t_e;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) traw[i][r][c] = src[((r * 2 + (c & 1)) * C::HALF + (c >> 1)) * 4];
        }
    };
    auto transform_store = [&](float* vbuf) {
#pragma unroll
        for (int i = 0; i < C::TPT; ++i) {
            const int vw = wave + 8 * i;
            const int t_kh = vw / C::NPR, t_pr = vw % C::NPR;
            float w[4][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {                      // B^T d (rows)
                w[0][c] = traw[i][0][c] - traw[i][2][c];
                w[1][c] = traw[i][1][c] + traw[i][2][c];
                w[2][c] = traw[i][2][c] - traw[i][1][c];
                w[3][c] = traw[i][1][c] - traw[i][3][c];
            }
            float* o = vbuf + ((t_kh * C::NP) + t_pr * 16 + t_pc) * 4 + t_e;
#pragma unroll
            for (int q = 0; q < 4; ++q) {                      // (B^T d) B (columns)
                o[(q * 4 + 0) * PLV] = w[q][0] - w[q][2];
                o[(q * 4 + 1) * PLV] = w[q][1] + w[q][2];
                o[(q * 4 + 2) * PLV] = w[q][2] - w[q][1];
                o[(q * 4 + 3) * PLV] = w[q][1] - w[q][3];
            }
        }
    };

    f32x4 acc[16][2];                                          // [plane][16-patch half of the wave's 32 patches]
    auto zero_acc = [&]() {
#pragma unroll
        for (int p = 0; p < 16; ++p)
#pragma unroll
            for (int t = 0; t < 2; ++t) { const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f}; acc[p][t] = z; }
    };
    // All 64 MFMAs of a step as one tight stretch (the wave that shares the SIMD does its vector / LDS work meanwhile:
    // see the step loop).  The fragment loads are volatile so that they stay single ds_read_b64 (2 LDS cycles, banks
    // mod 64, conflict-free here; merged into ds_read2_b64 they cost 8 cycles and conflict 2-way) -- and therefore in
    // program order: the loop is software-pipelined by hand, plane p + FD is requested before plane p is multiplied.
    constexpr int FD = 4;
    // KC 8: lane group kq supplies channels 2 kq + kk (kk = 0, 1: two MFMAs per 16x16 tile), one float2 = 8 bytes of the
    //       16-byte slot of half kh = kq >> 1;   KC 4: channel kq, one float of the slot (one MFMA per tile)
    const int u_off = C::KH == 2 ? ((kq >> 1) * TN + cq * 16 + lj) * 4 + (kq & 1) * 2 : (cq * 16 + lj) * 4 + kq;
    const int v_off = C::KH == 2 ? ((kq >> 1) * C::NP + mb * 32 + lj) * 4 + (kq & 1) * 2 : (mb * 32 + lj) * 4 + kq;
    auto mfma_all = [&](const float* vbuf, const float* ubuf) {
        typedef const volatile __attribute__((address_space(3))) f32x2_t* lds_v2;
        typedef const volatile __attribute__((address_space(3))) float* lds_v1;
        const __attribute__((address_space(3))) float* ub = (__attribute__((address_space(3))) float*)(ubuf + u_off);
        const __attribute__((address_space(3))) float* vb = (__attribute__((address_space(3))) float*)(vbuf + v_off);
        f32x2_t uf[FD + 2], v0[FD + 2], v1[FD + 2];
        auto load = [&](auto pc) {
            constexpr int p = decltype(pc)::value;
            constexpr int sl = p % (FD + 2);
            if (WINO_ABL & 64) { const f32x2_t c = {1.0f + p, 0.5f}; uf[sl] = c; v0[sl] = c; v1[sl] = c; return; }
            if constexpr (C::KH == 2) {
                uf[sl] = *(lds_v2)(ub + p * PLU);
                v0[sl] = *(lds_v2)(vb + p * PLV);
                v1[sl] = *(lds_v2)(vb + p * PLV + 16 * 4);
            } else {
                uf[sl][0] = *(lds_v1)(ub + p * PLU);
                v0[sl][0] = *(lds_v1)(vb + p * PLV);
                v1[sl][0] = *(lds_v1)(vb + p * PLV + 16 * 4);
            }
        };
        static_for<0, FD>([&](auto pc) { load(pc); });
        __builtin_amdgcn_sched_barrier(0);
        // planes in pairs: the second MFMA into an accumulator is issued four instructions (128 cycles) after the first
        // -- back to back they would wait for the 40-cycle dependent latency of v_mfma_f32_16x16x4_f32 (32-cycle issue)
        static_for<0, 8>([&](auto qc) {
            constexpr int p = 2 * decltype(qc)::value;
            constexpr int s0 = p % (FD + 2), s1 = (p + 1) % (FD + 2);
            if constexpr (p + FD < 16) load(IntC<p + FD>{});
            if constexpr (p + FD + 1 < 16) load(IntC<p + FD + 1>{});
            __builtin_amdgcn_sched_barrier(0);                 // the machine scheduler would sink the loads to their uses
            if (!(WINO_ABL & 16)) {
                acc[p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[s0][0], v0[s0][0], acc[p][0], 0, 0, 0);           // D = U . V^T
                acc[p][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[s0][0], v1[s0][0], acc[p][1], 0, 0, 0);
                acc[p + 1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[s1][0], v0[s1][0], acc[p + 1][0], 0, 0, 0);
                acc[p + 1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[s1][0], v1[s1][0], acc[p + 1][1], 0, 0, 0);
                if constexpr (C::KH == 2) {
                    acc[p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[s0][1], v0[s0][1], acc[p][0], 0, 0, 0);
                    acc[p][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[s0][1], v1[s0][1], acc[p][1], 0, 0, 0);
                    acc[p + 1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[s1][1], v0[s1][1], acc[p + 1][0], 0, 0, 0);
                    acc[p + 1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[s1][1], v1[s1][1], acc[p + 1][1], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    // ---- output side: lane = patch (row 2 mb + t, column lj), its registers = channels 16 cq + 4 kq + (0..3) ----
    const float slope_eff = d.post_act == 2 ? d.slope : 1.0f;
    // (FiLM / bias vectors straight from global memory: a load inside a conditional block earlier in the step makes the
    // compiler fall back to s_waitcnt vmcnt(0) for the staged input, which exposes the memory latency every step)
    auto epilogue = [&](const Tile& T) {
        const int cbase = T.ct * TN + cq * 16 + 4 * kq;
        const int eoff = (d.ebatch ? T.n * d.Cout : 0) + cbase;
        const f32x4 one = {1.0f, 1.0f, 1.0f, 1.0f}, zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
        const f32x4 es = d.escale ? *(const f32x4*)(d.escale + eoff) : one;
        const f32x4 et = d.eshift ? *(const f32x4*)(d.eshift + eoff) : zero4;
        const int ox = T.ox0 + 2 * lj;
        const int oyb = T.oy0 + 4 * mb;
        const long long pbase = ((long long)(T.n * d.Ho + oyb) * d.Wo + ox) * d.Cout + T.ct * TN + cq * 16 + 4 * kq;
        // the eight residual vectors first (all in flight together; masked lanes read element 0), then the arithmetic
        f32x4 rr[2][2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const bool ok = (oyb + 2 * t + a < d.Ho) && (ox + b < d.Wo);
                    const long long off = pbase + ((long long)(2 * t + a) * d.Wo + b) * d.Cout;
                    const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
                    rr[t][a][b] = d.res ? *(const f32x4*)(d.res + (ok ? off : 0)) : z;
                }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 t0[4], t1[4];
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) {                    // A^T M
                t0[nu] = acc[0 * 4 + nu][t] + acc[1 * 4 + nu][t] + acc[2 * 4 + nu][t];
                t1[nu] = acc[1 * 4 + nu][t] - acc[2 * 4 + nu][t] - acc[3 * 4 + nu][t];
            }
            f32x4 y[2][2];
            y[0][0] = t0[0] + t0[1] + t0[2];                    // (A^T M) A
            y[0][1] = t0[1] - t0[2] - t0[3];
            y[1][0] = t1[0] + t1[1] + t1[2];
            y[1][1] = t1[1] - t1[2] - t1[3];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const bool ok = (oyb + 2 * t + a < d.Ho) && (ox + b < d.Wo);
                    const long long off = pbase + ((long long)(2 * t + a) * d.Wo + b) * d.Cout;
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = fmaf(y[a][b][e], es[e], et[e]);
                        x = x > 0.0f ? x : x * slope_eff;
                        v[e] = x + rr[t][a][b][e];
                    }
                    if (ok) *(f32x4*)(d.dst + off) = v;
                }
        }
    };

    // ---- the step pipeline: ONE barrier per step ----
    //   step s computes on V(s), U(s); meanwhile it transforms raw(s+1) -> V(s+1), writes register set s & 1 (loaded
    //   two steps ago) as raw(s+2), refills it with the global loads of step s+4 and issues the LDS-DMA of U(s+1).
    struct Cur { int tile, ch; };
    auto adv = [&](Cur c) {
        Cur n;
        const bool last = c.ch + 1 == nchunk;
        n.tile = last ? c.tile + G : c.tile;
        n.ch = last ? 0 : c.ch + 1;
        return n;
    };
    if (lslot >= total) return;
    Cur cs = {lslot, 0};                       // step being computed
    Tile cur;                                   // its tile (epilogue)
    Tile lt;                                    // tile of the load cursor
    int lt_tile = -1;
    auto loads_for = [&](auto pc, Cur c) {      // steps past the end re-read the last decoded tile (harmless)
        if (c.tile < total && c.tile != lt_tile) { decode(c.tile, lt); lt_tile = c.tile; }
        issue_loads(pc, lt, c.ch);
    };
    decode(cs.tile, cur);
    int next_ct = cur.ct;                       // channel tile of step s+1
    const Cur c1 = adv(cs), c2 = adv(c1), c3 = adv(c2);
    Cur cl = adv(c3);                           // next loads to issue: step s+4
    zero_acc();
    float* rawA = s_raw;                        // raw(s+1) at the top of step s
    float* rawB = s_raw + C::RAWB_FLOATS;       // receives raw(s+2) during step s
    float* vb = s_v;
    float* vn = VB == 2 ? s_v + C::V_FLOATS : s_v;
    float* ubf = s_u;
    float* un = s_u + C::U_FLOATS;
    // prologue: V(0), U(0), raw(1) in LDS; the loads of steps 2 and 3 in flight (register sets 0 and 1)
    loads_for(IntC<0>{}, cs);
    issue_weights(cur, 0, ubf);
    loads_for(IntC<1>{}, c1);
    write_raw(IntC<0>{}, rawB);
    __syncthreads();
    transform_load(rawB);
    transform_store(vb);
    write_raw(IntC<1>{}, rawA);
    loads_for(IntC<0>{}, c2);
    loads_for(IntC<1>{}, c3);
    __syncthreads();
    // VALU issue between the two waves of a SIMD is arbitrated by priority, then age: without this the younger half
    // (waves 4-7) only gets the slots the older half leaves, and its vector work crawls while the older half is in its
    // MFMA stretch (measured 3850 vs 1200 cycles for the same instructions).
    if (__builtin_amdgcn_readfirstlane(wave) >= 4) __builtin_amdgcn_s_setprio(1);
    // one step; P = s & 1 selects the register set.  Returns false after the last step.
    auto step = [&](auto pc) -> bool {
        const bool last_ch = (cs.ch + 1 == nchunk);
        const Cur cn = adv(cs);
        if (last_ch && cn.tile < total) next_ct = (cn.tile % tiles_per_img) % nct;
        // U(s+1) by LDS-DMA.  The compiler does not see these (inline assembly) in its vmcnt bookkeeping: a wait for
        // OLDER loads placed after them also waits for them, so a wave issues them where no such wait follows
        // closely -- after the side work in the side-first waves (measured: 2500 cycles per step otherwise).
        auto dma = [&]() {
            Tile tw = cur;                                      // only .ct is used
            tw.ct = next_ct;
            issue_weights(tw, cn.ch, un);
        };
        // The two waves of a SIMD (w and w + 4) take the halves of the step -- the 64 MFMAs, and the vector / LDS /
        // memory work (independent of this step's MFMAs) -- in opposite order, so one of them is always in its MFMA
        // stretch: barrier-synchronised waves running the same order leave the matrix pipe idle during the ~450
        // other instructions of every step (measured: 50 % MFMA busy).  In both orders the DMA is issued before the
        // step's global loads, which is what the end-of-step wait relies on.
        if constexpr (VB == 1) {
            // one V image: all waves multiply, then (behind a barrier) all waves build the next image
            if (computes) mfma_all(vb, ubf);
            barrier_lds_only();                                 // every wave is done reading V(s)
            dma();
            if (!(WINO_ABL & 8)) { transform_load(rawA); transform_store(vn); }
            if (!(WINO_ABL & 128)) write_raw(pc, rawB);
            loads_for(pc, cl);
        } else if (wave < 4) {
            dma();
            if (computes) mfma_all(vb, ubf);
            if (!(WINO_ABL & 8)) { transform_load(rawA); transform_store(vn); }
            if (!(WINO_ABL & 128)) write_raw(pc, rawB);
            loads_for(pc, cl);                                  // step s+4, into the set that was just written out
        } else {
            if (!(WINO_ABL & 8)) { transform_load(rawA); transform_store(vn); }
            if (!(WINO_ABL & 128)) write_raw(pc, rawB);
            dma();
            loads_for(pc, cl);
            if (computes) mfma_all(vb, ubf);
        }
        // V(s+1), U(s+1), raw(s+2) complete.  The DMA is older than this step's loads only: the loads of the previous
        // step (other register set) are waited for as well -- they have had a whole step.
        barrier_lds_keep_loads<C::NIN>();
        if (last_ch) {
            if (computes && (!(WINO_ABL & 4) || d.N < 0)) epilogue(cur);
            zero_acc();
            if (cn.tile < total) {
                const int n = cn.tile / tiles_per_img;
                int bq = cn.tile - n * tiles_per_img;
                cur.n = n;
                cur.ct = bq % nct;
                bq /= nct;
                cur.ox0 = (bq % ntx) * 32;
                cur.oy0 = (bq / ntx) * TH;
            }
        }
        if (cn.tile >= total) return false;
        cs = cn;
        cl = adv(cl);
        { float* t = rawA; rawA = rawB; rawB = t; }
        { float* t = vb; vb = vn; vn = t; }
        { float* t = ubf; ubf = un; un = t; }
        return true;
    };
    while (true) {
        if (!step(IntC<0>{})) break;
        if (!step(IntC<1>{})) break;
    }
}

template <int TN, int TH, int KC, int VB, bool PRE>
static int launch_wino(const YondConvDesc& d, hipStream_t st) {
    using C = WinoCfg<TN, TH, KC, VB>;
    static bool attr_set = false;
    auto kern = conv_wino_kernel<TN, TH, KC, VB, PRE>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const long long total = (long long)(d.Cout / TN) * ((d.Wo + 31) / 32) * ((d.Ho + TH - 1) / TH) * d.N;
    if (total > 0x7fffffffLL) return YOND_EUNSUPPORTED;
    const int grid = total < 256 ? (int)total : 256;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), C::SMEM_BYTES, st, d);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

// U = G g G^T in float64, rounded once to float32, in the LDS order [ct][chunk of 8 channels][plane][kh][tn][4]
extern "C" int yond_pack_conv_wino_weight_f32(const float* w, int cout, int cin, int tn, float* dst) {
    if (!w || !dst || (tn != 64 && tn != 32) || cout % tn != 0) return YOND_EINVAL;
    const int kc = 8, KH = kc / 4;
    if (cin % kc != 0) return YOND_EINVAL;
    static const double Gm[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
    size_t o = 0;
    for (int ct = 0; ct < cout / tn; ++ct)
        for (int ch = 0; ch < cin / kc; ++ch)
            for (int p = 0; p < 16; ++p)
                for (int kh = 0; kh < KH; ++kh)
                    for (int j = 0; j < tn; ++j)
                        for (int e = 0; e < 4; ++e) {
                            const int co = ct * tn + j, ci = ch * kc + kh * 4 + e;
                            const float* g = w + ((size_t)co * cin + ci) * 9;
                            const int xi = p >> 2, nu = p & 3;
                            double s = 0.0;
                            for (int a = 0; a < 3; ++a)
                                for (int bq = 0; bq < 3; ++bq) s += Gm[xi][a] * (double)g[a * 3 + bq] * Gm[nu][bq];
                            dst[o++] = (float)s;
                        }
    return YOND_OK;
}

// tile width the Winograd kernel would use for a layer: 64 (Cout % 64 == 0), 32 (Cout % 32 == 0), 0; Cin % 8 == 0
extern "C" int yond_conv_wino_supported(int cin, int cout) {
    if (cin <= 0 || cout <= 0) return 0;
    if (cout % 64 == 0 && cin % 8 == 0) return 64;
    if (cout % 32 == 0 && cin % 8 == 0) return 32;
    return 0;
}

// called by yond_conv2d_f32 (conv.hip) for desc.algo == 1
int yond_conv_wino_dispatch(const YondConvDesc& d, hipStream_t st) {
    if (d.ksize != 3 || d.stride != 1 || d.shuffle) return YOND_EUNSUPPORTED;
    if (d.tn != 64 && d.tn != 32) return YOND_EINVAL;
    const int kc = 8;
    if (d.Cout % d.tn != 0 || d.C0 % kc != 0 || d.C1 % kc != 0 || d.C0 + d.C1 <= 0) return YOND_EUNSUPPORTED;
    if (d.Ho != d.H || d.Wo != d.W) return YOND_EINVAL;
    if (d.post_act != 0 && d.post_act != 2) return YOND_EUNSUPPORTED;
    if (d.tn == 64) return d.pre_act ? launch_wino<64, 8, 8, 2, true>(d, st) : launch_wino<64, 8, 8, 2, false>(d, st);
    return d.pre_act ? launch_wino<32, 16, 8, 1, true>(d, st) : launch_wino<32, 16, 8, 1, false>(d, st);
}
