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
// channels (weights are the A operand as in conv.hip).  With two waves per SIMD the transforms, the staging and
// the epilogue of one wave run under the MFMAs of the other.
// Per step, in one instruction stream:
//     planes 0..7  of step s   ||  (SiLU +) ds_write of the raw 10 x 34 input tile of step s+1      -> barrier
//     planes 8..15 of step s   ||  global loads of step s+2, input transform raw -> V[(s+1)&1]      -> barrier
// The transformed weights U of step s+1 arrive by LDS-DMA (packed in LDS order by yond_pack_conv_wino_weight_f32).
//
// LDS images:
//   raw [kh][10 rows][parity][17][4]   kh = 4-channel half of the chunk; even / odd columns apart, so the sixteen
//                                      lanes of a ds_read_b128 (consecutive patches, column stride 2) are contiguous
//   V   [2][16 planes][kh][64 patches][4]      U   [2][16 planes][kh][TN][4]
// MFMA kk (0, 1) of a chunk takes channel 2*kq + kk from lane group kq = lane>>4, i.e. float2 (kq & 1) of the
// 16-byte slot of half kh = kq >> 1 (any pairing works as long as U and V agree).
#include "common.h"

template <int TN>
struct WinoCfg {
    static constexpr int KC = 8;
    static constexpr int TH = 8, TW = 32;
    static constexpr int NP = (TH / 2) * (TW / 2);               // 64 patches
    static constexpr int IH = TH + 2, IW = TW + 2, HALF = IW / 2;
    static constexpr int RAW_FLOATS = 2 * IH * 2 * HALF * 4;     // 2720
    static constexpr int V_FLOATS = 16 * 2 * NP * 4;             // 8192
    static constexpr int U_FLOATS = 16 * 2 * TN * 4;
    static constexpr int NITEM = IH * IW * 2;                    // (pixel, kh) 16-byte items
    static constexpr int NT = 512;                               // threads per workgroup
    static constexpr int NIN = (NITEM + NT - 1) / NT;
    static constexpr int NUT = U_FLOATS / 4 / NT;                // LDS-DMA instructions per thread and weight slice
    static constexpr int EP_FLOATS = 4 * TN;
    static constexpr int SMEM_BYTES = (2 * V_FLOATS + 2 * U_FLOATS + RAW_FLOATS + EP_FLOATS) * 4;
};

__device__ __forceinline__ float wino_silu(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f));
}

// wait for this wave's LDS traffic only (not for global loads still in flight), then the workgroup barrier
__device__ __forceinline__ void barrier_lds_only() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int TN, bool PRE>
__global__ __launch_bounds__(512) void conv_wino_kernel(const YondConvDesc d) {
    using C = WinoCfg<TN>;
    static_assert(TN == 64, "wave map below: 2 patch blocks x 4 channel quarters");
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_v = smem;
    float* s_u = smem + 2 * C::V_FLOATS;
    float* s_raw = s_u + 2 * C::U_FLOATS;
    float* s_ep = s_raw + C::RAW_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lj = lane & 15, kq = lane >> 4;
    const int mb = wave & 1, cq = wave >> 1;

    const int nct = d.Cout / TN;
    const int ntx = (d.Wo + 31) / 32, nty = (d.Ho + 7) / 8;
    const int tiles_per_img = nct * ntx * nty;
    const int total = tiles_per_img * d.N;
    const int G = gridDim.x;
    const int lslot = (G % 8 == 0) ? (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : blockIdx.x;   // XCD-contiguous runs
    const int Cin = d.C0 + d.C1;
    const int nchunk = Cin / 8;
    const int my_kh = tid & 1;

    int raw_lds[C::NIN];
#pragma unroll
    for (int k = 0; k < C::NIN; ++k) {
        const int it = tid + k * C::NT;
        const int pix = it >> 1;
        const int py = pix / C::IW, px = pix % C::IW;
        raw_lds[k] = it < C::NITEM ? ((((my_kh * C::IH + py) * 2 + (px & 1)) * C::HALF + (px >> 1)) * 4) : -1;
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
        T.oy0 = ty * 8;
#pragma unroll
        for (int k = 0; k < C::NIN; ++k) {
            const int it = tid + k * C::NT;
            const int pix = it >> 1;
            const int py = pix / C::IW, px = pix % C::IW;
            const int gy = T.oy0 - 1 + py, gx = T.ox0 - 1 + px;
            T.goff[k] = (it < C::NITEM && gy >= 0 && gy < d.H && gx >= 0 && gx < d.W) ? ((n * d.H + gy) * d.W + gx) : -1;
        }
    };

    f32x4 vin[C::NIN];
    unsigned vin_ok = 0;
    auto issue_loads = [&](const Tile& T, int ch) {
        const int c0 = ch * 8;
        const float* src;
        int Cs, cc;
        if (c0 < d.C0) { src = d.src0; Cs = d.C0; cc = c0; }
        else { src = d.src1; Cs = d.C1; cc = c0 - d.C0; }
        vin_ok = 0;
#pragma unroll
        for (int k = 0; k < C::NIN; ++k) {
            const bool ok = T.goff[k] >= 0;                    // outside the image: read pixel 0, zeroed at the LDS write
            vin[k] = *(const f32x4*)(src + (size_t)(ok ? T.goff[k] : 0) * Cs + cc + my_kh * 4);
            vin_ok |= (ok ? 1u : 0u) << k;
        }
    };
    auto issue_weights = [&](const Tile& T, int ch, float* ubuf) {
        const float* wsrc = d.wpk + ((size_t)T.ct * nchunk + ch) * C::U_FLOATS;
#pragma unroll
        for (int k = 0; k < C::NUT; ++k) {
            const int it = tid + k * C::NT;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + it * 4),
                                             (__attribute__((address_space(3))) void*)(ubuf + (it - lane) * 4), 16, 0, 0);
        }
    };
    auto write_raw = [&](int k) {
        f32x4 v = vin[k];
        if (PRE) { v[0] = wino_silu(v[0]); v[1] = wino_silu(v[1]); v[2] = wino_silu(v[2]); v[3] = wino_silu(v[3]); }
        const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
        if (raw_lds[k] >= 0) *(f32x4*)(s_raw + raw_lds[k]) = ((vin_ok >> k) & 1u) ? v : z;       // conv zero padding
    };
    // input transform of one (patch, kh) task per lane of waves 0 (kh 0) and 1 (kh 1); the other six waves skip it
    const int t_kh = wave & 1;
    const int t_patch = lane;
    const int t_pr = t_patch >> 4, t_pc = t_patch & 15;
    auto transform = [&](float* vbuf) {
        if (wave < 2) {
            f32x4 w[4][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                f32x4 dd[4];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    dd[r] = *(const f32x4*)(s_raw + ((((t_kh * C::IH + 2 * t_pr + r) * 2 + (c & 1)) * C::HALF + t_pc + (c >> 1)) * 4));
                w[0][c] = dd[0] - dd[2];                       // B^T d (rows)
                w[1][c] = dd[1] + dd[2];
                w[2][c] = dd[2] - dd[1];
                w[3][c] = dd[1] - dd[3];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {                      // (B^T d) B (columns)
                float* o = vbuf + (((i * 4) * 2 + t_kh) * C::NP + t_patch) * 4;
                *(f32x4*)(o + 0 * 2 * C::NP * 4) = w[i][0] - w[i][2];
                *(f32x4*)(o + 1 * 2 * C::NP * 4) = w[i][1] + w[i][2];
                *(f32x4*)(o + 2 * 2 * C::NP * 4) = w[i][2] - w[i][1];
                *(f32x4*)(o + 3 * 2 * C::NP * 4) = w[i][1] - w[i][3];
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
    auto mfma_planes = [&](auto p0c, const float* vbuf, const float* ubuf) {
        constexpr int P0 = decltype(p0c)::value;
        const float* ub = ubuf + ((kq >> 1) * TN + cq * 16 + lj) * 4 + (kq & 1) * 2;
        const float* vb = vbuf + ((kq >> 1) * C::NP + mb * 32 + lj) * 4 + (kq & 1) * 2;
#pragma unroll
        for (int p = P0; p < P0 + 8; ++p) {
            const f32x2_t uf = *(const f32x2_t*)(ub + p * 2 * TN * 4);
            const f32x2_t v0 = *(const f32x2_t*)(vb + p * 2 * C::NP * 4);
            const f32x2_t v1 = *(const f32x2_t*)(vb + p * 2 * C::NP * 4 + 16 * 4);
            acc[p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[0], v0[0], acc[p][0], 0, 0, 0);   // D = U . V^T
            acc[p][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[0], v1[0], acc[p][1], 0, 0, 0);
            acc[p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[1], v0[1], acc[p][0], 0, 0, 0);
            acc[p][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[1], v1[1], acc[p][1], 0, 0, 0);
        }
    };

    // ---- output side: lane = patch (row 2 mb + t, column lj), its registers = channels 16 cq + 4 kq + (0..3) ----
    const float slope_eff = d.post_act == 2 ? d.slope : 1.0f;
    auto stage_ep = [&](const Tile& T, int par) {
        if (tid < 2 * TN) {
            const int c = tid < TN ? tid : tid - TN;
            const int cu = T.ct * TN + c;
            const int eoff = (d.ebatch ? T.n * d.Cout : 0) + cu;
            float v;
            if (tid < TN) v = d.escale ? d.escale[eoff] : 1.0f;
            else v = d.eshift ? d.eshift[eoff] : 0.0f;
            s_ep[par * 2 * TN + tid] = v;
        }
    };
    auto epilogue = [&](const Tile& T, int par) {
        const float* ep = s_ep + par * 2 * TN;
        const f32x4 es = *(const f32x4*)(ep + cq * 16 + 4 * kq);
        const f32x4 et = *(const f32x4*)(ep + TN + cq * 16 + 4 * kq);
        const int ox = T.ox0 + 2 * lj;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int oy = T.oy0 + 2 * (2 * mb + t);
            const long long pbase = ((long long)(T.n * d.Ho + oy) * d.Wo + ox) * d.Cout + T.ct * TN + cq * 16 + 4 * kq;
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
                    const bool ok = (oy + a < d.Ho) && (ox + b < d.Wo);
                    const long long off = pbase + ((long long)a * d.Wo + b) * d.Cout;
                    f32x4 rr = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (d.res && ok) rr = *(const f32x4*)(d.res + off);
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = fmaf(y[a][b][e], es[e], et[e]);
                        x = x > 0.0f ? x : x * slope_eff;
                        v[e] = x + rr[e];
                    }
                    if (ok) *(f32x4*)(d.dst + off) = v;
                }
        }
    };

    int tile = lslot;
    if (tile >= total) return;
    Tile cur, ld;
    decode(tile, cur);
    ld = cur;
    int ch = 0, b = 0, par = 0;
    zero_acc();
    // prologue: raw + U of step 0, transform, then the loads of step 1
    issue_loads(cur, 0);
    issue_weights(cur, 0, s_u);
#pragma unroll
    for (int k = 0; k < C::NIN; ++k) write_raw(k);
    __syncthreads();
    transform(s_v);
    {
        const bool last0 = nchunk == 1;
        if (last0 && tile + G < total) decode(tile + G, ld);
        issue_loads(ld, last0 ? 0 : 1);
    }
    __syncthreads();
    while (true) {
        const bool last_ch = (ch + 1 == nchunk);
        const int ntile = last_ch ? tile + G : tile;
        const int nch = last_ch ? 0 : ch + 1;
        const bool has_next = ntile < total;
        // step after next (its raw loads are issued in the second half of this step)
        const bool nlast = (nch + 1 == nchunk);
        const int n2tile = nlast ? ntile + G : ntile;
        const int n2ch = nlast ? 0 : nch + 1;
        float* vb = s_v + b * C::V_FLOATS;
        float* ubf = s_u + b * C::U_FLOATS;
        float* vnext = s_v + (b ^ 1) * C::V_FLOATS;
        float* unext = s_u + (b ^ 1) * C::U_FLOATS;
        // `ld` is the tile of step s+1 here (decoded one step ahead)
        issue_weights(ld, nch, unext);
        if (last_ch) stage_ep(cur, par);
        mfma_planes(IntC<0>{}, vb, ubf);
#pragma unroll
        for (int k = 0; k < C::NIN; ++k) write_raw(k);
        __syncthreads();                                        // raw(s+1) and U(s+1) are in LDS
        // tile of step s+2
        Tile* l2 = &ld;
        Tile ld2;
        if (nlast) {
            if (n2tile < total) { decode(n2tile, ld2); l2 = &ld2; }
        }
        issue_loads(*l2, n2ch);
        mfma_planes(IntC<8>{}, vb, ubf);
        transform(vnext);
        barrier_lds_only();                                     // V(s+1) complete; the loads of s+2 stay in flight
        if (last_ch) {
            epilogue(cur, par);
            par ^= 1;
            zero_acc();
            cur = ld;
        }
        if (nlast && n2tile < total) ld = ld2;
        if (!has_next) break;
        tile = ntile;
        ch = nch;
        b ^= 1;
    }
}

template <int TN, bool PRE>
static int launch_wino(const YondConvDesc& d, hipStream_t st) {
    using C = WinoCfg<TN>;
    static bool attr_set = false;
    auto kern = conv_wino_kernel<TN, PRE>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const long long total = (long long)(d.Cout / TN) * ((d.Wo + 31) / 32) * ((d.Ho + 7) / 8) * d.N;
    if (total > 0x7fffffffLL) return YOND_EUNSUPPORTED;
    const int grid = total < 256 ? (int)total : 256;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), C::SMEM_BYTES, st, d);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

// U = G g G^T in float64, rounded once to float32, in the LDS order [ct][chunk][plane][kh][TN][4]
extern "C" int yond_pack_conv_wino_weight_f32(const float* w, int cout, int cin, int tn, float* dst) {
    if (!w || !dst || tn != 64 || cout % tn != 0 || cin % 8 != 0) return YOND_EINVAL;
    static const double Gm[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
    size_t o = 0;
    for (int ct = 0; ct < cout / tn; ++ct)
        for (int ch = 0; ch < cin / 8; ++ch)
            for (int p = 0; p < 16; ++p)
                for (int kh = 0; kh < 2; ++kh)
                    for (int j = 0; j < tn; ++j)
                        for (int e = 0; e < 4; ++e) {
                            const int co = ct * tn + j, ci = ch * 8 + kh * 4 + e;
                            const float* g = w + ((size_t)co * cin + ci) * 9;
                            const int xi = p >> 2, nu = p & 3;
                            double s = 0.0;
                            for (int a = 0; a < 3; ++a)
                                for (int bq = 0; bq < 3; ++bq) s += Gm[xi][a] * (double)g[a * 3 + bq] * Gm[nu][bq];
                            dst[o++] = (float)s;
                        }
    return YOND_OK;
}

extern "C" int yond_conv_wino_supported(int cin, int cout) { return (cin > 0 && cin % 8 == 0 && cout > 0 && cout % 64 == 0) ? 1 : 0; }

// called by yond_conv2d_f32 (conv.hip) for desc.algo == 1
int yond_conv_wino_dispatch(const YondConvDesc& d, hipStream_t st) {
    if (d.ksize != 3 || d.stride != 1 || d.shuffle) return YOND_EUNSUPPORTED;
    if (!yond_conv_wino_supported(d.C0 + d.C1, d.Cout) || d.C0 % 8 != 0 || d.C1 % 8 != 0) return YOND_EUNSUPPORTED;
    if (d.tn != 64) return YOND_EINVAL;
    if (d.Ho != d.H || d.Wo != d.W) return YOND_EINVAL;
    if (d.post_act != 0 && d.post_act != 2) return YOND_EUNSUPPORTED;
    return d.pre_act ? launch_wino<64, true>(d, st) : launch_wino<64, false>(d, st);
}
