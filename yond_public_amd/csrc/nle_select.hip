// K6: exact order statistics (all requested ranks at once) + np.percentile(method='linear')
//     (YOND_SIDD.py:29, 82: np.percentile(img_lap, quants) over the whole packed frame).
//
// Three-level radix select on the order-preserving 32-bit key of a float: 16 + 8 + 8 bits, one streaming
// sweep of the data per level (16-byte loads, 48 MB per sweep at cfg 2), histograms privatised in LDS:
//   level 1  65,536 bins of key >> 16 (sign + exponent + 7 mantissa bits).  The NLE's data are standard
//            deviations of a [0,1] image, so the 16,384 bins of 0 <= x < 2 live in LDS (64 KB); anything else
//            (negative, >= 2) goes straight to the global table.  Neighbouring pixels of the smooth `lap` map
//            fall into the same bin, so runs of equal bins across the lanes of a wave are merged into one
//            LDS atomic (ballot + run length).
//   level 2  for keys whose upper 16 bits hold one of the ranks: 256 bins of bits 15..8 per slot
//   level 3  same for bits 7..0 below the 24-bit prefixes
// A one-workgroup resolve kernel after each sweep walks the cumulative counts and hands the next level its
// sorted list of prefixes ("slots", <= one per requested rank).  HBM-bound integer work; no MFMA.
#include "nle_common.h"

#define SEL_MAXT 64                  // max number of order statistics per call
#define SEL_L1_BINS 65536
#define SEL_WIN_LO 0x8000u           // key >> 16 of +0.0f
#define SEL_WIN_N 16384              // keys 0x8000 .. 0xBFFF: 0 <= x < 2
#define SEL_THREADS 1024
#define SEL_NONE 255
#define SEL_UNROLL 4                 // vectors in flight per thread in the sweeps

struct SelState {
    long long tgt_rank[SEL_MAXT];        // remaining rank inside the target's prefix group
    unsigned int tgt_prefix[SEL_MAXT];   // key bits resolved so far (right-aligned)
    int tgt_slot[SEL_MAXT];              // slot of the target in the level being counted
    unsigned int slot_prefix[SEL_MAXT];  // sorted distinct prefixes of the current level
    unsigned int l1_prefix[SEL_MAXT];    // level-1 slots (sorted 16-bit prefixes), kept for level 3
    int l1_first[SEL_MAXT];              // level-2 slots below each level-1 slot: [first, first + count)
    int l1_count[SEL_MAXT];
    int nslots, nl1, nt, pad;
    unsigned int hist2[SEL_MAXT * 256];
    unsigned int hist3[SEL_MAXT * 256];
    unsigned int hist1[SEL_L1_BINS];
};

struct SelRanks { long long r[SEL_MAXT]; };

// The sweeps read data[head .. head + 4*nvec) as 16-byte vectors; the (at most 6) elements before and after
// that aligned body are "edge" elements handled by the first threads of workgroup 0.
struct SelSpan { size_t head, nvec, n; };

__device__ __forceinline__ bool edge_index(const SelSpan& sp, int t, size_t* idx) {
    const size_t tail0 = sp.head + 4 * sp.nvec;
    if ((size_t)t < sp.head) { *idx = (size_t)t; return true; }
    const size_t j = tail0 + ((size_t)t - sp.head);
    if (j < sp.n) { *idx = j; return true; }
    return false;
}

__global__ __launch_bounds__(SEL_THREADS) void sel_hist1_kernel(const float* __restrict__ data, SelSpan sp, SelState* st) {
    extern __shared__ unsigned int s_h[];                    // [SEL_WIN_N]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < SEL_WIN_N; i += SEL_THREADS) s_h[i] = 0;
    __syncthreads();
    const f32x4* vec = (const f32x4*)(data + sp.head);
    const size_t nchunk = (sp.nvec + 63) / 64;               // a wave takes 64 vectors at a time (uniform trip count)
    const size_t wstride = (size_t)gridDim.x * (SEL_THREADS / 64);
    auto count4 = [&](const f32x4 x, bool valid) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // invalid lanes carry ids that differ from every neighbour and never count
            const unsigned int id = valid ? (f2key(x[e]) >> 16) : (0xFFFF0000u | (unsigned int)lane);
            const unsigned int prev = __shfl_up(id, 1);
            const bool run_head = (lane == 0) || (id != prev);
            const unsigned long long m = __ballot(run_head);
            if (run_head && valid) {
                const unsigned long long rest = (m >> lane) >> 1;
                const unsigned int len = rest ? (unsigned int)__ffsll((long long)rest) : (unsigned int)(64 - lane);
                const unsigned int w = id - SEL_WIN_LO;
                if (w < SEL_WIN_N) atomicAdd(&s_h[w], len);
                else atomicAdd(&st->hist1[id], len);
            }
        }
    };
    // SEL_UNROLL chunks per trip: their loads are all in flight before the first one is counted
    for (size_t c = (size_t)blockIdx.x * (SEL_THREADS / 64) + wave; c < nchunk; c += wstride * SEL_UNROLL) {
        f32x4 x[SEL_UNROLL];
        bool valid[SEL_UNROLL];
#pragma unroll
        for (int u = 0; u < SEL_UNROLL; ++u) {
            const size_t v = (c + u * wstride) * 64 + lane;
            valid[u] = (c + u * wstride) < nchunk && v < sp.nvec;
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            x[u] = valid[u] ? vec[v] : zero;
        }
#pragma unroll
        for (int u = 0; u < SEL_UNROLL; ++u) {
            if ((c + u * wstride) < nchunk) count4(x[u], valid[u]);      // wave-uniform condition
        }
    }
    size_t ei;
    if (blockIdx.x == 0 && tid < 8 && edge_index(sp, tid, &ei)) atomicAdd(&st->hist1[f2key(data[ei]) >> 16], 1u);
    __syncthreads();
    for (int i = tid; i < SEL_WIN_N; i += SEL_THREADS) {
        const unsigned int c = s_h[i];
        if (c) atomicAdd(&st->hist1[SEL_WIN_LO + i], c);
    }
}

// sorted distinct values of s_newp[0..nt) -> slots (s_slotp sorted, s_tslot per target), all in LDS;
// returns the number of slots (all threads), nt <= SEL_MAXT <= blockDim.x
__device__ __forceinline__ int assign_slots(const unsigned int* s_newp, int nt, unsigned int* s_slotp, int* s_tslot) {
    __shared__ int s_ns;
    __shared__ int s_isfirst[SEL_MAXT];
    const int t = threadIdx.x;
    if (t == 0) s_ns = 0;
    if (t < nt) {
        bool first = true;                                     // first occurrence of its prefix
        for (int u = 0; u < t; ++u) first = first && (s_newp[u] != s_newp[t]);
        s_isfirst[t] = first ? 1 : 0;
    }
    __syncthreads();
    if (t < nt) {
        const unsigned int p = s_newp[t];
        int pos = 0;                                           // number of distinct prefixes below p
        for (int u = 0; u < nt; ++u) pos += (s_isfirst[u] && s_newp[u] < p) ? 1 : 0;
        s_tslot[t] = pos;
        if (s_isfirst[t]) { s_slotp[pos] = p; atomicAdd(&s_ns, 1); }
    }
    __syncthreads();
    return s_ns;
}

__global__ __launch_bounds__(SEL_THREADS) void sel_resolve1_kernel(SelState* st, SelRanks ranks, int nt) {
    __shared__ unsigned long long s_wtot[SEL_THREADS / 64];
    __shared__ unsigned int s_newp[SEL_MAXT], s_slotp[SEL_MAXT];
    __shared__ int s_tslot[SEL_MAXT], s_owner[SEL_MAXT];
    __shared__ long long s_rank[SEL_MAXT];
    __shared__ unsigned long long s_before[SEL_MAXT];
    __shared__ __attribute__((aligned(16))) unsigned int s_bins[SEL_MAXT][64];
    constexpr int NW = SEL_THREADS / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < nt) s_rank[tid] = ranks.r[tid];
    // thread tid owns bins 64*tid .. 64*tid+63 and keeps them in registers
    const unsigned int* h = st->hist1 + tid * 64;
    uint4 q[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) q[i] = *(const uint4*)(h + 4 * i);
    unsigned long long mine = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) mine += (unsigned long long)q[i].x + q[i].y + q[i].z + q[i].w;
    unsigned long long incl = wave_incl_scan_u64(mine, lane);
    if (lane == 63) s_wtot[wave] = incl;
    __syncthreads();
    for (int w = 0; w < wave; ++w) incl += s_wtot[w];
    const unsigned long long excl = incl - mine;
    // the owner of the chunk that holds a rank publishes its 64 bins
    for (int t = 0; t < nt; ++t) {
        const unsigned long long rank = (unsigned long long)s_rank[t];
        if (rank >= excl && rank < incl) {
            s_owner[t] = tid;
            s_before[t] = excl;
#pragma unroll
            for (int i = 0; i < 16; ++i) *(uint4*)(&s_bins[t][4 * i]) = q[i];
        }
    }
    __syncthreads();
    // one wave per target: the chunk's 64 bins across the lanes
    for (int t = wave; t < nt; t += NW) {
        const unsigned long long rank = (unsigned long long)s_rank[t];
        const unsigned long long c = s_bins[t][lane];
        const unsigned long long bi = wave_incl_scan_u64(c, lane) + s_before[t];
        const unsigned long long m = __ballot(bi > rank);
        const int d = m ? (__ffsll((long long)m) - 1) : 63;
        const unsigned long long cum = __shfl(bi - c, d);
        if (lane == 0) {
            const unsigned int np = (unsigned int)(s_owner[t] * 64 + d);
            st->tgt_rank[t] = (long long)(rank - cum);
            st->tgt_prefix[t] = np;
            s_newp[t] = np;
        }
    }
    __syncthreads();
    const int ns = assign_slots(s_newp, nt, s_slotp, s_tslot);
    if (tid == 0) { st->nslots = ns; st->nl1 = ns; st->nt = nt; }
    if (tid < nt) st->tgt_slot[tid] = s_tslot[tid];
    if (tid < ns) { st->slot_prefix[tid] = s_slotp[tid]; st->l1_prefix[tid] = s_slotp[tid]; }
}

// LEVEL 2: count bits 15..8 under the level-1 slots; LEVEL 3: bits 7..0 under the level-2 slots
template <int LEVEL>
__global__ __launch_bounds__(SEL_THREADS) void sel_hist23_kernel(const float* __restrict__ data, SelSpan sp, SelState* st) {
    extern __shared__ unsigned int smem[];
    unsigned int* s_h = smem;                                             // [SEL_MAXT][256]
    unsigned char* s_tab = (unsigned char*)(smem + SEL_MAXT * 256);       // [SEL_WIN_N]: level-1 slot of a windowed key
    __shared__ unsigned int s_l1p[SEL_MAXT], s_p2[SEL_MAXT];
    __shared__ int s_first[SEL_MAXT], s_cnt[SEL_MAXT];
    const int tid = threadIdx.x;
    const int nl1 = st->nl1;
    const int ns = st->nslots;
    for (int i = tid; i < ns * 256; i += SEL_THREADS) s_h[i] = 0;
    for (int i = tid; i < SEL_WIN_N / 4; i += SEL_THREADS) ((unsigned int*)s_tab)[i] = 0xFFFFFFFFu;
    if (tid < nl1) { s_l1p[tid] = st->l1_prefix[tid]; s_first[tid] = st->l1_first[tid]; s_cnt[tid] = st->l1_count[tid]; }
    if (LEVEL == 3 && tid < ns) s_p2[tid] = st->slot_prefix[tid];
    __syncthreads();
    if (tid < nl1) {
        const unsigned int w = s_l1p[tid] - SEL_WIN_LO;
        if (w < SEL_WIN_N) s_tab[w] = (unsigned char)tid;
    }
    __syncthreads();
    auto count = [&](float x) {
        const unsigned int key = f2key(x);
        const unsigned int k16 = key >> 16;
        const unsigned int w = k16 - SEL_WIN_LO;
        int s1;
        if (w < SEL_WIN_N) {
            s1 = s_tab[w];
        } else {                                               // outside the LDS window: search the sorted slots
            int lo = 0, hi = nl1;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_l1p[mid] < k16) lo = mid + 1; else hi = mid; }
            s1 = (lo < nl1 && s_l1p[lo] == k16) ? lo : SEL_NONE;
        }
        if (s1 == SEL_NONE) return;
        if (LEVEL == 2) {
            atomicAdd(&s_h[s1 * 256 + ((key >> 8) & 255u)], 1u);
        } else {
            const unsigned int k24 = key >> 8;
            const int f = s_first[s1], l = f + s_cnt[s1];
            for (int i = f; i < l; ++i)
                if (s_p2[i] == k24) atomicAdd(&s_h[i * 256 + (key & 255u)], 1u);
        }
    };
    const f32x4* vec = (const f32x4*)(data + sp.head);
    const size_t vstride = (size_t)gridDim.x * SEL_THREADS;
    for (size_t v = (size_t)blockIdx.x * SEL_THREADS + tid; v < sp.nvec; v += vstride * SEL_UNROLL) {
        f32x4 x[SEL_UNROLL];
#pragma unroll
        for (int u = 0; u < SEL_UNROLL; ++u) {
            const size_t vu = v + u * vstride;
            x[u] = vec[vu < sp.nvec ? vu : v];
        }
#pragma unroll
        for (int u = 0; u < SEL_UNROLL; ++u) {
            if (v + u * vstride < sp.nvec) {
#pragma unroll
                for (int e = 0; e < 4; ++e) count(x[u][e]);
            }
        }
    }
    size_t ei;
    if (blockIdx.x == 0 && tid < 8 && edge_index(sp, tid, &ei)) count(data[ei]);
    __syncthreads();
    unsigned int* gh = LEVEL == 2 ? st->hist2 : st->hist3;
    for (int i = tid; i < ns * 256; i += SEL_THREADS) {
        const unsigned int c = s_h[i];
        if (c) atomicAdd(&gh[i], c);
    }
}

struct LerpArgs { double t[SEL_MAXT / 2]; int nq; };

// LEVEL 3 also writes the order statistics (out_vals, float32) and, when la.nq > 0, np.percentile's linear
// interpolation between each pair of them (out_q, float64).
template <int LEVEL>
__global__ __launch_bounds__(SEL_THREADS) void sel_resolve23_kernel(SelState* st, float* out_vals, LerpArgs la, double* out_q) {
    __shared__ unsigned int s_newp[SEL_MAXT], s_slotp[SEL_MAXT], s_pref[SEL_MAXT], s_l1p[SEL_MAXT];
    __shared__ int s_tslot[SEL_MAXT], s_slot[SEL_MAXT];
    __shared__ long long s_rank[SEL_MAXT];
    constexpr int NW = SEL_THREADS / 64, PER = SEL_MAXT / NW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = st->nt;
    const int nl1 = st->nl1;
    if (tid < nt) { s_rank[tid] = st->tgt_rank[tid]; s_slot[tid] = st->tgt_slot[tid]; s_pref[tid] = st->tgt_prefix[tid]; }
    if (tid < nl1) s_l1p[tid] = st->l1_prefix[tid];
    __syncthreads();
    const unsigned int* gh = LEVEL == 2 ? st->hist2 : st->hist3;
    // one wave per target (targets wave, wave + NW, ...): lane l holds bins 4l..4l+3; all loads first
    uint4 hv[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int t = wave + j * NW;
        const uint4 z = {0u, 0u, 0u, 0u};
        hv[j] = t < nt ? *(const uint4*)(gh + s_slot[t] * 256 + lane * 4) : z;
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int t = wave + j * NW;
        if (t >= nt) break;
        const long long c0 = hv[j].x, c1 = hv[j].y, c2 = hv[j].z, c3 = hv[j].w;
        const long long mine = c0 + c1 + c2 + c3;
        const long long incl = (long long)wave_incl_scan_u64((unsigned long long)mine, lane);
        const long long excl = incl - mine;
        const long long rank = s_rank[t];
        const bool here = rank >= excl && rank < incl;          // exactly one lane unless the data ran out
        const unsigned long long m = __ballot(here);
        const int src = m ? (__ffsll((long long)m) - 1) : 63;
        int d;
        long long cum;
        if (rank < excl + c0) { d = 0; cum = excl; }
        else if (rank < excl + c0 + c1) { d = 1; cum = excl + c0; }
        else if (rank < excl + c0 + c1 + c2) { d = 2; cum = excl + c0 + c1; }
        else { d = 3; cum = excl + c0 + c1 + c2; }
        d = __shfl(d, src);
        cum = __shfl(cum, src);
        if (lane == 0) {
            const unsigned int np = (s_pref[t] << 8) | (unsigned int)(src * 4 + d);
            s_newp[t] = np;
            if (LEVEL == 3) {
                out_vals[t] = key2f(np);
            } else {
                st->tgt_rank[t] = rank - cum;
                st->tgt_prefix[t] = np;
            }
        }
    }
    __syncthreads();
    if (LEVEL == 2) {
        const int ns = assign_slots(s_newp, nt, s_slotp, s_tslot);
        if (tid == 0) st->nslots = ns;
        if (tid < nt) st->tgt_slot[tid] = s_tslot[tid];
        if (tid < ns) st->slot_prefix[tid] = s_slotp[tid];
        if (tid < nl1) {                                       // level-2 slots are sorted, so each level-1 slot owns a range
            const unsigned int p16 = s_l1p[tid];
            int first = 0, cnt = 0;
            for (int i = 0; i < ns; ++i) {
                if ((s_slotp[i] >> 8) == p16) { if (!cnt) first = i; ++cnt; }
            }
            st->l1_first[tid] = first;
            st->l1_count[tid] = cnt;
        }
    } else if (tid < la.nq) {
        // numpy _lerp: diff = b - a in the data dtype; a + diff*t (t < 0.5) or b - diff*(1-t) (t >= 0.5); b == a -> a
        const float av = key2f(s_newp[2 * tid]), bv = key2f(s_newp[2 * tid + 1]);
        const float diff = __fsub_rn(bv, av);
        const double t = la.t[tid];
        double r = (t >= 0.5) ? __dsub_rn((double)bv, __dmul_rn((double)diff, 1.0 - t)) : __dadd_rn((double)av, __dmul_rn((double)diff, t));
        if (bv == av) r = (double)av;
        out_q[tid] = r;
    }
}

extern "C" size_t yond_select_ws_bytes(int nr) {
    (void)nr;
    return sizeof(SelState) + SEL_MAXT * sizeof(float) + 64;
}

static int select_ranks(const float* data, size_t n, const long long* ranks, int nr, float* out, void* ws, hipStream_t st,
                        const LerpArgs* lerp, double* out_q) {
    if (n > 0xFFFFFFFFull) return YOND_EUNSUPPORTED;          // 32-bit bin counters
    if (((uintptr_t)data & 3) || ((uintptr_t)ws & 15)) return YOND_EINVAL;
    SelState* state = (SelState*)ws;
    SelRanks rk;
    for (int i = 0; i < nr; ++i) {
        if (ranks[i] < 0 || (size_t)ranks[i] >= n) return YOND_EINVAL;
        rk.r[i] = ranks[i];
    }
    LerpArgs la;
    la.nq = 0;
    if (lerp) la = *lerp;
    SelSpan sp;
    sp.n = n;
    sp.head = ((16 - ((uintptr_t)data & 15)) & 15) / 4;
    if (sp.head > n) sp.head = n;
    sp.nvec = (n - sp.head) / 4;
    static bool attr = false;
    const int lds23 = SEL_MAXT * 256 * 4 + SEL_WIN_N;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)sel_hist1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SEL_WIN_N * 4);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute((const void*)sel_hist23_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds23);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute((const void*)sel_hist23_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds23);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    hipError_t me = hipMemsetAsync(state->hist2, 0, sizeof(unsigned int) * (2 * SEL_MAXT * 256 + SEL_L1_BINS), st);
    if (me != hipSuccess) return (int)me;
    size_t nb = (sp.nvec + SEL_THREADS * SEL_UNROLL - 1) / (SEL_THREADS * SEL_UNROLL);
    if (nb < 1) nb = 1;
    const size_t nb1 = nb > 512 ? 512 : nb, nb23 = nb > 256 ? 256 : nb;
    hipLaunchKernelGGL(sel_hist1_kernel, dim3((unsigned)nb1), dim3(SEL_THREADS), SEL_WIN_N * 4, st, data, sp, state);
    YOND_LAUNCH_CHECK();
    hipLaunchKernelGGL(sel_resolve1_kernel, dim3(1), dim3(SEL_THREADS), 0, st, state, rk, nr);
    YOND_LAUNCH_CHECK();
    hipLaunchKernelGGL(sel_hist23_kernel<2>, dim3((unsigned)nb23), dim3(SEL_THREADS), lds23, st, data, sp, state);
    YOND_LAUNCH_CHECK();
    hipLaunchKernelGGL(sel_resolve23_kernel<2>, dim3(1), dim3(SEL_THREADS), 0, st, state, out, la, out_q);
    YOND_LAUNCH_CHECK();
    hipLaunchKernelGGL(sel_hist23_kernel<3>, dim3((unsigned)nb23), dim3(SEL_THREADS), lds23, st, data, sp, state);
    YOND_LAUNCH_CHECK();
    hipLaunchKernelGGL(sel_resolve23_kernel<3>, dim3(1), dim3(SEL_THREADS), 0, st, state, out, la, out_q);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

extern "C" int yond_select_ranks_f32(const float* data, size_t n, const int64_t* ranks_host, int nr, float* out, void* ws,
                                     void* stream) {
    if (!data || !ranks_host || !out || !ws || n == 0 || nr <= 0 || nr > SEL_MAXT) return YOND_EINVAL;
    long long r[SEL_MAXT];
    for (int i = 0; i < nr; ++i) r[i] = (long long)ranks_host[i];
    return select_ranks(data, n, r, nr, out, ws, (hipStream_t)stream, nullptr, nullptr);
}

extern "C" int yond_percentiles_f32(const float* data, size_t n, const double* q_host, int nq, double* out, void* ws,
                                    void* stream) {
    if (!data || !q_host || !out || !ws || n == 0 || nq <= 0 || nq > SEL_MAXT / 2) return YOND_EINVAL;
    long long r[SEL_MAXT];
    LerpArgs la;
    for (int i = 0; i < nq; ++i) {
        if (!(q_host[i] >= 0.0 && q_host[i] <= 100.0)) return YOND_EINVAL;
        // numpy: virtual index = q/100 * (n-1); previous = floor, next = previous+1 clipped, gamma = frac
        const double vidx = (q_host[i] / 100.0) * (double)(n - 1);
        long long lo = (long long)floor(vidx);
        if (lo > (long long)n - 1) lo = (long long)n - 1;
        long long hi = lo + 1;
        if (hi > (long long)n - 1) hi = (long long)n - 1;
        r[2 * i] = lo;
        r[2 * i + 1] = hi;
        la.t[i] = vidx - (double)lo;
    }
    la.nq = nq;
    float* vals = (float*)((unsigned char*)ws + sizeof(SelState));
    return select_ranks(data, n, r, 2 * nq, vals, ws, (hipStream_t)stream, &la, out);
}
