// K6'/K7': the estimator's threshold selection (YOND_SIDD.py:22-49) in TWO sweeps of the lap map.
//
//   np.percentile needs 2 x 20 exact order statistics of 12 M values; score-3 needs, per threshold, the number of
//   occupied 1/1000 mean bins among lap <= ths[i].  nle_select.hip + the occupancy sweep of nle.hip did that in four
//   sweeps (3 radix levels + occupancy, 240 MB) with one-workgroup resolve kernels in between.  Here:
//
//   sweep 1 (nf_stats_kernel, or the producer of the lap map itself: the fused box kernel of nle.hip)
//       level-1 histogram of key >> 16 (16,384-bin LDS window for 0 <= x < 2) with run-length merging down the
//       columns, and per MEAN BIN the smallest lap seen (atomic max of ~key): a bin is occupied among lap <= T exactly
//       when its smallest lap is <= T, so the thresholds need not be known during the sweep -- the occupancy sweep
//       disappears.  The workgroup that arrives last locates the level-1 bin of every rank ("slot"), and lays out one
//       candidate range per slot (its bin count is exact).
//   sweep 2 (nf_collect_kernel)  the values that fall into a slot (about 0.4 % of the data per slot) are compacted
//       into the slot's range as their low 16 key bits (2 bytes each).
//   finish  (nf_final_kernel)    one workgroup per slot: 8 + 8 bit radix select inside the slot's candidates; the
//       last one to finish evaluates np.percentile's linear interpolation, npeaks from the per-bin minima, the
//       score and its first minimum (YOND_SIDD.py:37-47).
// Integer / streaming work bound by HBM; no MFMA.
#include "nle_common.h"

#define NF_NONE 255
#define NF_SEG 16            // rows per lane segment in the stand-alone stats sweep
#define NF_BATCH 4

// ------------------------------------------------------------------------------------------------------------
// sweep 1 as a kernel of its own (used when the lap map comes from elsewhere: collab mode, the function seam)
// ------------------------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(512) void nf_stats_kernel(const float* __restrict__ lap, const float* __restrict__ mean, int rows,
                                                       int width, NleState* st, NfArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned int s_dyn[];                // [NF_WIN_N] histogram window, [NF_BINS] per-bin ~min key
    unsigned int* s_h = s_dyn;
    unsigned int* s_mi = s_dyn + NF_WIN_N;
    const int tid = threadIdx.x;
    for (int i = tid; i < (NF_WIN_N + NF_BINS) / 4; i += 512) ((uint4*)s_dyn)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    unsigned int bmask = 0;                                // blocks of 4096 level-1 bins this lane added to in global memory
    auto flush = [&](unsigned int id, unsigned int cnt) {
        if (!cnt) return;
        const unsigned int w = id - NF_WIN_LO;
        if (w < NF_WIN_N) atomicAdd(&s_h[w], cnt);
        else { atomicAdd(&st->hist1[nf_hpos(id)], cnt); bmask |= 1u << (id >> 12); }
    };
    const int G = (width + VEC - 1) / VEC;
    const int nseg = (rows + NF_SEG - 1) / NF_SEG;
    const size_t nitems = (size_t)G * nseg;
    for (size_t item = (size_t)blockIdx.x * 512 + tid; item < nitems; item += (size_t)gridDim.x * 512) {
        const int sgm = (int)(item / G), g = (int)(item % G);
        const int r0 = sgm * NF_SEG, r1 = min(rows, r0 + NF_SEG);
        const int c0 = g * VEC;
        unsigned int rid[VEC], rcnt[VEC], cmax[VEC];
        int cbin[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) { rid[e] = 0; rcnt[e] = 0; cbin[e] = -1; cmax[e] = 0; }
        for (int rb = r0; rb < r1; rb += NF_BATCH) {
            f32x4 Lb[NF_BATCH], Mb[NF_BATCH];
#pragma unroll
            for (int j = 0; j < NF_BATCH; ++j) {
                const int r = min(rb + j, r1 - 1);
                const size_t base = (size_t)r * width + c0;
                if (VEC == 4) { Lb[j] = *(const f32x4*)(lap + base); Mb[j] = *(const f32x4*)(mean + base); }
                else { Lb[j][0] = lap[base]; Mb[j][0] = mean[base]; }
            }
#pragma unroll
            for (int j = 0; j < NF_BATCH; ++j) {
                if (rb + j >= r1) break;
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const unsigned int key = f2key(Lb[j][e]);
                    const unsigned int id = key >> 16;
                    if (id != rid[e]) { flush(rid[e], rcnt[e]); rid[e] = id; rcnt[e] = 0; }
                    rcnt[e] += 1;
                    const int bin = (int)__fmul_rn(fminf(fmaxf(Mb[j][e], 0.0f), 1.0f), 1000.0f);   // (mean.clip(0,1)*nbins).astype(int)
                    if (bin != cbin[e]) { cbin[e] = bin; cmax[e] = 0; }
                    const unsigned int inv = ~key;
                    if (inv > cmax[e]) { cmax[e] = inv; atomicMax(&s_mi[bin], inv); }
                }
            }
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) flush(rid[e], rcnt[e]);
    }
    __syncthreads();
    for (int i = tid; i < NF_WIN_N / 4; i += 512) {        // mostly empty: four bins per look
        const uint4 c4 = ((const uint4*)s_h)[i];
        if (c4.x | c4.y | c4.z | c4.w) {
            const unsigned int cc[4] = {c4.x, c4.y, c4.z, c4.w};
            bmask |= 1u << ((NF_WIN_LO + 4 * i) >> 12);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (cc[j]) atomicAdd(&st->hist1[nf_hpos(NF_WIN_LO + 4 * i + j)], cc[j]);
            }
        }
    }
    for (int i = tid; i < NF_BINS; i += 512) {
        const unsigned int v = s_mi[i];
        if (v) atomicMax(&st->maxinv[i], v);
    }
    nf_mark_blocks(st, bmask);
    if (nf_arrive_last(&st->ticket[0], gridDim.x)) nf_resolve1(st, a, s_dyn);
}

// ------------------------------------------------------------------------------------------------------------
// sweep 2: compact the values that fall into a slot (low 16 key bits) into the slot's candidate range
// ------------------------------------------------------------------------------------------------------------
#define NFC_THREADS 1024
#define NFC_UNROLL 4

__global__ __launch_bounds__(NFC_THREADS) void nf_collect_kernel(const float* __restrict__ data, size_t head, size_t nvec, size_t n,
                                                                 NleState* st, unsigned short* __restrict__ cand) {
    __shared__ unsigned char s_tab[NF_WIN_N];              // slot of a windowed level-1 bin (or NF_NONE)
    __shared__ unsigned int s_pref[NF_MAXT], s_off[NF_MAXT], s_cnt[NF_MAXT], s_base[NF_MAXT];
    const int tid = threadIdx.x;
    const int ns = st->nslots;
    for (int i = tid; i < NF_WIN_N / 4; i += NFC_THREADS) ((unsigned int*)s_tab)[i] = 0xFFFFFFFFu;
    if (tid < NF_MAXT) { s_pref[tid] = tid < ns ? st->slot_prefix[tid] : 0xFFFFFFFFu; s_off[tid] = tid < ns ? st->slot_off[tid] : 0; s_cnt[tid] = 0; }
    __syncthreads();
    if (tid < ns) {
        const unsigned int w = s_pref[tid] - NF_WIN_LO;
        if (w < NF_WIN_N) s_tab[w] = (unsigned char)tid;
    }
    __syncthreads();
    auto slot_of = [&](unsigned int key) -> int {
        const unsigned int k16 = key >> 16;
        const unsigned int w = k16 - NF_WIN_LO;
        if (w < NF_WIN_N) return s_tab[w];
        int lo = 0, hi = ns;                                // outside the LDS window: search the sorted slots
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_pref[mid] < k16) lo = mid + 1; else hi = mid; }
        return (lo < ns && s_pref[lo] == k16) ? lo : NF_NONE;
    };
    const f32x4* vec = (const f32x4*)(data + head);
    const size_t per_iter = (size_t)NFC_THREADS * NFC_UNROLL;
    const size_t niter = (nvec + per_iter - 1) / per_iter;
    for (size_t it = blockIdx.x; it < niter; it += gridDim.x) {
        unsigned int tag[NFC_UNROLL * 4];                  // (slot << 24) | (local position << 16 >> 16 ...) see below
        unsigned int low[NFC_UNROLL * 2];                  // two 16-bit low keys per word
        f32x4 x[NFC_UNROLL];
        bool valid[NFC_UNROLL];
#pragma unroll
        for (int u = 0; u < NFC_UNROLL; ++u) {
            const size_t v = it * per_iter + (size_t)u * NFC_THREADS + tid;
            valid[u] = v < nvec;
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            x[u] = valid[u] ? vec[v] : zero;
        }
#pragma unroll
        for (int u = 0; u < NFC_UNROLL; ++u) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned int key = f2key(x[u][e]);
                const int s = valid[u] ? slot_of(key) : NF_NONE;
                unsigned int t = 0xFFFFFFFFu;
                if (s != NF_NONE) t = ((unsigned int)s << 24) | atomicAdd(&s_cnt[s], 1u);     // local position < 2^14
                tag[u * 4 + e] = t;
                if (e & 1) low[u * 2 + (e >> 1)] |= (key & 0xFFFFu) << 16;
                else low[u * 2 + (e >> 1)] = key & 0xFFFFu;
            }
        }
        __syncthreads();
        if (tid < ns && s_cnt[tid]) s_base[tid] = atomicAdd(&st->slot_fill[tid], s_cnt[tid]);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NFC_UNROLL * 4; ++i) {
            const unsigned int t = tag[i];
            if (t != 0xFFFFFFFFu) {
                const int s = (int)(t >> 24);
                cand[(size_t)s_off[s] + s_base[s] + (t & 0xFFFFFFu)] = (unsigned short)((low[i >> 1] >> ((i & 1) * 16)) & 0xFFFFu);
            }
        }
        __syncthreads();
        if (tid < NF_MAXT) s_cnt[tid] = 0;
        __syncthreads();
    }
    // the (at most 6) elements before / after the 16-byte aligned body
    if (blockIdx.x == 0 && tid < 8) {
        const size_t tail0 = head + 4 * nvec;
        size_t idx = n;
        if ((size_t)tid < head) idx = (size_t)tid;
        else if (tail0 + ((size_t)tid - head) < n) idx = tail0 + ((size_t)tid - head);
        if (idx < n) {
            const unsigned int key = f2key(data[idx]);
            const int s = slot_of(key);
            if (s != NF_NONE) cand[(size_t)s_off[s] + atomicAdd(&st->slot_fill[s], 1u)] = (unsigned short)(key & 0xFFFFu);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// finish: one workgroup per slot selects inside its candidates; the last one evaluates percentiles and score 3
// ------------------------------------------------------------------------------------------------------------
#define NFF_THREADS 1024

__global__ __launch_bounds__(NFF_THREADS) void nf_final_kernel(NleState* st, const unsigned short* __restrict__ cand, NfArgs a,
                                                               int want_score) {
    __shared__ unsigned int s_h[256];
    __shared__ int s_tl[NF_MAXT];                           // targets of this slot
    __shared__ long long s_rk[NF_MAXT];
    __shared__ int s_b8[NF_MAXT];
    __shared__ int s_nt;
    __shared__ double s_ths[NF_MAXQ];
    __shared__ int s_np[NF_MAXQ];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = blockIdx.x;
    const int ns = st->nslots, nt = st->nt;
    if (s < ns) {
        const unsigned int cnt = st->slot_cnt[s];
        const unsigned short* c = cand + st->slot_off[s];
        const unsigned int prefix = st->slot_prefix[s];
        if (tid == 0) s_nt = 0;
        if (tid < 256) s_h[tid] = 0;
        __syncthreads();
        if (tid < nt && st->tgt_slot[tid] == s) {            // (one load per thread; a serial scan by one thread cost 20 us)
            const int k = atomicAdd(&s_nt, 1);
            s_tl[k] = tid;
            s_rk[k] = st->tgt_rank[tid];
        }
        __syncthreads();
        const int mt = s_nt;
        // candidate ranges start on 16-byte boundaries: 8 candidates per load, four loads in flight
        const uint4* c4 = (const uint4*)c;
        const unsigned int n8 = cnt / 8;
        auto count_hi = [&](uint4 q) {
            const unsigned int wv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) { atomicAdd(&s_h[(wv[j] >> 8) & 255u], 1u); atomicAdd(&s_h[wv[j] >> 24], 1u); }
        };
        for (unsigned int i = tid; i < n8; i += NFF_THREADS * 4) {
            uint4 q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) q[u] = c4[min(i + u * NFF_THREADS, n8 - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i + u * NFF_THREADS < n8) count_hi(q[u]);
        }
        for (unsigned int i = n8 * 8 + tid; i < cnt; i += NFF_THREADS) atomicAdd(&s_h[c[i] >> 8], 1u);
        __syncthreads();
        // wave 0: cumulative counts of the 256 bins (4 per lane), then every target finds its bin
        if (wave == 0) {
            const unsigned long long c0 = s_h[lane * 4], c1 = s_h[lane * 4 + 1], c2 = s_h[lane * 4 + 2], c3 = s_h[lane * 4 + 3];
            const unsigned long long tot = c0 + c1 + c2 + c3;
            const unsigned long long incl = wave_incl_scan_u64(tot, lane);
            const unsigned long long excl = incl - tot;
            for (int k = 0; k < mt; ++k) {
                const unsigned long long rank = (unsigned long long)s_rk[k];
                const unsigned long long m = __ballot(incl > rank);
                const int src = m ? (__ffsll((long long)m) - 1) : 63;
                int d;
                unsigned long long cum;
                if (rank < excl + c0) { d = 0; cum = excl; }
                else if (rank < excl + c0 + c1) { d = 1; cum = excl + c0; }
                else if (rank < excl + c0 + c1 + c2) { d = 2; cum = excl + c0 + c1; }
                else { d = 3; cum = excl + c0 + c1 + c2; }
                d = __shfl(d, src);
                cum = __shfl(cum, src);
                if (lane == 0) { s_b8[k] = src * 4 + d; s_rk[k] = (long long)(rank - cum); }
            }
        }
        __syncthreads();
        // second byte: one pass over the candidates per DISTINCT first byte among the targets (usually one or two)
        for (int k = 0; k < mt; ++k) {
            bool dup = false;
            for (int u = 0; u < k; ++u) dup = dup || (s_b8[u] == s_b8[k]);
            if (dup) continue;                              // uniform: handled together with the first target of that byte
            const unsigned int b8 = (unsigned int)s_b8[k];
            if (tid < 256) s_h[tid] = 0;
            __syncthreads();
            auto count_lo = [&](uint4 q) {
                const unsigned int wv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (((wv[j] >> 8) & 255u) == b8) atomicAdd(&s_h[wv[j] & 255u], 1u);
                    if ((wv[j] >> 24) == b8) atomicAdd(&s_h[(wv[j] >> 16) & 255u], 1u);
                }
            };
            for (unsigned int i = tid; i < n8; i += NFF_THREADS * 4) {
                uint4 q[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) q[u] = c4[min(i + u * NFF_THREADS, n8 - 1)];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (i + u * NFF_THREADS < n8) count_lo(q[u]);
            }
            for (unsigned int i = n8 * 8 + tid; i < cnt; i += NFF_THREADS) {
                const unsigned int v = c[i];
                if ((v >> 8) == b8) atomicAdd(&s_h[v & 255u], 1u);
            }
            __syncthreads();
            if (wave == 0) {
                const unsigned long long c0 = s_h[lane * 4], c1 = s_h[lane * 4 + 1], c2 = s_h[lane * 4 + 2], c3 = s_h[lane * 4 + 3];
                const unsigned long long tot = c0 + c1 + c2 + c3;
                const unsigned long long incl = wave_incl_scan_u64(tot, lane);
                const unsigned long long excl = incl - tot;
                for (int u = k; u < mt; ++u) {
                    if ((unsigned int)s_b8[u] != b8) continue;
                    const unsigned long long rank = (unsigned long long)s_rk[u];
                    const unsigned long long m = __ballot(incl > rank);
                    const int src = m ? (__ffsll((long long)m) - 1) : 63;
                    int d;
                    if (rank < excl + c0) d = 0;
                    else if (rank < excl + c0 + c1) d = 1;
                    else if (rank < excl + c0 + c1 + c2) d = 2;
                    else d = 3;
                    d = __shfl(d, src);
                    if (lane == 0) {
                        const unsigned int key = (prefix << 16) | (b8 << 8) | (unsigned int)(src * 4 + d);
                        __hip_atomic_store(&st->vals[s_tl[u]], key2f(key), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            __syncthreads();
        }
    }
    if (!nf_arrive_last(&st->ticket[1], gridDim.x)) return;
    // ---- last workgroup: np.percentile's lerp, npeaks, score 3 ----
    // (every other workgroup has left: sweep 2's fill counters and arrival ticket go back to zero, so that the call can be
    // repeated on the same workspace -- sweep 1's state is untouched)
    if (tid < NF_MAXT) st->slot_fill[tid] = 0;
    if (tid == 0) st->ticket[1] = 0;
    const int nq = a.nq;
    if (tid < nq) {
        // numpy _lerp: diff = b - a in the data dtype; a + diff*t (t < 0.5) or b - diff*(1-t) (t >= 0.5); b == a -> a
        const float av = __hip_atomic_load(&st->vals[2 * tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float bv = __hip_atomic_load(&st->vals[2 * tid + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float diff = __fsub_rn(bv, av);
        const double t = a.lerp_t[tid];
        double r = (t >= 0.5) ? __dsub_rn((double)bv, __dmul_rn((double)diff, 1.0 - t)) : __dadd_rn((double)av, __dmul_rn((double)diff, t));
        if (bv == av) r = (double)av;
        s_ths[tid] = r;
        st->ths[tid] = r;
        s_np[tid] = 0;
    }
    __syncthreads();
    if (!want_score) return;
    for (int b = tid; b < NF_BINS; b += NFF_THREADS) {
        const unsigned int inv = st->maxinv[b];
        if (inv) {
            const double lmin = (double)key2f(~inv);        // smallest lap of the bin
            int first = nq;                                 // first threshold that admits the bin (YOND_SIDD.py:37: data <= ths[i]);
            for (int i = nq - 1; i >= 0; --i)               // percentiles ascend, so it counts for every later one too
                if (lmin <= s_ths[i]) first = i;
            if (first < nq) atomicAdd(&s_np[first], 1);
        }
    }
    __syncthreads();
    if (tid == 0) {
        for (int i = 1; i < nq; ++i) s_np[i] += s_np[i - 1];
        double best = INFINITY;
        int bi = 0;
        for (int i = 0; i < nq; ++i) {
            st->npeaks[i] = s_np[i];
            const double score = __ddiv_rn(s_ths[i], __dmul_rn(a.quants[i], (double)s_np[i]));   // YOND_SIDD.py:45
            if (i >= 1 && score < best) { best = score; bi = i; }                                 // first minimum of score[1:]
        }
        st->sel[0] = (double)bi;
        st->sel[1] = s_ths[bi];
        st->sel[2] = a.quants[bi];
        st->sel[3] = best;
    }
}

// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------

extern "C" size_t yond_nle_ws_bytes(size_t n) {
    // state + candidate ranges: 2 bytes per element at most (every element in a slot), each range padded to 16 bytes
    return nf_state_bytes() + 2 * n + 16 * NF_MAXT + 256;
}

int nf_make_args(size_t n, const double* q_host, int nq, NfArgs* a) {   // declared in nle_common.h
    if (!q_host || nq <= 0 || nq > NF_MAXQ || n == 0) return YOND_EINVAL;
    for (int i = 0; i < nq; ++i) {
        if (!(q_host[i] >= 0.0 && q_host[i] <= 100.0)) return YOND_EINVAL;
        // numpy: virtual index = q/100 * (n-1); previous = floor, next = previous+1 clipped, gamma = frac
        const double vidx = (q_host[i] / 100.0) * (double)(n - 1);
        long long lo = (long long)floor(vidx);
        if (lo > (long long)n - 1) lo = (long long)n - 1;
        long long hi = lo + 1;
        if (hi > (long long)n - 1) hi = (long long)n - 1;
        a->ranks[2 * i] = lo;
        a->ranks[2 * i + 1] = hi;
        a->lerp_t[i] = vidx - (double)lo;
        a->quants[i] = q_host[i];
    }
    for (int i = nq; i < NF_MAXQ; ++i) { a->ranks[2 * i] = a->ranks[2 * i + 1] = 0; a->lerp_t[i] = 0.0; a->quants[i] = 1.0; }
    a->nt = 2 * nq;
    a->nq = nq;
    return YOND_OK;
}

int nf_launch_stats(const float* lap, const float* mean, size_t n, int width, const double* q_host, int nq, void* ws,
                    hipStream_t st, bool reset) {                       // declared in nle_common.h
    if (!lap || !mean || !ws || n == 0 || n > 0xFFFFFFFFull || ((uintptr_t)ws & 15)) return YOND_EINVAL;
    if (width <= 0 || n % (size_t)width != 0 || n / (size_t)width > 0x7FFFFFFFull) return YOND_EINVAL;
    NfArgs a;
    int rc = nf_make_args(n, q_host, nq, &a);
    if (rc) return rc;
    hipError_t e = hipSuccess;
    if (reset) {
        e = hipMemsetAsync(ws, 0, nf_state_bytes(), st);
        if (e != hipSuccess) return (int)e;
    }
    static bool attr = false;
    const int lds = (NF_WIN_N + NF_BINS) * 4;
    if (!attr) {
        e = hipFuncSetAttribute((const void*)nf_stats_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute((const void*)nf_stats_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    const int rows = (int)(n / (size_t)width);
    const bool vec = (width % 4 == 0) && !(((uintptr_t)lap | (uintptr_t)mean) & 15);
    const int G = vec ? width / 4 : width;
    const size_t nitems = (size_t)G * ((rows + NF_SEG - 1) / NF_SEG);
    size_t nb = (nitems + 511) / 512;
    const size_t cap = (size_t)yond_exp_long("YOND_STATS_WGS", 256);  // one workgroup per CU: fewer flushes onto the same counters (56 vs 61 us at 512)
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    if (vec) hipLaunchKernelGGL(nf_stats_kernel<4>, dim3((unsigned)nb), dim3(512), lds, st, lap, mean, rows, width, (NleState*)ws, a);
    else hipLaunchKernelGGL(nf_stats_kernel<1>, dim3((unsigned)nb), dim3(512), lds, st, lap, mean, rows, width, (NleState*)ws, a);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

extern "C" int yond_nle_stats_f32(const float* lap, const float* mean, size_t n, int width, const double* q_host, int nq,
                                  void* ws, void* stream) {
    return nf_launch_stats(lap, mean, n, width, q_host, nq, ws, (hipStream_t)stream, true);
}

extern "C" int yond_nle_threshold_f32(const float* lap, size_t n, const double* q_host, int nq, int want_score, void* ws,
                                      void* stream) {
    if (!lap || !ws || n == 0 || n > 0xFFFFFFFFull || ((uintptr_t)ws & 15) || ((uintptr_t)lap & 3)) return YOND_EINVAL;
    NfArgs a;
    int rc = nf_make_args(n, q_host, nq, &a);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    unsigned short* cand = (unsigned short*)((unsigned char*)ws + nf_state_bytes());
    size_t head = ((16 - ((uintptr_t)lap & 15)) & 15) / 4;
    if (head > n) head = n;
    const size_t nvec = (n - head) / 4;
    size_t nb = (nvec + (size_t)NFC_THREADS * NFC_UNROLL - 1) / ((size_t)NFC_THREADS * NFC_UNROLL);
    const size_t ccap = (size_t)yond_exp_long("YOND_COLLECT_WGS", 512);  // two workgroups per CU: 128 / 256 / 384 / 512 / 768 -> 111.5 / 102.0 / 99.5 / 97.5 / 99.1 us for the three launches
    if (nb > ccap) nb = ccap;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(nf_collect_kernel, dim3((unsigned)nb), dim3(NFC_THREADS), 0, st, lap, head, nvec, n, (NleState*)ws, cand);
    YOND_LAUNCH_CHECK();
    hipLaunchKernelGGL(nf_final_kernel, dim3(NF_MAXT), dim3(NFF_THREADS), 0, st, (NleState*)ws, cand, a, want_score);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

// layout of the result head for the host binding: offsets in bytes of {ths, sel, mom, npeaks, frame_max_key}
extern "C" int yond_nle_state_layout(int* off /*[5]*/) {
    if (!off) return YOND_EINVAL;
    off[0] = (int)offsetof(NleState, ths);
    off[1] = (int)offsetof(NleState, sel);
    off[2] = (int)offsetof(NleState, mom);
    off[3] = (int)offsetof(NleState, npeaks);
    off[4] = (int)offsetof(NleState, frame_max_key);
    return YOND_OK;
}
