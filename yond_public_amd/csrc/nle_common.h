// Shared pieces of the noise-level estimator's kernels (nle.hip, nle_select.hip, nle_fast.hip).
#pragma once
#include <stddef.h>
#include <math.h>
#include "common.h"

// order-preserving 32-bit key of a float (total order of floats as unsigned) and its inverse
__device__ __forceinline__ unsigned int f2key(float f) {
    const unsigned int b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned int k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

__device__ __forceinline__ unsigned long long wave_incl_scan_u64(unsigned long long v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long up = __shfl_up(v, o);
        if (lane >= o) v += up;
    }
    return v;
}

// ---- state of one threshold estimate (yond_nle_*; nle_fast.hip, written to by the fused box kernel of nle.hip) ----
#define NF_MAXT 64            // order statistics per call (two per quantile)
#define NF_MAXQ 32            // quantiles per call
#define NF_BINS 1024          // mean bins: 1001 used (np.bincount(minlength=nbins+1), YOND_SIDD.py:40)
#define NF_L1 65536           // level-1 histogram: key >> 16
#define NF_WIN_LO 0x8000u     // key >> 16 of +0.0f
#define NF_WIN_N 16384        // keys 0x8000 .. 0xBFFF: 0 <= x < 2 (standard deviations of a [0,1] image) live in LDS

struct NleState {
    // results: the host reads this head (NLE_HEAD_BYTES) in one copy
    double ths[NF_MAXQ];             // np.percentile(lap, quants, 'linear')
    double sel[4];                   // i*, ths[i*], quants[i*], score[i*]      (YOND_SIDD.py:45-47)
    double mom[10];                  // {n, Sm, Sv, Smm, Smv} x {all, 1e-4 < mean < 0.8} over lap < sel[1]
    int npeaks[NF_MAXQ];             // occupied mean bins among lap <= ths[i]   (YOND_SIDD.py:37-43)
    float vals[NF_MAXT];             // the order statistics behind ths
    unsigned int frame_max_key;      // f2key of the frame maximum (0 = not collected)
    int nslots, nt, nq;
    // select state
    unsigned int ticket[4];          // [0] sweep 1, [1] sweep 2 arrivals; [3] bit b: block b of 4096 level-1 bins is not empty
    long long tgt_rank[NF_MAXT];     // remaining rank inside the target's level-1 bin
    unsigned int tgt_prefix[NF_MAXT];
    int tgt_slot[NF_MAXT];
    unsigned int slot_prefix[NF_MAXT];   // sorted distinct level-1 bins that hold a rank
    unsigned int slot_off[NF_MAXT], slot_cnt[NF_MAXT], slot_fill[NF_MAXT];   // candidate ranges of sweep 2
    unsigned int maxinv[NF_BINS];    // per mean bin: ~key of the SMALLEST lap seen (0 = bin empty): the bin is occupied
                                     // among lap <= T  <=>  its smallest lap <= T
    unsigned int hist1[NF_L1];
    unsigned long long dbg[64];      // cycle stamps of diagnostic builds (-DBF_STAMPS)
};
#define NLE_HEAD_DOUBLES (NF_MAXQ + 4 + 10)

struct NfArgs {
    long long ranks[NF_MAXT];
    double lerp_t[NF_MAXQ];
    double quants[NF_MAXQ];
    int nt, nq;
};

// ------------------------------------------------------------------------------------------------------------
// level-1 resolve, run by ONE workgroup (the last to arrive) after every level-1 count has been added to st->hist1
// ------------------------------------------------------------------------------------------------------------
// Where the count of level-1 bin `id` lives inside hist1: neighbouring bins 64 words apart (a 64 x 64 transpose inside every
// aligned block of 4096).  The bins a frame fills are a few hundred NEIGHBOURS, and device-scope atomics of ~1000
// workgroups on the same cache lines serialize: with the counts in id order the flush of the box kernels took ~110 us,
// spread out ~25 us.
__device__ __forceinline__ unsigned int nf_hpos(unsigned int id) {
    return (id & ~0xFFFu) | ((id & 63u) << 6) | ((id >> 6) & 63u);
}

// every writer of hist1 marks the block of 4096 bins it touched (one atomic per wave), so that the resolve reads those only
__device__ __forceinline__ void nf_mark_blocks(NleState* st, unsigned int mask /* per lane: bit (id >> 12) */) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) mask |= __shfl_xor(mask, d);
    if ((threadIdx.x & 63) == 0 && mask) atomicOr(&st->ticket[3], mask);
}

__device__ inline void nf_resolve1(NleState* st, const NfArgs& a, unsigned int* s_scratch /* >= 2048 words of LDS, 16-byte aligned */) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x, nw = nthr >> 6;
    const int nt = a.nt;
    unsigned int* s_g = s_scratch;                                              // [1024] sums of 64 consecutive bins
    unsigned long long* s_wtot = (unsigned long long*)(s_scratch + NF_L1 / 64); // [16]
    unsigned long long* s_before = s_wtot + 16;                                 // [64]
    int* s_owner = (int*)(s_before + NF_MAXT);                                  // [64]
    unsigned int* s_newp = (unsigned int*)(s_owner + NF_MAXT);                  // [64]
    int* s_first = (int*)(s_newp + NF_MAXT);                                    // [64]
    unsigned int* s_slotp = (unsigned int*)(s_first + NF_MAXT);                 // [64]
    int* s_tslot = (int*)(s_slotp + NF_MAXT);                                   // [64]
    unsigned int* s_cnt = (unsigned int*)(s_tslot + NF_MAXT);                   // [64]
    unsigned int* s_pcnt = s_cnt + NF_MAXT;                                     // [64]
    int* s_ns = (int*)(s_pcnt + NF_MAXT);                                       // [1]
    // group sums: in the transposed layout row r of a block holds bin (64 c + r) of every group c, so a wave reads whole
    // rows (16 lanes x 16 bytes, four rows per load) and adds down the columns
    const unsigned int bmask = st->ticket[3];                                   // blocks that hold anything (a frame: one or two)
    int busy = 0;
    for (int b = 0; b < NF_L1 / 4096; ++b) {
        const bool on = (bmask >> b) & 1u;
        const int who = on ? (busy++ % nw) : (b % nw);
        if (who != wave) continue;
        if (!on) { s_g[b * 64 + lane] = 0; continue; }
        const uint4* base = (const uint4*)(st->hist1 + (size_t)b * 4096) + (lane & 15);
        unsigned int a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll 8
        for (int r = lane >> 4; r < 64; r += 4) {
            const uint4 q = base[r * 16];
            a0 += q.x; a1 += q.y; a2 += q.z; a3 += q.w;
        }
        a0 += __shfl_xor(a0, 16); a1 += __shfl_xor(a1, 16); a2 += __shfl_xor(a2, 16); a3 += __shfl_xor(a3, 16);
        a0 += __shfl_xor(a0, 32); a1 += __shfl_xor(a1, 32); a2 += __shfl_xor(a2, 32); a3 += __shfl_xor(a3, 32);
        if (lane < 16) *(uint4*)(s_g + b * 64 + 4 * lane) = make_uint4(a0, a1, a2, a3);
    }
    __syncthreads();
    const int per = (NF_L1 / 64) / nthr;                                        // groups per thread: 4, 2 or 1
    unsigned long long mine = 0;
    for (int j = 0; j < per; ++j) mine += s_g[tid * per + j];
    unsigned long long incl = wave_incl_scan_u64(mine, lane);
    if (lane == 63) s_wtot[wave] = incl;
    __syncthreads();
    for (int w = 0; w < wave; ++w) incl += s_wtot[w];
    const unsigned long long excl = incl - mine;
    for (int t = 0; t < nt; ++t) {
        const unsigned long long rank = (unsigned long long)a.ranks[t];
        if (rank >= excl && rank < incl) { s_owner[t] = tid; s_before[t] = excl; }
    }
    __syncthreads();
    // one wave per target: the group among the owner's, then the group's 64 bins across the lanes.  The loads of a
    // wave's targets are issued together (this workgroup is the tail of the kernel: every round trip counts)
    constexpr int TB = 8;
    for (int t0 = wave; t0 < nt; t0 += nw * TB) {
        unsigned long long cum[TB], cnt[TB];
        int g[TB];
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            const int t = t0 + j * nw;
            cum[j] = 0; cnt[j] = 0; g[j] = 0;
            if (t < nt) {
                const unsigned long long rank = (unsigned long long)a.ranks[t];
                cum[j] = s_before[t];
                g[j] = s_owner[t] * per;
                for (int i = 0; i < per - 1; ++i) {
                    const unsigned long long c = s_g[s_owner[t] * per + i];
                    if (g[j] == s_owner[t] * per + i && rank >= cum[j] + c) { cum[j] += c; ++g[j]; }
                }
                cnt[j] = st->hist1[nf_hpos((unsigned int)g[j] * 64u + (unsigned int)lane)];
            }
        }
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            const int t = t0 + j * nw;
            if (t < nt) {
                const unsigned long long rank = (unsigned long long)a.ranks[t];
                const unsigned long long bi = wave_incl_scan_u64(cnt[j], lane) + cum[j];
                const unsigned long long m = __ballot(bi > rank);
                const int src = m ? (__ffsll((long long)m) - 1) : 63;
                const unsigned long long before = __shfl(bi - cnt[j], src);     // elements below the bin
                const unsigned long long inbin = __shfl(cnt[j], src);
                if (lane == 0) {
                    const unsigned int np = (unsigned int)g[j] * 64u + (unsigned int)src;
                    st->tgt_rank[t] = (long long)(rank - before);
                    st->tgt_prefix[t] = np;
                    s_newp[t] = np;
                    s_pcnt[t] = (unsigned int)inbin;
                }
            }
        }
    }
    __syncthreads();
    // slots = sorted distinct prefixes
    if (tid == 0) *s_ns = 0;
    if (tid < nt) {
        bool first = true;
        for (int u = 0; u < tid; ++u) first = first && (s_newp[u] != s_newp[tid]);
        s_first[tid] = first ? 1 : 0;
    }
    __syncthreads();
    if (tid < nt) {
        const unsigned int p = s_newp[tid];
        int pos = 0;
        for (int u = 0; u < nt; ++u) pos += (s_first[u] && s_newp[u] < p) ? 1 : 0;
        s_tslot[tid] = pos;
        if (s_first[tid]) { s_slotp[pos] = p; s_cnt[pos] = s_pcnt[tid]; atomicAdd(s_ns, 1); }
    }
    __syncthreads();
    const int ns = *s_ns;
    if (tid < nt) st->tgt_slot[tid] = s_tslot[tid];
    if (tid < ns) {
        unsigned int off = 0;
        for (int u = 0; u < tid; ++u) off += (s_cnt[u] + 7u) & ~7u;             // 16-byte aligned ranges of 2-byte entries
        st->slot_prefix[tid] = s_slotp[tid];
        st->slot_cnt[tid] = s_cnt[tid];
        st->slot_off[tid] = off;
    }
    if (tid == 0) { st->nslots = ns; st->nt = nt; st->nq = a.nq; }
}

// every workgroup calls this after its last global atomic of the sweep; true in exactly one workgroup: the last.
// What the last workgroup reads was written by device-scope atomics only, and those are performed at the point all XCDs
// share: waiting for their acknowledgement (vmcnt) orders them before the ticket.  A release fence (__threadfence) would
// also write back every dirty line of the XCD's L2 -- the maps the same kernel has just stored -- once per workgroup:
// ~70 us per frame with the ~1000 workgroups of the box kernels.
__device__ inline bool nf_arrive_last(unsigned int* ticket, unsigned int nblocks) {
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this thread's atomics have been performed
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(ticket, 1u) == nblocks - 1) ? 1 : 0;
    __syncthreads();
    if (s_last) __threadfence();                           // acquire: nothing of the other workgroups is read from a stale cache line
    return s_last != 0;
}


// host helpers (nle_fast.hip)
int nf_make_args(size_t n, const double* q_host, int nq, NfArgs* a);
// sweep 1 on a workspace (reset: zero the state first; false when an earlier kernel has already put the frame maximum there)
int nf_launch_stats(const float* lap, const float* mean, size_t n, int width, const double* q_host, int nq, void* ws,
                    hipStream_t st, bool reset);
static inline size_t nf_state_bytes() { return (sizeof(NleState) + 255) & ~(size_t)255; }
