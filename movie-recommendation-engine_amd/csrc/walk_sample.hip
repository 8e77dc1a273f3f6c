// walk_sample.hip -- random-walk neighbour sampling on gfx950.
//
// Replaces RandomWalkSampler._single_walk / sample_neighbors / batch_sample_neighbors
// (reference utils/random_walk.py:52-142).  One 64-lane wave owns one start node:
//   1. walk phase   : lanes = walks (two per lane); each step is searchsorted(cdf[row], u, 'right') on the
//                     fp64 per-row CDF (the arithmetic np.random.choice(p=...) performs, :76-79) -- in LDS for
//                     the start row all walks share (step 0), from one 64- or 32-byte bucket record otherwise, through
//                     the packed blocks / plain arrays as fallbacks (all bit-identical); with destination records
//                     (ps_dest_info_build) the row of a walk's second node comes out of LDS with the first pick;
//                     visited ids are staged in LDS in Counter-insertion order (walk-major).
//   2. count phase  : an LDS open-addressing table keyed by node id gives every position its
//                     visit count (ds atomics) and first-visit position (atomic min).
//   3. select phase : classes of equal count are swept from the highest count down; inside a
//                     class, a wave ballot + prefix popcount over positions gives the
//                     first-visit rank, i.e. python's stable sorted(..., reverse=True)[:T] (:107).
// No global atomics, no inter-wave communication: results do not depend on scheduling.
#include "ps_common.h"

namespace {

struct WalkArgs {
    const int64_t *rowptr;
    const int32_t *col;
    const double *cdf;
    int64_t V;
    const int64_t *starts;
    int64_t B;
    int W, L, T;
    int rng_mode;
    const double *uniforms;
    const int64_t *uoff;
    uint32_t seed_lo, seed_hi, call;
    const uint32_t *nodeinfo;
    const int32_t *guide;
    const unsigned char *packed;
    int32_t *ids;
    int32_t *counts;
    int32_t *nvalid;
    int hs_log2;
    int stage_blocks;   // start rows of at most this many 128-byte blocks are searched in LDS
    int region_words;   // LDS words shared by the staged row (walk phase) and the hash table (count phase)
    const unsigned char *buckets;   // 64-byte bucket records (ps_bucket_build), 32-byte half records (ps_bucket_build_half) or NULL
    int half_buckets;               // 1: `buckets` holds the 32-byte form (PS_WALK_HALF_BUCKETS)
    int bitmap_words;               // LDS words of the count-class bitmap: W * L / 32 + 1, padded
    int rounds;                     // independent samples per start node (one per GCN layer), all in one wave
    int64_t round_stride;           // PS_RNG_STREAM: uniforms of round r start at r * round_stride + uoff[i]
    const uint2 *dest_info;         // per edge e: (row start, degree) of col[e] (ps_dest_info_build) or NULL
    int split;                      // 1: a wave walks ONE round of a start node (wave index = round * B + node) instead of all of them
};

// One Philox4x32-10 block = the uniforms of two consecutive steps of a walk: counter (node, walk, step / 2, call);
// words (0, 1) -> even step, (2, 3) -> odd step.
__device__ __forceinline__ void philox_uniform2(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2,
                                                uint32_t c3, double &u_even, double &u_odd) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    // genrand_res53 combination of two 32-bit words
    u_even = ((double)(c0 >> 5) * 67108864.0 + (double)(c1 >> 6)) * (1.0 / 9007199254740992.0);
    u_odd = ((double)(c2 >> 5) * 67108864.0 + (double)(c3 >> 6)) * (1.0 / 9007199254740992.0);
}

// PS_RNG_STREAM_RAW: the stream as the generator leaves it (untempered MT19937 state words); uniform i = words 2i, 2i+1,
// tempered and combined like numpy's random_sample (genrand_res53) -- saves the separate conversion pass over the stream
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}
__device__ __forceinline__ double stream_uniform(const double *uniforms, int64_t i, bool raw) {
    if (!raw) return uniforms[i];
    const uint2 w = reinterpret_cast<const uint2 *>(uniforms)[i];
    return ((double)(mt_temper(w.x) >> 5) * 67108864.0 + (double)(mt_temper(w.y) >> 6)) * (1.0 / 9007199254740992.0);
}

__device__ __forceinline__ int64_t uniform_i64(int64_t v) {
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

constexpr int LIN_PROBES = 12;   // forward-scan probes after the guide lookup before falling back to bisection

// (row start, out-degree) of node v: one 8-byte record when nodeinfo is present
typedef uint32_t eidx_t;   // edge index: E < 2^32 is enforced by ps_csr_build

__device__ __forceinline__ void load_row(const int64_t *rowptr, const uint32_t *nodeinfo, int64_t v, eidx_t &lo,
                                         eidx_t &hi) {
    if (nodeinfo) {
        const uint2 ni = reinterpret_cast<const uint2 *>(nodeinfo)[v];
        lo = ni.x;
        hi = lo + ni.y;
    } else {
        lo = (eidx_t)rowptr[v];
        hi = (eidx_t)rowptr[v + 1];
    }
}

// Edge-record accessors: three arrays, or the interleaved 128-byte blocks [8 cdf | 8 col | 8 guide]
__device__ __forceinline__ double edge_cdf(const double *cdf, const unsigned char *packed, eidx_t e) {
    return packed ? reinterpret_cast<const double *>(packed + (size_t)(e >> 3) * 128)[e & 7] : cdf[e];
}
__device__ __forceinline__ int32_t edge_col(const int32_t *col, const unsigned char *packed, eidx_t e) {
    return packed ? reinterpret_cast<const int32_t *>(packed + (size_t)(e >> 3) * 128 + 64)[e & 7] : col[e];
}

// cdf[e], cdf[e + 1] with one 16-byte load per lane (8-byte aligned; inside a packed block the pair must not
// leave the block's 8 CDF slots)
struct __attribute__((packed, aligned(8))) CdfPair { double c0, c1; };
__device__ __forceinline__ bool cdf_pair_ok(const unsigned char *packed, eidx_t e) { return !packed || (e & 7) < 7; }
__device__ __forceinline__ void edge_cdf_pair(const double *cdf, const unsigned char *packed, eidx_t e, double &c0,
                                              double &c1) {
    const CdfPair *p = packed ? reinterpret_cast<const CdfPair *>(packed + (size_t)(e >> 3) * 128 + (e & 7) * 8)
                              : reinterpret_cast<const CdfPair *>(cdf + e);
    const CdfPair v = *p;
    c0 = v.c0;
    c1 = v.c1;
}

struct __attribute__((packed, aligned(4))) ColPair { int32_t k0, k1; };
__device__ __forceinline__ void edge_col_pair(const int32_t *col, const unsigned char *packed, eidx_t e, int32_t &k0,
                                              int32_t &k1) {
    const ColPair *p = packed ? reinterpret_cast<const ColPair *>(packed + (size_t)(e >> 3) * 128 + 64 + (e & 7) * 4)
                              : reinterpret_cast<const ColPair *>(col + e);
    const ColPair v = *p;
    k0 = v.k0;
    k1 = v.k1;
}

// The same accessors over a copy of the start row's packed blocks in LDS (block b0 is at `base`): step 0 of every
// walk of a wave reads the SAME row, so rows of up to `stage_blocks` blocks are streamed into LDS once (coalesced
// 16 B per lane) and searched there instead of through ~5 dependent gathers per walk.
struct GlobalEdges {
    const double *cdf;
    const int32_t *col;
    const int32_t *guide;
    const unsigned char *packed;
    __device__ __forceinline__ bool has_guide() const { return guide || packed; }
    __device__ __forceinline__ bool pair_ok(eidx_t e) const { return cdf_pair_ok(packed, e); }
    __device__ __forceinline__ double c(eidx_t e) const { return edge_cdf(cdf, packed, e); }
    __device__ __forceinline__ int32_t k(eidx_t e) const { return edge_col(col, packed, e); }
    __device__ __forceinline__ void c2(eidx_t e, double &c0, double &c1) const { edge_cdf_pair(cdf, packed, e, c0, c1); }
    __device__ __forceinline__ void k2(eidx_t e, int32_t &k0, int32_t &k1) const { edge_col_pair(col, packed, e, k0, k1); }
    __device__ __forceinline__ int32_t g(eidx_t e) const {
        return packed ? reinterpret_cast<const int32_t *>(packed + (size_t)(e >> 3) * 128 + 96)[e & 7] : guide[e];
    }
};

struct LdsEdges {
    const unsigned char *base;   // LDS
    eidx_t b0;
    __device__ __forceinline__ const unsigned char *blk(eidx_t e) const { return base + ((e >> 3) - b0) * 128; }
    __device__ __forceinline__ bool has_guide() const { return true; }
    __device__ __forceinline__ bool pair_ok(eidx_t e) const { return (e & 7) < 7; }
    __device__ __forceinline__ double c(eidx_t e) const { return reinterpret_cast<const double *>(blk(e))[e & 7]; }
    __device__ __forceinline__ int32_t k(eidx_t e) const { return reinterpret_cast<const int32_t *>(blk(e) + 64)[e & 7]; }
    __device__ __forceinline__ void c2(eidx_t e, double &c0, double &c1) const {
        const double *p = reinterpret_cast<const double *>(blk(e)) + (e & 7);
        c0 = p[0];
        c1 = p[1];
    }
    __device__ __forceinline__ void k2(eidx_t e, int32_t &k0, int32_t &k1) const {
        const int32_t *p = reinterpret_cast<const int32_t *>(blk(e) + 64) + (e & 7);
        k0 = p[0];
        k1 = p[1];
    }
    __device__ __forceinline__ int32_t g(eidx_t e) const { return reinterpret_cast<const int32_t *>(blk(e) + 96)[e & 7]; }
};

// searchsorted(cdf[lo:hi], u, side='right') for two independent walks in lockstep -> destination ids nA, nB.
// With a guide table the search starts at bucket floor(u * deg) (guide = #{cdf <= (j/deg)(1-2^-50)} <= answer) and scans
// forward: each iteration looks at two consecutive entries and their destinations (one 16-byte and one 8-byte load per
// lane, issued together so the dependent chain stays one gather per iteration; fetching the destination after the search
// had settled measured 6 % slower); after LIN_PROBES entries, or without a guide, it bisects.
template <class Acc>
__device__ __forceinline__ void search_two(const Acc &acc, bool aliveA, eidx_t loA, eidx_t hiA, double uA, bool aliveB,
                                           eidx_t loB, eidx_t hiB, double uB, int32_t &nA, int32_t &nB, eidx_t &eA, eidx_t &eB) {
    eidx_t lA = loA, hA = loA, lB = loB, hB = loB;
    int nA_ = LIN_PROBES, nB_ = LIN_PROBES;
    if (acc.has_guide()) {
        if (aliveA) {
            const uint32_t deg = (uint32_t)(hiA - loA);
            uint32_t j = (uint32_t)(uA * (double)deg);
            if (j >= deg) j = deg - 1;
            lA = loA + (eidx_t)acc.g(loA + j);
            nA_ = 0;
        }
        if (aliveB) {
            const uint32_t deg = (uint32_t)(hiB - loB);
            uint32_t j = (uint32_t)(uB * (double)deg);
            if (j >= deg) j = deg - 1;
            lB = loB + (eidx_t)acc.g(loB + j);
            nB_ = 0;
        }
    }
    if (aliveA) hA = hiA;
    if (aliveB) hB = hiB;
    nA = -1;
    nB = -1;
    while (true) {
        const bool a_ = lA < hA, b_ = lB < hB;
        if (!a_ && !b_) break;
        const bool linA = nA_ < LIN_PROBES, linB = nB_ < LIN_PROBES;
        const eidx_t mA = linA ? lA : lA + ((hA - lA) >> 1);
        const eidx_t mB = linB ? lB : lB + ((hB - lB) >> 1);
        const bool a2 = a_ && linA && (mA + 1 < hA) && acc.pair_ok(mA);
        const bool b2 = b_ && linB && (mB + 1 < hB) && acc.pair_ok(mB);
        double cA0 = 0.0, cA1 = 2.0, cB0 = 0.0, cB1 = 2.0;
        int32_t kA0 = -1, kA1 = -1, kB0 = -1, kB1 = -1;
        if (a2) { acc.c2(mA, cA0, cA1); acc.k2(mA, kA0, kA1); }
        else if (a_) { cA0 = acc.c(mA); if (linA) kA0 = acc.k(mA); }
        if (b2) { acc.c2(mB, cB0, cB1); acc.k2(mB, kB0, kB1); }
        else if (b_) { cB0 = acc.c(mB); if (linB) kB0 = acc.k(mB); }
        if (a_) {
            if (linA) {
                if (cA0 > uA) { hA = mA; lA = mA; nA = kA0; }
                else if (a2 && cA1 > uA) { lA = mA + 1; hA = lA; nA = kA1; }
                else { lA = mA + (a2 ? 2 : 1); nA_ += a2 ? 2 : 1; }
            } else {
                if (cA0 <= uA) lA = mA + 1; else hA = mA;
            }
        }
        if (b_) {
            if (linB) {
                if (cB0 > uB) { hB = mB; lB = mB; nB = kB0; }
                else if (b2 && cB1 > uB) { lB = mB + 1; hB = lB; nB = kB1; }
                else { lB = mB + (b2 ? 2 : 1); nB_ += b2 ? 2 : 1; }
            } else {
                if (cB0 <= uB) lB = mB + 1; else hB = mB;
            }
        }
    }
    if (aliveA) { if (lA >= hiA) lA = hiA - 1; if (nA < 0) nA = acc.k(lA); }
    if (aliveB) { if (lB >= hiB) lB = hiB - 1; if (nB < 0) nB = acc.k(lB); }
    eA = lA;                                             // the chosen edge (every branch above leaves l on it)
    eB = lB;
}
template <class Acc>
__device__ __forceinline__ void search_two(const Acc &acc, bool aliveA, eidx_t loA, eidx_t hiA, double uA, bool aliveB,
                                           eidx_t loB, eidx_t hiB, double uB, int32_t &nA, int32_t &nB) {
    eidx_t eA, eB;
    search_two(acc, aliveA, loA, hiA, uA, aliveB, loB, hiB, uB, nA, nB, eA, eB);
}

// The same search through the 64-byte bucket records: record lo + floor(u * deg) holds the five CDF entries starting
// at the bucket's guide position and their destinations, so one 64-byte sector (three 16-byte loads per lane, a
// fourth for the rare fifth candidate) replaces the dependent guide -> CDF -> destination gathers.  A lane whose answer
// lies beyond the fifth candidate (>= 5 CDF entries inside one 1/deg-wide bucket) repeats the search the long way.
struct BucketHit {
    double c0, c1, c2, c3;
    int32_t k0, k1, k2, k3;
};
__device__ __forceinline__ const unsigned char *bucket_of(const unsigned char *buckets, eidx_t lo, eidx_t hi, double u) {
    const uint32_t deg = (uint32_t)(hi - lo);
    uint32_t j = (uint32_t)(u * (double)deg);
    if (j >= deg) j = deg - 1;
    return buckets + (size_t)(lo + j) * 64;
}
__device__ __forceinline__ BucketHit bucket_load(const unsigned char *r) {
    const double2 q0 = reinterpret_cast<const double2 *>(r)[0];
    const double2 q1 = reinterpret_cast<const double2 *>(r)[1];
    const int4 q2 = reinterpret_cast<const int4 *>(r)[2];
    return BucketHit{q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, q2.z, q2.w};
}
__device__ __forceinline__ int32_t bucket_pick(const BucketHit &h, double u) {   // -1: beyond the fourth candidate
    return h.c0 > u ? h.k0 : h.c1 > u ? h.k1 : h.c2 > u ? h.k2 : h.c3 > u ? h.k3 : -1;
}

template <class Acc>
__device__ __forceinline__ void search_two_buckets(const unsigned char *buckets, const Acc &acc, bool aliveA, eidx_t loA,
                                                   eidx_t hiA, double uA, bool aliveB, eidx_t loB, eidx_t hiB, double uB,
                                                   int32_t &nA, int32_t &nB) {
    nA = -1;
    nB = -1;
    const unsigned char *rA = nullptr, *rB = nullptr;
    BucketHit hA{}, hB{};
    if (aliveA) { rA = bucket_of(buckets, loA, hiA, uA); hA = bucket_load(rA); }
    if (aliveB) { rB = bucket_of(buckets, loB, hiB, uB); hB = bucket_load(rB); }
    if (aliveA) nA = bucket_pick(hA, uA);
    if (aliveB) nB = bucket_pick(hB, uB);
    bool moreA = aliveA && nA < 0, moreB = aliveB && nB < 0;
    if (__ballot(moreA || moreB) != 0ull) {                      // fifth candidate
        if (moreA && reinterpret_cast<const double *>(rA)[6] > uA) { nA = reinterpret_cast<const int32_t *>(rA)[14]; moreA = false; }
        if (moreB && reinterpret_cast<const double *>(rB)[6] > uB) { nB = reinterpret_cast<const int32_t *>(rB)[14]; moreB = false; }
        if (__ballot(moreA || moreB) != 0ull) {                  // the long way for what is left
            int32_t fA = -1, fB = -1;
            search_two(acc, moreA, loA, hiA, uA, moreB, loB, hiB, uB, fA, fB);
            if (moreA) nA = fA;
            if (moreB) nB = fB;
        }
    }
}

// 32-byte half records (graphs whose 64-byte records would not fit, BASELINE config 5: 64 GB instead of 128 at 2 x 10^9 edges):
// [c0 c1 c2 c3 | k0 k1 k2 k3] = four candidates' CDF entries rounded DOWN to fp32 and their destinations.  cdf_i lies in
// [lo_i, hi_i) with lo_i = (double)c_i, hi_i = the next float: the first i with u < hi_i is the pick if also u < lo_i (then
// cdf_{i-1} <= u < cdf_i is proven); a u inside a sliver [lo_i, hi_i), or beyond the fourth candidate, repeats the search the long
// way.  Two 16-byte loads from one half of a 64-byte sector.
__device__ __forceinline__ int32_t half_pick(const float4 &c, const int4 &k, double u) {
    const double l0 = (double)c.x, l1 = (double)c.y, l2 = (double)c.z, l3 = (double)c.w;
    const double h0 = (double)__uint_as_float(__float_as_uint(c.x) + 1u), h1 = (double)__uint_as_float(__float_as_uint(c.y) + 1u);
    const double h2 = (double)__uint_as_float(__float_as_uint(c.z) + 1u), h3 = (double)__uint_as_float(__float_as_uint(c.w) + 1u);
    if (u < h0) return u < l0 ? k.x : -1;
    if (u < h1) return u < l1 ? k.y : -1;
    if (u < h2) return u < l2 ? k.z : -1;
    if (u < h3) return u < l3 ? k.w : -1;
    return -1;
}
template <class Acc>
__device__ __forceinline__ void search_two_half(const unsigned char *buckets, const Acc &acc, bool aliveA, eidx_t loA, eidx_t hiA,
                                                double uA, bool aliveB, eidx_t loB, eidx_t hiB, double uB, int32_t &nA, int32_t &nB) {
    nA = -1;
    nB = -1;
    float4 cA = make_float4(0.f, 0.f, 0.f, 0.f), cB = cA;
    int4 kA = make_int4(-1, -1, -1, -1), kB = kA;
    if (aliveA) {
        const uint32_t deg = (uint32_t)(hiA - loA);
        uint32_t j = (uint32_t)(uA * (double)deg);
        if (j >= deg) j = deg - 1;
        const unsigned char *r = buckets + (size_t)(loA + j) * 32;
        cA = reinterpret_cast<const float4 *>(r)[0];
        kA = reinterpret_cast<const int4 *>(r)[1];
    }
    if (aliveB) {
        const uint32_t deg = (uint32_t)(hiB - loB);
        uint32_t j = (uint32_t)(uB * (double)deg);
        if (j >= deg) j = deg - 1;
        const unsigned char *r = buckets + (size_t)(loB + j) * 32;
        cB = reinterpret_cast<const float4 *>(r)[0];
        kB = reinterpret_cast<const int4 *>(r)[1];
    }
    if (aliveA) nA = half_pick(cA, kA, uA);
    if (aliveB) nB = half_pick(cB, kB, uB);
    const bool moreA = aliveA && nA < 0, moreB = aliveB && nB < 0;
    if (__ballot(moreA || moreB) != 0ull) {                      // the long way for what is left
        int32_t fA = -1, fB = -1;
        search_two(acc, moreA, loA, hiA, uA, moreB, loB, hiB, uB, fA, fB);
        if (moreA) nA = fA;
        if (moreB) nB = fB;
    }
}

// Start state of searchsorted(cdf[lo:hi], u, 'right'): with a guide table the search starts at the bucket
// floor(u * deg) (guide = #{cdf <= (j-1)/deg} <= answer) and first scans forward; `n` counts probes.
__device__ __forceinline__ void search_init(const int32_t *guide, const unsigned char *packed, eidx_t lo, eidx_t hi,
                                            double u, eidx_t &l, int &n) {
    l = lo;
    n = LIN_PROBES;
    if (guide || packed) {
        const uint32_t deg = (uint32_t)(hi - lo);
        uint32_t j = (uint32_t)(u * (double)deg);
        if (j >= deg) j = deg - 1;
        const eidx_t e = lo + j;
        l = lo + (eidx_t)(packed ? reinterpret_cast<const int32_t *>(packed + (size_t)(e >> 3) * 128 + 96)[e & 7] : guide[e]);
        n = 0;
    }
}

#ifndef PS_WS_DEBUG
#define PS_WS_DEBUG 0      // knock-outs (tools/ws_knockout.sh; wrong results): 1 step 0 without its search, 2 walks stop after step 0,
#endif                     // 4 no count / select phases, 8 no uniforms (constant 0.5)
#if PS_WS_DEBUG & 16      // timeline (tools/ws_trace.py): s_memtime at the phase boundaries of the first 16384 start nodes
__device__ unsigned long long ps_ws_trace_buf[16384 * 12];
#define PS_WS_STAMP(slot) do { if (lane == 0 && i < 16384) ps_ws_trace_buf[i * 12 + (slot)] = __builtin_readcyclecounter(); } while (0)
extern "C" int ps_debug_ws_trace(unsigned long long *host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ps_ws_trace_buf), sizeof(unsigned long long) * 16384 * 12);
}
#else
#define PS_WS_STAMP(slot) do {} while (0)
#endif
constexpr int WAVES_PER_BLOCK = 1;   // one start node per workgroup: the dispatcher load-balances uneven nodes
constexpr int BITMAP_WORDS = 40;   // counts <= 1024 -> 33 words, padded (larger W * L: WalkArgs::bitmap_words)

// STREAM: compiled per RNG mode (the Philox variant carries no stream addressing, the stream variant no Philox state:
// 83-85 VGPRs instead of 90 for both in one kernel, which is what leaves room for the up-front uniform loads below)
// DEST: compiled with / without the staged destination records (the extra live values take the stream variant from 95 to 101 VGPRs =
// four waves per SIMD instead of five; launches without dest_info keep the kernel they had)
template <int NP, bool STREAM, bool DEST = false>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void walk_sample_kernel(WalkArgs a) {
    extern __shared__ int32_t smem[];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int HS = 1 << a.hs_log2;
    const int P = a.W * a.L;
    const int R = a.split ? 1 : a.rounds;                  // rounds walked by one wave
    const int per_wave = R * NP * 64 + a.region_words + a.bitmap_words;
    int32_t *posb_all = smem + wv * per_wave;               // visited ids of every round, [round][walk * L + step]
    int32_t *hkey = posb_all + R * NP * 64;
    int32_t *hcnt = hkey + HS;
    int32_t *hfirst = hcnt + HS;
    uint32_t *bitmap = reinterpret_cast<uint32_t *>(hkey + a.region_words);
    const GlobalEdges grow{a.cdf, a.col, a.guide, a.packed};
    const int nbw = (P >> 5) + 1;
    const uint64_t lanemask_lt = (1ull << lane) - 1ull;

    const int64_t nwork = a.split ? a.B * a.rounds : a.B;
    for (int64_t wi = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wv; wi < nwork; wi += (int64_t)gridDim.x * WAVES_PER_BLOCK) {
        const int64_t i = a.split ? wi % a.B : wi;
        const int rd0 = a.split ? (int)(wi / a.B) : 0;      // first round of this wave
        const int64_t s = uniform_i64(a.starts[i]);
        eidx_t lo0 = 0, hi0 = 0;
        if (s >= 0 && s < a.V) {
            lo0 = (eidx_t)uniform_i64(a.rowptr[s]);
            hi0 = (eidx_t)uniform_i64(a.rowptr[s + 1]);
        }
        if (hi0 == lo0) {   // isolated start node -> ([], [])  (random_walk.py:109-110)
            for (int r = rd0; r < rd0 + R; ++r) {
                for (int t = lane; t < a.T; t += 64) { a.ids[(r * a.B + i) * a.T + t] = -1; a.counts[(r * a.B + i) * a.T + t] = 0; }
                if (lane == 0) a.nvalid[r * a.B + i] = 0;
            }
            continue;
        }
        PS_WS_STAMP(0);
        constexpr bool stream = STREAM;
        const bool raw = a.rng_mode == PS_RNG_STREAM_RAW;
        // PS_RNG_STREAM_WALKS (graphs with reachable sinks): uoff holds one stream position PER WALK, [B * W] -- a walk that stops at a
        // sink consumes fewer than L uniforms (utils/random_walk.py:68-69), so the positions are not a multiple of W * L apart
        const bool perwalk = stream && a.rng_mode == PS_RNG_STREAM_WALKS;
        const int64_t ubase0 = (stream && !perwalk) ? uniform_i64(a.uoff[i]) : 0;

        // ---------------- walk phase: all rounds, the start row is staged once ----------
        for (int j = lane; j < R * NP * 64; j += 64) posb_all[j] = -1;
        // the start row's packed blocks -> LDS (over the hash-table area, which is initialised after the walks)
        const eidx_t b0 = lo0 >> 3, nblk = ((hi0 - 1) >> 3) - b0 + 1;
        const bool staged = a.packed != nullptr && nblk <= (eidx_t)a.stage_blocks;
        // with dest_info the (row start, degree) records of the row's destinations follow the blocks in LDS: a walk then knows
        // the row of its step-1 node when step 0 has picked it, and step 1 is ONE dependent gather (the bucket record) instead
        // of two (node record, then bucket record)
        const bool have_dest = DEST && staged && a.dest_info != nullptr;
        const uint2 *ldest = reinterpret_cast<const uint2 *>(hkey + nblk * 32);      // LDS: [hi0 - lo0] records
        if (staged) {
            const uint4 *src = reinterpret_cast<const uint4 *>(a.packed + (size_t)b0 * 128);
            uint4 *dst = reinterpret_cast<uint4 *>(hkey);
            for (int q = lane; q < (int)nblk * 8; q += 64) dst[q] = src[q];
            if (have_dest) {
                uint2 *dd = reinterpret_cast<uint2 *>(hkey + nblk * 32);
                for (eidx_t e = lo0 + lane; e < hi0; e += 64) dd[e - lo0] = a.dest_info[e];
            }
        }
        const LdsEdges lrow{reinterpret_cast<const unsigned char *>(hkey), b0};
        ps_wave_lds_sync();
        PS_WS_STAMP(1);
        // two walks per lane (w and w + 64) advance in lockstep: their CDF probes are independent, so
        // every iteration of the search loop keeps two loads in flight per lane.
        for (int rd = 0; rd < R; ++rd) {
        int32_t *posb = posb_all + rd * NP * 64;
        const int64_t ubase = ubase0 + (rd0 + rd) * a.round_stride;
        const uint32_t call = a.call + (uint32_t)(rd0 + rd);
        for (int w0 = 0; w0 < a.W; w0 += 128) {
            const int wA = w0 + lane, wB = w0 + 64 + lane;
            const bool actA = wA < a.W, actB = wB < a.W;
            bool aliveA = actA, aliveB = actB;
            int32_t curA = (int32_t)s, curB = (int32_t)s;
            eidx_t edA = lo0, edB = lo0;                     // edge taken by step 0 (inside the staged row)
            double uA1 = 2.0, uB1 = 2.0;
            // stream mode, raw words, L = 2 (the reference's default walk): the four words of a walk's two uniforms are one
            // aligned 16-byte load, requested before the walk starts (uoff is a multiple of W * L, so 2 w is even)
            const bool pre = stream && raw && a.L == 2;
            uint4 pwA = make_uint4(0u, 0u, 0u, 0u), pwB = pwA;
            if (pre) {
                const uint4 *src = reinterpret_cast<const uint4 *>(a.uniforms);
                if (actA) pwA = src[(ubase >> 1) + wA];
                if (actB) pwB = src[(ubase >> 1) + wB];
            }
            for (int st = 0; st < a.L; ++st) {
                eidx_t loA = lo0, hiA = hi0, loB = lo0, hiB = hi0;
                if (st == 1 && have_dest) {
                    if (aliveA) { const uint2 d = ldest[edA - lo0]; loA = d.x; hiA = d.x + d.y; }
                    if (aliveB) { const uint2 d = ldest[edB - lo0]; loB = d.x; hiB = d.x + d.y; }
                } else if (st > 0) {
                    if (aliveA) load_row(a.rowptr, a.nodeinfo, curA, loA, hiA);
                    if (aliveB) load_row(a.rowptr, a.nodeinfo, curB, loB, hiB);
                }
                if (hiA == loA) aliveA = false;       // sink: the walk stops (random_walk.py:68-69)
                if (hiB == loB) aliveB = false;
                double uA = 2.0, uB = 2.0;
                if (stream) {
                    if (pre) {
                        const uint32_t a0 = st ? pwA.z : pwA.x, a1 = st ? pwA.w : pwA.y, b0 = st ? pwB.z : pwB.x, b1 = st ? pwB.w : pwB.y;
                        if (aliveA) uA = ((double)(mt_temper(a0) >> 5) * 67108864.0 + (double)(mt_temper(a1) >> 6)) * (1.0 / 9007199254740992.0);
                        if (aliveB) uB = ((double)(mt_temper(b0) >> 5) * 67108864.0 + (double)(mt_temper(b1) >> 6)) * (1.0 / 9007199254740992.0);
                    } else if (perwalk) {
                        if (aliveA) uA = a.uniforms[a.uoff[i * a.W + wA] + st];
                        if (aliveB) uB = a.uniforms[a.uoff[i * a.W + wB] + st];
                    } else {
                        if (aliveA) uA = stream_uniform(a.uniforms, ubase + (int64_t)wA * a.L + st, raw);
                        if (aliveB) uB = stream_uniform(a.uniforms, ubase + (int64_t)wB * a.L + st, raw);
                    }
                } else if ((st & 1) == 0) {
                    philox_uniform2(a.seed_lo, a.seed_hi, (uint32_t)s, (uint32_t)wA, (uint32_t)(st >> 1), call, uA, uA1);
                    philox_uniform2(a.seed_lo, a.seed_hi, (uint32_t)s, (uint32_t)wB, (uint32_t)(st >> 1), call, uB, uB1);
                } else {
                    uA = uA1;
                    uB = uB1;
                }
                int32_t nA = -1, nB = -1;
                if (PS_WS_DEBUG & 8) { uA = 0.5; uB = 0.25; }
                if ((PS_WS_DEBUG & 2) && st > 0) { aliveA = false; aliveB = false; }
                if ((PS_WS_DEBUG & 1) && st == 0) { nA = grow.k(lo0 + (eidx_t)((uint32_t)lane % (uint32_t)(hi0 - lo0))); nB = nA; }
                else if (staged && st == 0) search_two(lrow, aliveA, loA, hiA, uA, aliveB, loB, hiB, uB, nA, nB, edA, edB);
                else if (a.buckets && a.half_buckets) search_two_half(a.buckets, grow, aliveA, loA, hiA, uA, aliveB, loB, hiB, uB, nA, nB);
                else if (a.buckets) search_two_buckets(a.buckets, grow, aliveA, loA, hiA, uA, aliveB, loB, hiB, uB, nA, nB);
                else search_two(grow, aliveA, loA, hiA, uA, aliveB, loB, hiB, uB, nA, nB);
                if (aliveA) curA = nA;
                if (aliveB) curB = nB;
                if (actA) posb[wA * a.L + st] = nA;
                if (actB) posb[wB * a.L + st] = nB;
                if (w0 == 0) PS_WS_STAMP(2 + rd * 2 + (st ? 1 : 0));
            }
        }
        }
        PS_WS_STAMP(6);
        for (int rd = 0; rd < ((PS_WS_DEBUG & 4) ? 0 : R); ++rd) {
        const int32_t *posb = posb_all + rd * NP * 64;
        int32_t *oid = a.ids + ((int64_t)(rd0 + rd) * a.B + i) * a.T;
        int32_t *ocn = a.counts + ((int64_t)(rd0 + rd) * a.B + i) * a.T;
        // ---------------- count phase -------------------------------------------------
        for (int h = lane; h < HS; h += 64) { hkey[h] = -1; hcnt[h] = 0; hfirst[h] = 0x7fffffff; }
        for (int b = lane; b < nbw; b += 64) bitmap[b] = 0u;
        ps_wave_lds_sync();
        // (inserting the NP positions of a lane together -- all pending compare-and-swaps of a probing round in flight, then the
        // count / first-visit atomics in one batch -- measured 2-4 x SLOWER for this phase: 22 K / 37 K cycles against 9.9 K)
        int32_t vid[NP], slot[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int p = j * 64 + lane;
            vid[j] = posb[p];
            slot[j] = -1;
            if (vid[j] >= 0) {
                uint32_t h = ((uint32_t)vid[j] * 2654435761u) >> (32 - a.hs_log2);
                while (true) {
                    const int old = atomicCAS(&hkey[h], -1, vid[j]);
                    if (old == -1 || old == vid[j]) break;
                    h = (h + 1) & (uint32_t)(HS - 1);
                }
                atomicAdd(&hcnt[h], 1);
                atomicMin(&hfirst[h], p);
                slot[j] = (int32_t)h;
            }
        }
        ps_wave_lds_sync();
        int32_t cr[NP];   // visit count if this position is the first visit of its node, else 0
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            cr[j] = 0;
            if (slot[j] >= 0 && hfirst[slot[j]] == j * 64 + lane) {
                cr[j] = hcnt[slot[j]];
                atomicOr(&bitmap[cr[j] >> 5], 1u << (cr[j] & 31));
            }
        }
        ps_wave_lds_sync();
        PS_WS_STAMP(7 + rd * 2);
        // ---------------- select phase ------------------------------------------------
        int emitted = 0;
        for (int wd = nbw - 1; wd >= 0 && emitted < a.T; --wd) {
            uint32_t bits = __builtin_amdgcn_readfirstlane(bitmap[wd]);
            while (bits != 0u && emitted < a.T) {
                const int b = 31 - __builtin_clz(bits);
                bits &= ~(1u << b);
                const int c = wd * 32 + b;
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const bool flag = cr[j] == c;
                    const uint64_t mask = __ballot(flag);
                    if (flag) {
                        const int r = emitted + __popcll(mask & lanemask_lt);
                        if (r < a.T) { oid[r] = vid[j]; ocn[r] = c; }
                    }
                    emitted += __popcll(mask);
                }
            }
        }
        const int nv = emitted < a.T ? emitted : a.T;
        for (int t = nv + lane; t < a.T; t += 64) { oid[t] = -1; ocn[t] = 0; }
        if (lane == 0) a.nvalid[(int64_t)(rd0 + rd) * a.B + i] = nv;
        ps_wave_lds_sync();
        PS_WS_STAMP(8 + rd * 2);
        }
    }
}

__global__ void walk_paths_kernel(const int64_t *rowptr, const int32_t *col, const double *cdf, int64_t V,
                                  const int64_t *starts, int64_t B, int L, int rng_mode, const double *uniforms,
                                  const int64_t *uoff, uint32_t k0, uint32_t k1, uint32_t call, int walk_mod,
                                  const uint32_t *nodeinfo, const int32_t *guide, int32_t *paths) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = starts[i];
        int64_t cur = s;
        bool alive = s >= 0 && s < V;
        const int64_t ubase = (rng_mode == PS_RNG_STREAM) ? uoff[i] : 0;
        for (int st = 0; st < L; ++st) {
            int32_t nxt = -1;
            if (alive) {
                eidx_t lo, hi;
                load_row(rowptr, nodeinfo, cur, lo, hi);
                if (hi == lo) {
                    alive = false;
                } else {
                    double u, u_odd;
                    if (rng_mode == PS_RNG_STREAM) {
                        u = uniforms[ubase + st];
                    } else {
                        philox_uniform2(k0, k1, (uint32_t)s, (uint32_t)(walk_mod > 0 ? i % walk_mod : i), (uint32_t)(st >> 1), call, u, u_odd);
                        if (st & 1) u = u_odd;
                    }
                    eidx_t l, h = hi;
                    int n;
                    search_init(guide, nullptr, lo, hi, u, l, n);
                    while (l < h) {
                        const eidx_t mid = (n < LIN_PROBES) ? l : l + ((h - l) >> 1);
                        if (cdf[mid] <= u) l = mid + 1; else h = mid;
                        ++n;
                    }
                    if (l >= hi) l = hi - 1;
                    nxt = col[l];
                    cur = nxt;
                }
            }
            paths[i * L + st] = nxt;
        }
    }
}

// uoff[i] = W*L * #{j < i : outdeg(starts[j]) > 0}.  One block; a pass covers 64 rows of 1024 start nodes: all
// (start id -> row bounds) loads of a pass are in flight together (the first version took one dependent global round trip
// per 1024 nodes: 92 us for 59 047), the flags stay in a 64-bit mask per thread, the 64 x 16 per-(row, wave) counts
// are scanned once in LDS.
__global__ __launch_bounds__(1024) void uniform_offsets_kernel(const int64_t *rowptr, int64_t V, const int64_t *starts,
                                                               int64_t B, int64_t WL, int64_t *uoff, int64_t *total) {
    __shared__ int cnt[64 * 16 + 1];
    __shared__ int wtot[16];
    __shared__ int64_t carry;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (t == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < B; base += 64 * 1024) {
        uint64_t mine = 0;                                   // bit r: my node of row r is active
#pragma unroll 8
        for (int r = 0; r < 64; ++r) {
            const int64_t i = base + (int64_t)r * 1024 + t;
            int act = 0;
            if (i < B) {
                const int64_t s = starts[i];
                act = (s >= 0 && s < V && rowptr[s + 1] > rowptr[s]) ? 1 : 0;
            }
            const uint64_t m = __ballot(act);
            mine |= (uint64_t)act << r;
            if (lane == 0) cnt[r * 16 + wv] = __popcll(m);
        }
        __syncthreads();
        // exclusive scan of the 1024 (row, wave) counts, element t per thread
        const int c = cnt[t];
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
        }
        if (lane == 63) wtot[wv] = incl;
        __syncthreads();
        int woff = 0, tot = 0;
        for (int k = 0; k < 16; ++k) { if (k < wv) woff += wtot[k]; tot += wtot[k]; }
        __syncthreads();
        cnt[t] = woff + incl - c;
        __syncthreads();
        const int64_t c0 = carry;
#pragma unroll 8
        for (int r = 0; r < 64; ++r) {
            const int64_t i = base + (int64_t)r * 1024 + t;
            const uint64_t m = __ballot((mine >> r) & 1ull);
            if (i < B) uoff[i] = (c0 + cnt[r * 16 + wv] + __popcll(m & ((1ull << lane) - 1ull))) * WL;
        }
        __syncthreads();
        if (t == 0) carry = c0 + tot;
        __syncthreads();
    }
    if (t == 0) total[0] = carry * WL;
}

__global__ void graph_stats_kernel(const int64_t *rowptr, const int32_t *col, int64_t E, int64_t V,
                                   unsigned long long *flags) {
    int64_t maxdeg = 0;
    int sink = 0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        const int32_t d = col[e];
        if (rowptr[d + 1] == rowptr[d]) sink = 1;
    }
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (int64_t)gridDim.x * blockDim.x) {
        const int64_t d = rowptr[v + 1] - rowptr[v];
        maxdeg = d > maxdeg ? d : maxdeg;
    }
    if (__any(sink) && (threadIdx.x & 63) == 0) atomicOr(&flags[0], 1ull);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int64_t other = __shfl_xor(maxdeg, o, 64);
        maxdeg = other > maxdeg ? other : maxdeg;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(&flags[1], (unsigned long long)maxdeg);
}

}  // namespace

// two-layer launches on SYN-25M, tools/ws_split_probe.py (us, one wave per node / per (node, layer)): 3 000 nodes 43.2 / 36.7, 7 381 69.2 / 59.1,
// 14 762 112.7 / 105.4, 29 524 200.3 / 208.2, 59 047 371.5 / 406.5
constexpr int64_t SPLIT_MAX_WAVES = 32768;

static int walk_sample_launch(const int64_t *rowptr, const int32_t *col, const double *cdf, int64_t V,
                              const int64_t *starts, int64_t B, int W, int L, int T, int rng_mode,
                              const double *uniforms, const int64_t *uoff, uint64_t seed, uint32_t call,
                              const uint32_t *nodeinfo, const int32_t *guide, const void *packed, const void *buckets,
                              const void *dest_info, int rounds, int64_t round_stride, int32_t *ids, int32_t *counts, int32_t *nvalid,
                              ps_stream_t stream) {
    const int half_buckets = (rng_mode & PS_WALK_HALF_BUCKETS) ? 1 : 0;      // a flag beside the RNG mode: `buckets` is the 32-byte form
    rng_mode &= ~PS_WALK_HALF_BUCKETS;
    if (B < 0 || W <= 0 || L <= 0 || T <= 0 || V < 0 || rounds <= 0 || rounds > 8 || round_stride < 0) return PS_EINVAL;
    if (B == 0) return PS_OK;
    if (!rowptr || !starts || !ids || !counts || !nvalid) return PS_EINVAL;
    if (!packed && (!col || !cdf)) return PS_EINVAL;           // the packed blocks hold the same values: col / cdf / guide may then be NULL
    if (rng_mode != PS_RNG_STREAM && rng_mode != PS_RNG_PHILOX && rng_mode != PS_RNG_STREAM_RAW && rng_mode != PS_RNG_STREAM_WALKS)
        return PS_EINVAL;
    if (rng_mode != PS_RNG_PHILOX && (!uniforms || !uoff)) return PS_EINVAL;
    if (rng_mode == PS_RNG_STREAM_WALKS && rounds != 1) return PS_EUNSUPPORTED;    // per-walk positions: one sample per launch
    if (!packed && (nodeinfo == nullptr) != (guide == nullptr)) return PS_EINVAL;
    if (packed && !nodeinfo) return PS_EINVAL;
    if (buckets && (!nodeinfo || reinterpret_cast<size_t>(buckets) % 64 != 0)) return PS_EINVAL;
    if (dest_info && (!packed || reinterpret_cast<size_t>(dest_info) % 8 != 0)) return PS_EINVAL;
    const int64_t P = (int64_t)W * L;
    if (P > 4096) return PS_EUNSUPPORTED;                  // 64 positions per lane: registers (vid / slot / cr) and 160 KiB of LDS end here
    int np = 1;
    while (np * 64 < P) np <<= 1;
    int hs_log2 = 6;
    while ((1 << hs_log2) * 4 < 5 * P) ++hs_log2;      // table >= 1.25 P slots (load factor <= 0.8)
    WalkArgs a{rowptr, col, cdf, V, starts, B, W, L, T, rng_mode, uniforms, uoff,
               (uint32_t)seed, (uint32_t)(seed >> 32), call, nodeinfo, guide, reinterpret_cast<const unsigned char *>(packed), ids, counts, nvalid, hs_log2, 0, 0, reinterpret_cast<const unsigned char *>(buckets), half_buckets, 0,
               rounds, round_stride, reinterpret_cast<const uint2 *>(dest_info), 0};
    // Few start nodes (a rank's shard of a multi-GPU job): one wave per (node, round) instead of one per node -- the launch is a
    // handful of generations of resident waves (256 CUs x 20), and waves of half the life waste half as much in the last one
    // (7 381 nodes x 2 layers: measured below); many nodes: all rounds of a node in one wave (the start row is staged once: 4 %).
    // PS_WALK_SPLIT=0 / 1 forces a form (tests run both).
    {
        const char *e = getenv("PS_WALK_SPLIT");
        a.split = rounds > 1 && (e ? atoi(e) != 0 : B * rounds <= SPLIT_MAX_WAVES);
    }
    const int wave_rounds = a.split ? 1 : rounds;
    // LDS budget: the kernel holds 24 waves per CU by registers; 160 KB / 24 leaves ~6.6 KB per wave, and whatever
    // the position buffers and the hash table do not need of that lets longer start rows be staged.
    const int hash_words = 3 * (1 << hs_log2);
    const int bitmap_words = P <= 1024 ? BITMAP_WORDS : (int)(((P >> 5) + 1 + 7) & ~7);
    a.bitmap_words = bitmap_words;
    // (8 KiB per wave -- what 20 resident waves leave -- stages 84 % of SYN-25M's start rows instead of 78 % and measured no faster,
    // r04: 364-397 us against 374-381 for the two-layer launch, alternating in one process; 9 KiB slower)
    int region_words = (6656 / 4) - wave_rounds * np * 64 - bitmap_words;
    if (region_words < hash_words) region_words = hash_words;
    region_words &= ~31;                                    // whole 128-byte blocks
    a.region_words = region_words;
    // a staged row = its 128-byte blocks (32 words per 8 edges) [+ 8 bytes per edge of destination records]
    if (np > 4) a.dest_info = nullptr;                      // (the DEST kernels exist for W * L <= 256, the reference's 100 x 2 among them)
    a.stage_blocks = packed ? region_words / (a.dest_info ? 48 : 32) : 0;
    const size_t lds = (size_t)WAVES_PER_BLOCK * (wave_rounds * np * 64 + region_words + bitmap_words) * sizeof(int32_t);
    if (lds > 160 * 1024) return PS_EUNSUPPORTED;            // (e.g. eight fused layers of 4096 positions each)
    int64_t grid = ps_cdiv(a.split ? B * rounds : B, WAVES_PER_BLOCK);
    if (grid > (int64_t)1 << 30) grid = (int64_t)1 << 30;
    hipStream_t st = ps_stream(stream);
    // dynamic LDS beyond 64 KiB (W * L > 1024 only) has to be allowed per kernel and device, once
#define PS_WS_LAUNCH(NP_)                                                                                                      \
    do {                                                                                                                       \
        if (lds > 64 * 1024) {                                                                                                 \
            static PsPerDevice done;                                                                                           \
            int dv = 0;                                                                                                        \
            if (hipGetDevice(&dv) != hipSuccess || dv < 0 || dv >= 64) return PS_ELAUNCH;                                      \
            if (!done.get(dv)) {                                                                                               \
                if (hipFuncSetAttribute(reinterpret_cast<const void *>(walk_sample_kernel<NP_, false>),                        \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||               \
                    hipFuncSetAttribute(reinterpret_cast<const void *>(walk_sample_kernel<NP_, true>),                         \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return PS_ELAUNCH; \
                done.set(dv, 1);                                                                                               \
            }                                                                                                                  \
        }                                                                                                                      \
        if (a.dest_info != nullptr && NP_ <= 4) {                                                                              \
            constexpr int ND = NP_ <= 4 ? NP_ : 4;      /* only these are instantiated with DEST */                                \
            if (rng_mode == PS_RNG_PHILOX)                                                                                     \
                hipLaunchKernelGGL((walk_sample_kernel<ND, false, true>), dim3((unsigned)grid), dim3(64 * WAVES_PER_BLOCK), lds, st, a); \
            else                                                                                                               \
                hipLaunchKernelGGL((walk_sample_kernel<ND, true, true>), dim3((unsigned)grid), dim3(64 * WAVES_PER_BLOCK), lds, st, a);  \
        } else if (rng_mode == PS_RNG_PHILOX)                                                                                  \
            hipLaunchKernelGGL((walk_sample_kernel<NP_, false>), dim3((unsigned)grid), dim3(64 * WAVES_PER_BLOCK), lds, st, a); \
        else                                                                                                                   \
            hipLaunchKernelGGL((walk_sample_kernel<NP_, true>), dim3((unsigned)grid), dim3(64 * WAVES_PER_BLOCK), lds, st, a);  \
    } while (0)
    switch (np) {
        case 1: PS_WS_LAUNCH(1); break;
        case 2: PS_WS_LAUNCH(2); break;
        case 4: PS_WS_LAUNCH(4); break;
        case 8: PS_WS_LAUNCH(8); break;
        case 16: PS_WS_LAUNCH(16); break;
        case 32: PS_WS_LAUNCH(32); break;
        case 64: PS_WS_LAUNCH(64); break;
        default: return PS_EUNSUPPORTED;
    }
#undef PS_WS_LAUNCH
    PS_CHECK_LAUNCH();
    return PS_OK;
}

extern "C" int ps_walk_sample(const int64_t *rowptr, const int32_t *col, const double *cdf, int64_t V,
                              const int64_t *starts, int64_t B, int W, int L, int T, int rng_mode,
                              const double *uniforms, const int64_t *uoff, uint64_t seed, uint32_t call,
                              const uint32_t *nodeinfo, const int32_t *guide, const void *packed, const void *buckets,
                              const void *dest_info, int32_t *ids, int32_t *counts, int32_t *nvalid, ps_stream_t stream) {
    return walk_sample_launch(rowptr, col, cdf, V, starts, B, W, L, T, rng_mode, uniforms, uoff, seed, call, nodeinfo, guide,
                              packed, buckets, dest_info, 1, 0, ids, counts, nvalid, stream);
}

extern "C" int ps_walk_sample_layers(const int64_t *rowptr, const int32_t *col, const double *cdf, int64_t V,
                                     const int64_t *starts, int64_t B, int W, int L, int T, int rng_mode,
                                     const double *uniforms, const int64_t *uoff, int64_t layer_stride, uint64_t seed,
                                     uint32_t call, const uint32_t *nodeinfo, const int32_t *guide, const void *packed,
                                     const void *buckets, const void *dest_info, int layers, int32_t *ids, int32_t *counts,
                                     int32_t *nvalid, ps_stream_t stream) {
    return walk_sample_launch(rowptr, col, cdf, V, starts, B, W, L, T, rng_mode, uniforms, uoff, seed, call, nodeinfo, guide,
                              packed, buckets, dest_info, layers, layer_stride, ids, counts, nvalid, stream);
}

extern "C" int ps_walk_paths(const int64_t *rowptr, const int32_t *col, const double *cdf, int64_t V,
                             const int64_t *starts, int64_t B, int L, int rng_mode, const double *uniforms,
                             const int64_t *uoff, uint64_t seed, uint32_t call, int walk_mod,
                             const uint32_t *nodeinfo, const int32_t *guide, int32_t *paths, ps_stream_t stream) {
    if (B < 0 || L <= 0) return PS_EINVAL;
    if (B == 0) return PS_OK;
    if (!rowptr || !col || !cdf || !starts || !paths) return PS_EINVAL;
    if (rng_mode != PS_RNG_STREAM && rng_mode != PS_RNG_PHILOX) return PS_EINVAL;
    if (rng_mode == PS_RNG_STREAM && (!uniforms || !uoff)) return PS_EINVAL;
    if ((nodeinfo == nullptr) != (guide == nullptr)) return PS_EINVAL;
    int64_t grid = ps_cdiv(B, 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(walk_paths_kernel, dim3((unsigned)grid), dim3(256), 0, ps_stream(stream), rowptr, col, cdf, V,
                       starts, B, L, rng_mode, uniforms, uoff, (uint32_t)seed, (uint32_t)(seed >> 32), call, walk_mod, nodeinfo, guide, paths);
    PS_CHECK_LAUNCH();
    return PS_OK;
}

extern "C" int ps_uniform_offsets(const int64_t *rowptr, int64_t V, const int64_t *starts, int64_t B, int W, int L,
                                  int64_t *uoff, int64_t *total, ps_stream_t stream) {
    if (!total || B < 0 || W <= 0 || L <= 0) return PS_EINVAL;
    if (B > 0 && (!rowptr || !starts || !uoff)) return PS_EINVAL;
    hipLaunchKernelGGL(uniform_offsets_kernel, dim3(1), dim3(1024), 0, ps_stream(stream), rowptr, V, starts, B,
                       (int64_t)W * L, uoff, total);
    PS_CHECK_LAUNCH();
    return PS_OK;
}

extern "C" int ps_graph_stats(const int64_t *rowptr, const int32_t *col, int64_t E, int64_t V, int64_t *flags,
                              ps_stream_t stream) {
    if (!rowptr || !flags || (E > 0 && !col)) return PS_EINVAL;
    hipStream_t st = ps_stream(stream);
    if (hipMemsetAsync(flags, 0, 2 * sizeof(int64_t), st) != hipSuccess) return PS_ELAUNCH;
    int64_t n = E > V ? E : V;
    int64_t grid = ps_cdiv(n > 0 ? n : 1, 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(graph_stats_kernel, dim3((unsigned)grid), dim3(256), 0, st, rowptr, col, E, V,
                       reinterpret_cast<unsigned long long *>(flags));
    PS_CHECK_LAUNCH();
    return PS_OK;
}
