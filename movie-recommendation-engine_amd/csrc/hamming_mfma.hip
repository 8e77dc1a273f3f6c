// hamming_mfma.hip -- the Hamming k-NN scan of faiss.IndexLSH.search (reference utils/nearest_neighbors.py:47-68)
// as an EXACT integer contraction on the gfx950 matrix cores.
//
// A code of nbits bits is expanded once into nbits values s_j = +1 (bit set) / -1 (bit clear).  Then
//     dot(q, x) = sum_j s_j(q) s_j(x) = nbits - 2 * hamming(q, x)        (an integer)
// so the all-pairs distance table is a GEMM.  The signs are stored as fp4 (e2m1: +1 = 0x2, -1 = 0xA, padding 0x0) and
// contracted with the block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 (both block scales 2^0): every product is +-1 or 0
// and every partial sum an integer below 2^24, so the f32 accumulator is exact (tools/ubench/mfma_fp4_probe.hip checks
// the instruction against an integer dot product and times it: 16.0 ns per 32x32x64 MFMA per SIMD = 8.4 POP/s of sign
// products on random data, 2.2 x the int8 form v_mfma_i32_32x32x32_i8 that the first version of this file used, at half
// the operand bytes).  Results are bit-identical to the popcount scan (csrc/hamming_topk.hip): the k smallest by
// (distance, id), ascending.
//
// Sign planes ("fragment order"): for a tile of 32 codes and a 64-bit K step s, one 1 KiB block; lane `lane`'s 16 bytes
//     planes[(tile * KS + s) * 1024 + lane * 16 ..] = the 32 nibbles s_{64 s + 32 (lane >> 5) + j}(code 32 tile + (lane & 31)),
//     j = 0..31, low nibble first
// i.e. exactly the 16 bytes lane `lane` feeds to the MFMA as its A (items) or B (queries) operand, so a tile is
// brought into LDS by global_load_lds_dwordx4 (no registers, lane-linear image, conflict-free ds_read_b128) and a
// query tile is 16 coalesced dwordx4 loads into registers.  A and B use the same (lane half, byte) -> k map, so the
// contraction is over matching bits whatever k permutation the hardware applies inside a step.
//
// Three passes (all exact; no host synchronisation):
//   bound   : over the first fifth of the table every lane keeps the KM best "maximum dot of a 16-item group" values; the k-th
//             smallest group minimum over a query's lanes (bound_select_kernel) bounds its k-th best distance from above (k distinct
//             groups = k distinct items).
//   collect : over ALL items, a 32 x 32 tile of dots is KS MFMAs; lane = query, registers = items, so the admission test is ONE
//             per-lane threshold.  The accumulators start at BIAS + r / 16, so an element carries its row in its fraction bits and
//             the maximum of a hit lane is appended to the lane's candidate column without looking at the other 15; a second hit
//             in the same lane and tile is detected exactly and only then are all rows walked.  A column that would overflow is
//             compacted and the lane's threshold tightened (exact: ids ascend during the sweep).
//   merge   : every slice leaves one sorted k-list per query (the two lanes of a query merge theirs in LDS);
//             slice_merge_kernel merges them by (distance, row): 16 lanes per query, k rounds of a DPP row minimum
//             (slice_merge64_kernel, one wave per query, when few queries over a large table are cut into 17..64 slices).
// Queries arrive as sign planes (ps_hamming_topk_mfma) or as packed codes (ps_hamming_topk_mfma_codes: the pipelined kernels'
// workgroups expand their own 32 queries in registers, query_frag; for the older kernels the launcher expands into the workspace).
// Two generations of sweep kernels live here:
//   hamming_pipe_kernel<KS, MODE>   (r04; 256- and 512-bit codes, k <= 12: the BASELINE configurations) -- two workgroups per CU,
//       a ring of single tiles, two accumulator sets per wave: the detection of tile t - 1 runs in the gaps of the MFMA chain of
//       tile t.  10 000 x 59 047, k = 11: 0.128 ms at 256 bit, 0.168 ms at 512 bit (r03: 0.194 / 0.207 on the same box).
//   hamming_mfma_kernel<KS, MODE, KM, DB>  (r02 / r03; every other served shape: 64- / 128-bit codes, k up to 32) -- 8 waves, a
//       ring of two- / four-tile entries, waves 4-7 one entry late.
// What the r04 measurements said about this pass (tools/hm_probe_run.sh knock-outs, tools/hm_pmc.sh counters, tools/hm_times.py
// per-workgroup stamps), because three plausible theories were wrong first: it is not the barrier (without it: slower), not the
// LDS round trip of the fragments (requested a tile ahead: no gain), not instruction-cache pressure (a 70 KB unrolled loop and a
// 20 KB loop ran alike).  A SIMD issues about one instruction per four cycles whatever its kind while four waves run the same code in
// near lockstep, so the sweep costs its instruction count (~55 per tile and wave with the usual exit, ~70 with a hit) -- and the
// kernel's time was set by the ~60 of 480 workgroups in which some lane's column overflowed: the serial in-LDS compaction took
// ~30 K cycles with the whole workgroup at its barrier, because room for a whole tile (16 rows per lane) was demanded after
// every tile.  Asking only for the room the next append needs, and compacting by bisection when a column really is full, took the
// 256-bit collect pass from 153 to ~100 us (random codes: 94; the straight-line pipeline and the leaner keys are the rest of it).
#include "ps_common.h"
#include <type_traits>

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

// experiments only (tools/hm_probe.sh builds variants with -DPS_HM_DEBUG=bits; results are WRONG with any bit set):
// 1 no tile epilogue, 2 no LDS-DMA, 4 no barrier, 8 no MFMA, 16 no fragment reads
#ifndef PS_HM_DEBUG
#define PS_HM_DEBUG 0
#endif
#if PS_HM_DEBUG & 32      // event counters of the collect pass (tools/hm_counts.py): tiles, slow-path entries, group entries, row ballots, appends, compactions
__device__ unsigned long long ps_hm_counts[8];
#define PS_HM_COUNT(slot, n) do { if (MODE == 1 && lane == 0) atomicAdd(&ps_hm_counts[slot], (unsigned long long)(n)); } while (0)
extern "C" int ps_debug_hm_counts(unsigned long long *host, int reset) {
    unsigned long long z[8] = {};
    if (reset) return (int)hipMemcpyToSymbol(HIP_SYMBOL(ps_hm_counts), z, sizeof(z));
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ps_hm_counts), sizeof(z));
}
#else
#define PS_HM_COUNT(slot, n) do {} while (0)
#endif
#if PS_HM_DEBUG & 512     // per-workgroup start / end stamps of the pipelined collect pass (tools/hm_times.py): load balance
__device__ unsigned long long ps_hm_times[4 * 4096];
extern "C" int ps_debug_hm_times_clear() {
    static unsigned long long z[4 * 4096];
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(ps_hm_times), z, sizeof(z));
}
extern "C" int ps_debug_hm_times(unsigned long long *host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ps_hm_times), sizeof(unsigned long long) * 4 * 4096);
}
#endif
constexpr uint32_t EMPTY_KEY = 0xffffffffu;
constexpr int NBUF = 4;                    // ring entries: one being multiplied (from registers), one being read into registers, two in flight
constexpr int WAVES = 8;
constexpr int PAD_TILES = 4;               // plane tables are padded to whole ring entries (2 or 4 item tiles of 32 codes)
constexpr float NO_DOT = -1048576.0f;       // "no item": below every real dot (|dot| <= 1024)
// item tiles per ring entry = per barrier: 16 MFMAs per wave and barrier for 512- and 256-bit codes
template <int KS> struct EntryTiles { static constexpr int value = KS >= 4 ? 2 : 4; };

// ---- sign planes ------------------------------------------------------------------------------------------------
// 32 code bits -> 32 fp4 nibbles: bit b -> +1 (0x2) / -1 (0xA) = 0xA ^ (b << 3); four bits are spread to the nibble positions
// 0, 4, 8, 12 by OR-ing shifted copies (a multiply would carry between the overlapping copies)
__device__ __forceinline__ uint4 expand_word(uint32_t bits) {
    auto four = [](uint32_t n) { return (n | (n << 3) | (n << 6) | (n << 9)) & 0x1111u; };
    auto eight = [&](uint32_t byte) { return 0xaaaaaaaau ^ ((four(byte & 15u) | (four(byte >> 4) << 16)) << 3); };
    return make_uint4(eight(bits & 255u), eight((bits >> 8) & 255u), eight((bits >> 16) & 255u), eight(bits >> 24));
}

// one thread = one 16-byte piece (tile, step, lane): the 32 code bits of word 2 s + (lane >> 5) as 32 fp4 nibbles
__global__ __launch_bounds__(256) void lsh_expand_kernel(const uint32_t *__restrict__ codes, int64_t n, int KS,
                                                         int64_t pieces, uint4 *__restrict__ planes) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < pieces; p += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(p & 63);
        const int64_t ts = p >> 6;
        const int s = (int)(ts % KS);
        const int64_t tile = ts / KS;
        const int64_t row = tile * 32 + (lane & 31);
        uint4 o = make_uint4(0u, 0u, 0u, 0u);
        if (row < n) o = expand_word(codes[row * (2 * KS) + 2 * s + (lane >> 5)]);
        planes[p] = o;
    }
}

struct HArgs {
    const unsigned char *qplanes;   // the queries' sign planes, or
    const uint32_t *qcodes;         // (qplanes == nullptr) their packed codes: a workgroup expands its own 32 queries (query_frag)
    const unsigned char *dbplanes;
    int64_t nq, N;
    int64_t tile_begin, tile_end;   // tiles swept by this launch
    int64_t tiles_per_slice;
    int nqb, slices;
    int k, nbits, shift;
    int cap;                        // collect: slots of a lane's candidate column (k + 16 <= cap)
    int64_t id_offset;
    const int32_t *thr0;            // collect: per-query admission bound (hamming distance, inclusive)
    int32_t *bl;                    // bound: [nq][slices * 2][KM] group-minimum distances, ascending
    int list_base;                  // collect: first output list of this launch (lists of earlier launches precede)
    int32_t *out_d;                 // collect: [slices][nq][k] distances / table rows (-1 = none), ascending
    int32_t *out_r;
};

// The 16 bytes lane `lane` feeds to the MFMA for K step s of query tile qtile: read from the plane table, or -- packed query codes
// -- built here (~40 integer instructions per step, once per workgroup, against a sweep of hundreds of tiles; saves the
// ps_lsh_expand launch in front of every search)
template <int KS>
__device__ __forceinline__ v4i query_frag(const HArgs &a, int64_t qtile, int s, int lane) {
    if (a.qplanes != nullptr) return *reinterpret_cast<const v4i *>(a.qplanes + ((qtile * KS + s) * 64 + lane) * 16);
    const int64_t q = qtile * 32 + (lane & 31);
    uint4 o = make_uint4(0u, 0u, 0u, 0u);                   // rows past the end: zero nibbles, as in a padded plane table
    if (q < a.nq) o = expand_word(a.qcodes[q * (2 * KS) + 2 * s + (lane >> 5)]);
    return v4i{(int)o.x, (int)o.y, (int)o.z, (int)o.w};
}

// LDS-DMA issued from inline asm: hipcc (ROCm 7.2) cannot prove that a ds_read of ring buffer i does not alias the
// global_load_lds into buffer i+2 and drains vmcnt(0) before every tile's first fragment read when the builtin is
// used; the asm form is invisible to that pass, and the ring is ordered by hand (counted vmcnt + s_barrier below).
// No compiler-visible vector-memory operation may be outstanding while these are in flight (its own counted waits
// would miscount): the sweep is bracketed by explicit vmcnt(0) waits.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void lds_dma16(const void *gptr, uint32_t lds_byte_offset) {
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(lds_byte_offset) : "memory", "m0");
}
#pragma clang diagnostic pop

// one K step: 32 x 32 x 64 signs; only the first four registers of the 8-register operands are read for fp4
__device__ __forceinline__ v16f sign_mfma(const v4i &a, const v4i &b, const v16f &c) {
    const v8i A = {a[0], a[1], a[2], a[3], 0, 0, 0, 0};
    const v8i B = {b[0], b[1], b[2], b[3], 0, 0, 0, 0};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}

// The first K step of a tile in the collect pass: D = A x B + C with C = the row-encoding constants in registers of their own.
// Written in asm because the compiler selects the form with the accumulator tied to C and copies the 16 constants into the
// accumulator registers in front of every tile (eight v_mov_b64 per tile beside ~50 epilogue instructions).  The destination is
// early-clobber (a 32 x 32 MFMA writes while it still reads); the operands come from ds_read_b128 / kernel-lifetime registers,
// whose waits the compiler inserts for asm operands as for any use; the next MFMA reads this result as its C with exactly the
// same registers, which the matrix pipe orders itself.
__device__ __forceinline__ v16f sign_mfma_first(const v4i &a, const v4i &b, const v16f &c, int scale) {
    v16f d;
    asm("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %3, %4, %4 op_sel_hi:[0,0,0] cbsz:4 blgp:4"
        : "=&v"(d) : "v"(a), "v"(b), "v"(c), "v"(scale));
    return d;
}

// MODE 0 = bound pass, 1 = collect pass.  One wave = one tile of 32 queries (16 query-fragment registers per 32 bits
// of code).  Per-tile work besides the MFMAs is ~11 vector instructions (this file is compiled with -fno-honor-nans:
// every value is an integer held in f32, so the maxima are bare v_max3_f32 without sNaN-quieting copies): the maximum of a
// lane's 16 dots as a two-level tree and one compare; everything else happens only when some lane of the wave has a hit.
// DB = true : one workgroup per CU (two waves per SIMD), two register sets of item fragments (the LDS reads of entry i + 1 under
//             the MFMAs of entry i), 4-deep ring.  Any k <= 32.
// DB = false: TWO workgroups per CU (four waves per SIMD, <= 128 VGPRs: one fragment set, smaller entries, 3-deep ring, candidate
//             columns of k + 16 slots so that both fit the LDS): a tile's epilogue costs a wave ~15 cycles per instruction
//             whatever it is (two thirds of the r02 collect pass), and only more resident waves hide that.  k <= 12.
template <int KS, int MODE, int KM, bool DB>
__global__ __launch_bounds__(512, DB ? 2 : 4) void hamming_mfma_kernel(HArgs a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    constexpr int IT = DB ? EntryTiles<KS>::value : (KS >= 8 ? 1 : KS == 4 ? 2 : 4);
    constexpr int NB = DB ? NBUF : 3;
    constexpr int TILE_BYTES = KS * 1024;
    constexpr int ENTRY_BYTES = IT * TILE_BYTES;           // one ring entry = IT consecutive item tiles
    constexpr int PIECES = IT * KS;                        // 1 KiB LDS-DMA pieces per entry
    constexpr int PPW = (PIECES + WAVES - 1) / WAVES;      // pieces per wave and entry (a wave issues PPW or none)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const bool late = wv >= 4;                              // SIMD partners are waves w and w + 4

    // XCD-aware block order: blocks b, b + 8, ... share an XCD (and its L2); give them consecutive logical ids so that
    // an XCD sweeps one or two table slices, not all of them (bijective remap; speed only)
    const int G = gridDim.x, b = blockIdx.x;
    const int gq = G >> 3, gr = G & 7, xcd = b & 7;
    const int logical = (xcd < gr ? xcd * (gq + 1) : gr * (gq + 1) + (xcd - gr) * gq) + (b >> 3);
    const int slice = logical / a.nqb, qb = logical - slice * a.nqb;
    const int64_t qtile = (int64_t)qb * WAVES + wv;
    const int64_t nqtiles = (a.nq + 31) >> 5;
    const int64_t q = qtile * 32 + li;
    const bool q_ok = q < a.nq;

    int64_t t0 = a.tile_begin + (int64_t)slice * a.tiles_per_slice;
    int64_t t1 = t0 + a.tiles_per_slice;
    if (t1 > a.tile_end) t1 = a.tile_end;
    const int nt = t1 > t0 ? (int)((t1 - t0 + IT - 1) / IT) : 0;     // ring entries; t0 and tiles_per_slice are multiples of IT
    const int64_t last_tile = (a.N - 1) >> 5;               // tiles from here on hold padding rows (zero nibbles)
    // the same bound relative to this slice, as a 32-bit scalar: a 64-bit signed compare is a VECTOR instruction on this ISA
    // (v_cmp_lt_i64 + a move), two of the ~12 instructions every tile's epilogue pays
    const int64_t last_rel64 = last_tile - t0;
    const int last_rel = last_rel64 > 0x7fffffff ? 0x7fffffff : last_rel64 < 0 ? 0 : (int)last_rel64;

    // query fragments (B operand): 16 bytes per K step, resident for the whole sweep
    v4i bq[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        bq[s] = v4i{0, 0, 0, 0};
        if (qtile < nqtiles)                                 // (packed query codes: the launcher expands them first, launch_passes)
            bq[s] = *reinterpret_cast<const v4i *>(a.qplanes + ((qtile * KS + s) * 64 + lane) * 16);
    }

    const uint32_t lds_base = (uint32_t)reinterpret_cast<uintptr_t>(smem);       // low half of the flat address = LDS offset
    // entry e of this slice = tiles t0 + IT e .. + IT - 1 = PIECES consecutive 1 KiB pieces of the plane table
    auto prefetch = [&](int e) __attribute__((always_inline)) {
        const int buf = DB ? (e & (NBUF - 1)) : e % NB;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int p = wv + WAVES * i;
            if (p < PIECES && !(PS_HM_DEBUG & 2))
                lds_dma16(a.dbplanes + (((t0 + (int64_t)e * IT) * KS + p) * 64 + lane) * 16, lds_base + (uint32_t)(buf * PIECES + p) * 1024u);
        }
    };

    // ---- per-lane state ----
    const int CAP = a.cap;
    uint32_t *cand = reinterpret_cast<uint32_t *>(smem + NB * ENTRY_BYTES) + wv * (CAP * 64);  // [slot][lane]
    int cnt = 0;
    // Collect pass: the accumulators start at BIAS + r / 16 instead of 0 (r = register = row of the lane's 16), so an element
    // is v = BIAS + dot + r / 16: exact in f32 (|dot| <= 512, 4 fraction bits, v in [2560, 3585) = one binade), ordered by
    // (dot, r), and the maximum of a lane's 16 elements names its row: bits 12.. of the mantissa hold 1024 + dot, bits 8..11
    // r.  dot >= t  <=>  v >= BIAS + t for integers, so thresholds live in the same domain.
    constexpr float BIAS = 3072.0f;
    float thr = 3.0e38f;                                    // admit iff v >= thr
    float best[KM];                                         // bound pass: KM largest group maxima of the dot, descending
    v16f cinit;
#pragma unroll
    for (int r = 0; r < 16; ++r) cinit[r] = MODE == 1 ? BIAS + (float)r * 0.0625f : 0.f;
    if (MODE == 1) {
        if (q_ok && !(PS_HM_DEBUG & 64)) thr = BIAS + (float)(a.nbits - 2 * a.thr0[q]);     // 64: nothing passes (the usual exit only)
    } else {
#pragma unroll
        for (int j = 0; j < KM; ++j) best[j] = NO_DOT;
    }

    // a lane's column -> its k best keys in slots 0..k-1 (ascending), cnt = min(cnt, k); threshold tightened when the
    // column holds k keys: a later item (larger id) that only ties the k-th distance can never displace it
    auto compact = [&]() __attribute__((always_inline)) {
        const int k = a.k;
        for (int p = 0; p < k; ++p) {
            uint32_t bestk = (p < cnt) ? cand[p * 64 + lane] : EMPTY_KEY;
            const uint32_t head = bestk;
            int bj = p;
            for (int j = p + 1; j < CAP; ++j) {
                const uint32_t v = (j < cnt) ? cand[j * 64 + lane] : EMPTY_KEY;
                if (v < bestk) { bestk = v; bj = j; }
            }
            if (p < cnt) {
                cand[bj * 64 + lane] = head;
                cand[p * 64 + lane] = bestk;
            }
        }
        cnt = cnt < k ? cnt : k;
        if (cnt == k) {
            const int hk = (int)(cand[(k - 1) * 64 + lane] >> a.shift);
            const float nthr = BIAS + (float)(a.nbits - 2 * hk + 2);
            thr = nthr > thr ? nthr : thr;
        }
    };

    // acc is read in place (no copies): the usual exit is 8 + 3 maxima, one compare, one scalar branch
    const int ham_c = a.nbits + 1024;
    auto append_bits = [&](uint32_t bits, uint32_t row_base) __attribute__((always_inline)) {            // one element -> key in the lane's column
        const uint32_t ham = (uint32_t)(ham_c - (int)((bits >> 12) & 0x7ffu)) >> 1;
        const uint32_t r = (bits >> 8) & 15u;
        cand[cnt * 64 + lane] = (ham << a.shift) | (row_base + (r & 3u) + 8u * (r >> 2));
        ++cnt;
    };
    float dbg_sink = 0.f;
    auto epilogue = [&](v16f &acc, int i) __attribute__((always_inline)) {
        if (PS_HM_DEBUG & 2048) {                           // experiment: the accumulators are read (the 8 maxima), no compare, no branch:
            // what the matrix work costs.  collect 115 us (whole epilogue 159, usual exit only 125).  A build whose epilogue does NOT
            // read them (bit 1, or ten unrelated vector instructions) lets the compiler drop every MFMA whose result is overwritten
            // unread: its 58 / 67 us are the ring and the barriers, not a matrix-pipe floor (r03's first reading of bit 1 was wrong)
            const float a0 = fmaxf(fmaxf(acc[0], acc[1]), acc[2]), a1 = fmaxf(fmaxf(acc[3], acc[4]), acc[5]);
            const float a2 = fmaxf(fmaxf(acc[6], acc[7]), acc[8]), a3 = fmaxf(fmaxf(acc[9], acc[10]), acc[11]);
            const float a4 = fmaxf(fmaxf(acc[12], acc[13]), acc[14]), a5 = acc[15];
            dbg_sink = fmaxf(dbg_sink, fmaxf(fmaxf(fmaxf(a0, a1), a2), fmaxf(fmaxf(a3, a4), a5)));
            return;
        }
        const int64_t t = t0 + i;
        if (__builtin_expect(i >= last_rel, 0)) {           // wave-uniform, once per table: mask the padding rows of its end
            const int64_t left64 = a.N - t * 32;            // valid rows of this tile (<= 0: a padding tile of the last entry)
            const int left = (int)(left64 < 0 ? 0 : left64 > 32 ? 32 : left64) - 4 * lh;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if ((r & 3) + 8 * (r >> 2) >= left) acc[r] = MODE == 1 ? 0.f : NO_DOT;
        }
        // maximum of the lane's 16 elements as a TERNARY tree (v_max3_f32): triples {0,1,2} .. {12,13,14} and {15}, then two
        // triples of those: 5 + 2 + 1 = 8 instructions (r03 first version: four groups of four, 8 + 3: collect 168 -> 159 us;
        // knock-outs: matrix work + these 8 maxima alone 115 us, with this exit only 125, everything 159)
        const float a0 = fmaxf(fmaxf(acc[0], acc[1]), acc[2]), a1 = fmaxf(fmaxf(acc[3], acc[4]), acc[5]);
        const float a2 = fmaxf(fmaxf(acc[6], acc[7]), acc[8]), a3 = fmaxf(fmaxf(acc[9], acc[10]), acc[11]);
        const float a4 = fmaxf(fmaxf(acc[12], acc[13]), acc[14]), a5 = acc[15];
        const float mu = fmaxf(fmaxf(a0, a1), a2), mv = fmaxf(fmaxf(a3, a4), a5);
        const float m = fmaxf(mu, mv);
        if (MODE == 0) {
            // a lane's list changes only when its new group maximum beats the list's last entry: after the first few
            // tiles that is rare, and the insertion network (2 KM instructions) runs for the whole wave only then
            if (__ballot(m > best[KM - 1]) == 0ull) return;
            float x = m;
#pragma unroll
            for (int j = 0; j < KM; ++j) {
                const float hi = fmaxf(best[j], x);
                x = fminf(best[j], x);
                best[j] = hi;
            }
            return;
        }
        PS_HM_COUNT(0, 1);
        if (__ballot(m >= thr) == 0ull) return;             // the usual exit (~1/3 of the tiles at 10 000 x 59 047 x 512 bit)
        PS_HM_COUNT(1, 1);
        // Some lane holds an element >= thr: about one per 1 000 elements, i.e. usually ONE row of ONE lane of the wave's
        // 64 x 16.  Every scalar branch on a vector compare costs this wave a round trip that its SIMD partner's MFMA stream
        // does not hide (the r02 version walked groups and rows with ~12 such branches: ~900 cycles per entry, 2/3 of the
        // collect pass), so the common case is straight-line: the maximum itself names its row (fraction bits) and is
        // appended; that is complete unless a second element of the lane passes too.  Two distinct rows differ in r / 3 or in
        // r % 3, so a second hit exists iff the second-largest maximum over the triples {3 j ..} (sg: the smaller of the two
        // upper maxima or the median of a triple of triples) or over the residue classes {r % 3 = j} (sh: the median of their
        // three maxima) passes: 14 more vector instructions, one branch, and only then the walk over all rows.
        const float sg = fmaxf(fmaxf(fminf(mu, mv), __builtin_amdgcn_fmed3f(a0, a1, a2)), __builtin_amdgcn_fmed3f(a3, a4, a5));
        const float h0 = fmaxf(fmaxf(fmaxf(acc[0], acc[3]), acc[6]), fmaxf(fmaxf(acc[9], acc[12]), acc[15]));
        const float h1 = fmaxf(fmaxf(fmaxf(fmaxf(acc[1], acc[4]), acc[7]), acc[10]), acc[13]);
        const float h2 = fmaxf(fmaxf(fmaxf(fmaxf(acc[2], acc[5]), acc[8]), acc[11]), acc[14]);
        const float sh = __builtin_amdgcn_fmed3f(h0, h1, h2);
        const uint32_t base = (uint32_t)i * 32u + 4u * lh;
        // Room in the columns (r04, see hamming_pipe_kernel): the usual tile appends ONE key per lane, so after a tile only room for
        // one more is demanded; a walk counts what it is about to append and compacts first only if a column would overflow.
        // (Demanding room for 16 rows after every tile made ~60 of a launch's 480 workgroups compact -- a serial selection in LDS,
        // ~30 K cycles with the workgroup at its barrier -- and those workgroups set the kernel's time.)
        if (__ballot(fmaxf(sg, sh) >= thr) == 0ull) {
            if (m >= thr) append_bits(__float_as_uint(m), base);
        } else {
            PS_HM_COUNT(2, 1);
            int nl = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) nl += acc[r] >= thr ? 1 : 0;
            if (__ballot(cnt + nl > CAP) != 0ull) { PS_HM_COUNT(5, 1); compact(); }      // (k + 16 <= CAP: it fits afterwards)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (acc[r] >= thr) append_bits(__float_as_uint(acc[r]), base);
        }
        if (__ballot(cnt > CAP - 1) != 0ull) { PS_HM_COUNT(5, 1); compact(); }   // room for the next tile's single append
    };

    // ---- sweep ----
    // Ring of NBUF entries in LDS (filled by LDS-DMA, every wave 1/8 of an entry), two register sets of item fragments:
    // while the MFMAs of entry i run from one set, the ds_read_b128s of entry i + 1 fill the other, so the LDS round trip
    // (all 8 waves read the whole entry: 128 KiB per entry and CU = half of the LDS bandwidth over an entry's MFMA time)
    // is under the matrix work instead of in front of it (r02: reads, then MFMAs, every entry: the sweep without any
    // epilogue took 2 200 cycles per entry against 1 024 of MFMA issue per SIMD).  Waves 4-7 run the epilogue of entry
    // i - 1 BEFORE the MFMAs of entry i (on the same accumulator registers, no copy), waves 0-3 after them, so the two
    // waves of a SIMD alternate between the matrix pipe and the VALU instead of meeting at both.
    // History on MI355X (10 000 x 59 047 x 512 bit, k = 11): int8 signs, one tile per barrier: collect 338 us + bound 58 us;
    // fp4 signs: 281 + 45; two tiles per barrier: 262 + 45 (r02, 0.32 ms for the call; popcount kernel 0.71 ms).
    // Measured alternatives that were NOT faster (int8 version): 2 query tiles per wave (250 VGPRs, candidate columns in
    // global memory): 561 us; SIMD partners half a tile apart: 381 / 399 us; a shared per-query histogram of admitted
    // candidates that tightens thresholds during the sweep: same total (its memory operations sit in the ring's waits).
    // Query fragments / bounds are in: nothing of the compiler's is in flight from here on.  The builtin form, so that
    // hipcc's own wait-count pass knows it (an asm wait is invisible to it and it would wait vmcnt(0) again at the
    // first use of a query fragment INSIDE the loop, i.e. drain the ring on every tile)
    __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0)
#pragma unroll
    for (int j = 0; j < NB - 1; ++j)
        if (j < nt) prefetch(j);
    v4i avA[IT][KS], avB[DB ? IT : 1][DB ? KS : 1];
    v16f acc[IT];
#pragma unroll
    for (int u = 0; u < IT; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = NO_DOT;

    auto read_entry = [&](int e, v4i (&av)[IT][KS]) __attribute__((always_inline)) {
        const unsigned char *tb = smem + (DB ? (e & (NBUF - 1)) : e % NB) * ENTRY_BYTES + lane * 16;
#pragma unroll
        for (int u = 0; u < IT; ++u)
#pragma unroll
            for (int s = 0; s < KS; ++s)
                av[u][s] = (PS_HM_DEBUG & 16) ? v4i{e, s, lane, u} : *reinterpret_cast<const v4i *>(tb + (u * KS + s) * 1024);
    };
    // own pieces of entry e have landed when at most `younger` younger entries' pieces are outstanding
    auto wait_entry = [&](int e) __attribute__((always_inline)) {
        const int younger = nt - 1 - e;
        if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW * 2) : "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    auto step = [&](int i, v4i (&cur)[IT][KS], v4i (&nxt)[IT][KS]) __attribute__((always_inline)) {
        // entry i + 1: own pieces landed; after the barrier everybody's have, and everybody has finished the MFMAs of entry
        // i - 1, i.e. has entry i in registers and is done with the buffer of entry i - 1, which the next LDS-DMA overwrites
        if (i + 1 < nt) wait_entry(i + 1);
        if (!(PS_HM_DEBUG & 4)) __builtin_amdgcn_s_barrier();
        if (i + NBUF - 1 < nt) prefetch(i + NBUF - 1);
        if (i + 1 < nt) read_entry(i + 1, nxt);
        __builtin_amdgcn_sched_barrier(0);
        if (late && i > 0 && !(PS_HM_DEBUG & 1)) {
            if (PS_HM_DEBUG & 128) __builtin_amdgcn_s_setprio(3);
#pragma unroll
            for (int u = 0; u < IT; ++u) epilogue(acc[u], (i - 1) * IT + u);
            if (PS_HM_DEBUG & 128) __builtin_amdgcn_s_setprio(0);
        }
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int u = 0; u < IT; ++u) {
                const v16f c = s == 0 ? cinit : acc[u];
                if (PS_HM_DEBUG & 8) acc[u][s & 15] += (float)(cur[u][s][0] ^ bq[s][1]);
                else acc[u] = sign_mfma(cur[u][s], bq[s], c);
            }
        if (!late && !(PS_HM_DEBUG & 1)) {
            if (PS_HM_DEBUG & 128) __builtin_amdgcn_s_setprio(3);
#pragma unroll
            for (int u = 0; u < IT; ++u) epilogue(acc[u], i * IT + u);
            if (PS_HM_DEBUG & 128) __builtin_amdgcn_s_setprio(0);
        }
    };
    if constexpr (DB) {
        if (nt > 0) {
            wait_entry(0);
            if (!(PS_HM_DEBUG & 4)) __builtin_amdgcn_s_barrier();
            read_entry(0, avA);
        }
        for (int i = 0; i < nt; i += 2) {
            step(i, avA, avB);
            if (i + 1 < nt) step(i + 1, avB, avA);
        }
    } else {
        // one fragment set: the reads of entry i are waited for in front of its MFMAs; the other three waves of the SIMD (two of
        // them of the CU's second workgroup, which shares no barrier with this one) fill the gap
        for (int i = 0; i < nt; ++i) {
            // entry i: own pieces landed (entry i + 1 may be in flight); after the barrier everybody's have, and everybody is
            // done with the MFMAs (= the fragment reads) of entry i - 1, whose buffer the next LDS-DMA overwrites
            if (nt - 1 - i >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!(PS_HM_DEBUG & 4)) __builtin_amdgcn_s_barrier();
            if (i + NB - 1 < nt) prefetch(i + NB - 1);
            read_entry(i, avA);
            __builtin_amdgcn_sched_barrier(0);
            if (late && i > 0 && !(PS_HM_DEBUG & 1)) {
#pragma unroll
                for (int u = 0; u < IT; ++u) epilogue(acc[u], (i - 1) * IT + u);
            }
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int u = 0; u < IT; ++u) {
                    if (PS_HM_DEBUG & 8) acc[u][s & 15] += (float)(avA[u][s][0] ^ bq[s][1]);
                    else if (s == 0 && MODE == 1 && !(PS_HM_DEBUG & 256)) acc[u] = sign_mfma_first(avA[u][s], bq[s], cinit, 0x7f7f7f7f);
                    else acc[u] = sign_mfma(avA[u][s], bq[s], s == 0 ? cinit : acc[u]);
                }
            if (!late && !(PS_HM_DEBUG & 1)) {
#pragma unroll
                for (int u = 0; u < IT; ++u) epilogue(acc[u], i * IT + u);
            }
        }
    }
    if (late && nt > 0 && !(PS_HM_DEBUG & 1)) {
#pragma unroll
        for (int u = 0; u < IT; ++u) epilogue(acc[u], (nt - 1) * IT + u);
    }
    if (PS_HM_DEBUG & 1) { if (acc[0][0] + acc[IT - 1][5] == 12345678.f) cnt = 1; }
    if (PS_HM_DEBUG & 2048) { if (dbg_sink == 12345678.f) cnt = 1; }

    // ---- results ----
    if (MODE == 0) {
        if (q_ok) {
            int32_t *dst = a.bl + ((int64_t)q * (a.slices * 2) + slice * 2 + lh) * KM;
#pragma unroll
            for (int j = 0; j < KM; ++j) dst[j] = best[j] == NO_DOT ? 0x7fffffff : (a.nbits - (int)best[j]) >> 1;
        }
        return;
    }
    // the two lanes of a query (rows 4 lh + ... of every tile) merge their sorted columns: one k-list per (slice, query)
    compact();
    ps_wave_lds_sync();
    const int cnt_hi = __shfl(cnt, li + 32, 64);
    if (q_ok && lh == 0) {
        const uint32_t idmask = (1u << a.shift) - 1u;
        const int64_t o = ((int64_t)(a.list_base + slice) * a.nq + q) * a.k;
        int ia = 0, ib = 0;
        for (int p = 0; p < a.k; ++p) {
            const uint32_t ka = ia < cnt ? cand[ia * 64 + lane] : EMPTY_KEY;
            const uint32_t kb = ib < cnt_hi ? cand[ib * 64 + lane + 32] : EMPTY_KEY;
            const uint32_t key = ka < kb ? ka : kb;
            if (ka < kb) ++ia; else ++ib;
            const bool has = key != EMPTY_KEY;
            a.out_d[o + p] = has ? (int32_t)(key >> a.shift) : 0x7fffffff;
            a.out_r[o + p] = has ? (int32_t)((key & idmask) + (uint32_t)(t0 * 32)) : -1;
        }
    }
}

// ---- r04: the same two passes as ONE software pipeline per wave -----------------------------------------------------------------
// hamming_mfma_kernel<KS, *, 4, false> above leaves the overlap of a tile's epilogue with matrix work to the OTHER waves of the SIMD
// (waves 4-7 run theirs one entry late).  Knock-outs at 256 bit (10 000 x 59 047, k = 11; tools/hm_probe_run.sh) showed what that
// costs once the matrix work no longer dominates: matrix work + the eight maxima 65 us, + compare / branch of the usual exit 83 us,
// everything 153 us -- and neither the barrier nor the LDS-DMA is in it (without them: 164 / 80 us).  Every tile's maxima wait for
// the tile's last MFMA, every branch on them for the maxima, and four waves per SIMD do not hide a chain that each of them has.
// Here a wave keeps TWO accumulator sets: the detection of tile t - 1 (eight v_max3 and the compare, straight-line, results in
// scalar registers long before they are branched on) is interleaved instruction by instruction with the dependent MFMA chain of
// tile t, whose gaps it fills; only a tile with a hit leaves the straight line, after the chain of the next tile has been issued.
// All waves run the same program (no late half), the padding rows of the table's end are handled by a separate tail loop (the
// in-place masking made the compiler copy all 16 accumulators in front of every epilogue), and the per-tile code is a quarter of
// the old loop body.  Same arithmetic, same candidate columns, same results bit for bit.
// code sizes the pipelined kernels serve, and their ring depth in tiles (24 KiB beside the 56 KiB of candidate columns: two
// workgroups per CU)
template <int KS> struct PipeServed { static constexpr bool value = KS == 4 || KS == 8; };
template <int KS> struct PipeRing { static constexpr int value = KS <= 4 ? 24 / KS : 3; };

template <int KS, int MODE>
__global__ __launch_bounds__(512, 4) void hamming_pipe_kernel(HArgs a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    constexpr int KM = 4;
    constexpr int NB = PipeRing<KS>::value;                 // ring of single tiles
    constexpr int PD = NB - 1;                              // tiles requested ahead
    constexpr int TILE_BYTES = KS * 1024;
    static_assert(KS <= WAVES, "one LDS-DMA piece per wave and tile");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
#if PS_HM_DEBUG & 512
    const unsigned long long ps_t_start = __builtin_readcyclecounter();
#endif

    const int G = gridDim.x, b = blockIdx.x;                // XCD-aware block order (see hamming_mfma_kernel)
    const int gq = G >> 3, gr = G & 7, xcd = b & 7;
    const int logical = (xcd < gr ? xcd * (gq + 1) : gr * (gq + 1) + (xcd - gr) * gq) + (b >> 3);
    const int slice = logical / a.nqb, qb = logical - slice * a.nqb;
    const int64_t qtile = (int64_t)qb * WAVES + wv;
    const int64_t nqtiles = (a.nq + 31) >> 5;
    const int64_t q = qtile * 32 + li;
    const bool q_ok = q < a.nq;

    int64_t t0 = a.tile_begin + (int64_t)slice * a.tiles_per_slice;
    int64_t t1 = t0 + a.tiles_per_slice;
    if (t1 > a.tile_end) t1 = a.tile_end;
    const int nt = t1 > t0 ? (int)(t1 - t0) : 0;            // tiles of this slice (the plane table is padded to whole PAD_TILES)
    const int64_t last_tile = (a.N - 1) >> 5;               // tiles from here on hold padding rows
    const int64_t last_rel64 = last_tile - t0;
    const int last_rel = last_rel64 > 0x7fffffff ? 0x7fffffff : last_rel64 < 0 ? 0 : (int)last_rel64;
    const int n_main = (last_rel < nt ? last_rel : nt) & ~1;  // leading tiles free of padding rows, an even number

    v4i bq[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        bq[s] = v4i{0, 0, 0, 0};
        if (qtile < nqtiles) bq[s] = query_frag<KS>(a, qtile, s, lane);
    }
    const uint32_t lds_base = (uint32_t)reinterpret_cast<uintptr_t>(smem);
    const bool loader = wv < KS;                            // waves 0 .. KS - 1 bring in one 1 KiB piece of every tile
    auto prefetch = [&](int t) __attribute__((always_inline)) {
        if (loader) lds_dma16(a.dbplanes + (((t0 + (int64_t)t) * KS + wv) * 64 + lane) * 16, lds_base + (uint32_t)((t % NB) * KS + wv) * 1024u);
    };

    const int CAP = a.cap;
    uint32_t *cand = reinterpret_cast<uint32_t *>(smem + NB * TILE_BYTES) + wv * (CAP * 64);  // [slot][lane]
    int cnt = 0;
    constexpr float BIAS = 3072.0f;                         // see hamming_mfma_kernel: v = BIAS + dot + r / 16
    float thr = 3.0e38f;
    float best[KM];
    v16f cinit;
#pragma unroll
    for (int r = 0; r < 16; ++r) cinit[r] = MODE == 1 ? BIAS + (float)r * 0.0625f : 0.f;
    if (MODE == 1) {
        if (q_ok && !(PS_HM_DEBUG & 64)) thr = BIAS + (float)(a.nbits - 2 * a.thr0[q]);     // 64: nothing passes (the usual exit only)
    } else {
#pragma unroll
        for (int j = 0; j < KM; ++j) best[j] = NO_DOT;
    }
    // Candidate keys of this kernel: [2047 - d (11 bits) | tile index in the slice (17 bits) | r (4 bits)] with d = 1024 + dot and r the
    // row code of the element -- three fields cut straight out of the element's float bits (v = 2048 + d + r / 16 has d in mantissa
    // bits 12..22 and r in bits 8..11: a shift, a bit-field extract, a bit-field insert and an inversion instead of the nine
    // instructions of the (distance, row id) key above).  Ascending keys = ascending distance, then ascending id WITHIN a lane (for a
    // fixed lane half the id grows with (tile, r)); the ids of a query's two lanes interleave (id = 32 tile + 8 (r >> 2) + 4 lh +
    // (r & 3)), so the lists are rewritten as (distance << shift | row id) keys before the two lanes merge.
#if PS_HM_DEBUG & 512
    int ps_dbg_appends = 0, ps_dbg_compactions = 0;
#endif
    auto wave_max = [&](int v) __attribute__((always_inline)) -> int {
        v = max(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true));
        v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true));
        v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true));
        v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true));
        return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
                   max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
    };
    // exact: a lane's column -> its k smallest keys in slots 0..k-1, ascending (selection in LDS, bounded by the fullest column of
    // the wave); the threshold is tightened when the column holds k keys
    auto compact = [&]() __attribute__((always_inline)) {
        const int k = a.k;
        const int mx = wave_max(cnt);
        const int kk = k < mx ? k : mx;
        for (int p = 0; p < kk; ++p) {
            uint32_t bestk = (p < cnt) ? cand[p * 64 + lane] : EMPTY_KEY;
            const uint32_t head = bestk;
            int bj = p;
            for (int j = p + 1; j < mx; ++j) {
                const uint32_t v = (j < cnt) ? cand[j * 64 + lane] : EMPTY_KEY;
                if (v < bestk) { bestk = v; bj = j; }
            }
            if (p < cnt) {
                cand[bj * 64 + lane] = head;
                cand[p * 64 + lane] = bestk;
            }
        }
        cnt = cnt < k ? cnt : k;
        if (cnt == k) {                                     // only a strictly larger dot can still enter: d >= d_k + 2 (dots share their parity)
            const int dk = 2047 - (int)(cand[(k - 1) * 64 + lane] >> 21);
            const float nthr = (float)(2048 + dk + 2);
            thr = nthr > thr ? nthr : thr;
        }
    };
    // Making room DURING the sweep (a column has filled up: its query's bound came out loose).  The selection above walks the column
    // in LDS serially, ~30 K cycles with the whole workgroup waiting at its barrier -- tools/hm_times.py showed the few workgroups
    // that hold such a query setting the kernel's time (260 K cycles against 175-195 K for a CU's pair).  Here the column goes to
    // registers once; the k-th smallest DISTANCE field T is found by bisection (11 steps of compare-and-count over the registers,
    // per lane); keys with a distance beyond T are dropped, the survivors (>= k, plus ties at T) are written back in their
    // order, and the threshold moves to "strictly better than T".  Exact: everything dropped is beaten by k kept keys, and later
    // items (larger ids) that only tie T cannot displace them.  Only if ties leave no room is the serial selection used.
    auto compact_in_sweep = [&](int need) __attribute__((always_inline)) {
        const int k = a.k;
#if PS_HM_DEBUG & 512
        ++ps_dbg_compactions;
#endif
        const int mx = wave_max(cnt);                       // slots to look at (the fullest column of the wave)
        // (the column is re-read from LDS in every step: 28 keys in registers beside the two accumulator sets spill, and scratch
        // traffic inside the counted-vmcnt ring is what tools/check_asm_contracts.py exists to refuse; the reads of a step are
        // independent of each other, so a step costs one LDS round trip, not `mx` of them)
        int lo = 0, hi = 2047;
#pragma unroll 1
        for (int it = 0; it < 11; ++it) {
            const int mid = (lo + hi) >> 1;
            int c = 0;
#pragma unroll 4
            for (int j = 0; j < mx; ++j) {
                const uint32_t key = cand[j * 64 + lane];
                c += (j < cnt && (int)(key >> 21) <= mid) ? 1 : 0;
            }
            if (c >= k) hi = mid; else lo = mid + 1;
        }
        const bool full = cnt >= k;                         // lanes with fewer than k keys keep everything
        int nkeep = 0;
#pragma unroll 4
        for (int j = 0; j < mx; ++j) {
            const uint32_t key = cand[j * 64 + lane];
            nkeep += (j < cnt && (!full || (int)(key >> 21) <= lo)) ? 1 : 0;
        }
        if (__ballot(nkeep > CAP - need) != 0ull) { compact(); return; }       // ties at T fill the column: exact selection by id
        int pos = 0;
#pragma unroll 1
        for (int j = 0; j < mx; ++j) {                      // in place: pos <= j, a slot is read before anything is written to it
            const uint32_t key = cand[j * 64 + lane];
            if (j < cnt && (!full || (int)(key >> 21) <= lo)) { cand[pos * 64 + lane] = key; ++pos; }
        }
        cnt = pos;
        if (full) {
            const float nthr = (float)(2048 + (2047 - lo) + 2);
            thr = nthr > thr ? nthr : thr;
        }
    };
    auto append_bits = [&](uint32_t bits, uint32_t tile4) __attribute__((always_inline)) {   // tile4 = tile index << 4 (a scalar)
        const uint32_t tw = ((bits >> 8) & 15u) | tile4;
        const uint32_t key = ((0xffe00000u & (bits << 9)) | (~0xffe00000u & tw)) ^ 0xffe00000u;
        cand[cnt * 64 + lane] = key;
        ++cnt;
#if PS_HM_DEBUG & 512
        ++ps_dbg_appends;
#endif
    };

    // what the straight-line part leaves behind for a tile: the lane maximum and the wave's two verdicts as SCALARS (`any`: some lane
    // has a hit / a list that changes; `any2`: some lane has a second hit in this tile), computed in the MFMA gaps long before they
    // are branched on -- no branch of the per-tile path waits for a vector compare
    struct Det { float m; unsigned long long any, any2; bool appended; };   // appended: the lane maximum went into the column in the MFMA gaps
    int cmax = 0;                                           // wave-uniform upper bound of the lanes' column fill
    // the part of a tile's epilogue that runs only when some lane of the wave has something to do; `i` = the tile's index relative to t0
    auto finish = [&](const v16f &acc, const Det &d, int i) __attribute__((always_inline)) {
        if (d.any == 0ull) return;
        if (MODE == 0) {
            float x = d.m;
#pragma unroll
            for (int j = 0; j < KM; ++j) {
                const float hi = fmaxf(best[j], x);
                x = fminf(best[j], x);
                best[j] = hi;
            }
            return;
        }
        const uint32_t base = (uint32_t)i << 4;
        // Room in the columns is scalar bookkeeping: `cmax` bounds every lane's fill from above, and the lanes are asked (and, if one
        // of them is really that full, compacted) only when the bound says that the next append might not fit.  A compaction is a
        // serial selection over a lane's column in LDS, ~30 K cycles during which the whole workgroup waits at its barrier
        // (tools/hm_times.py: the ~60 workgroups of a 480-workgroup launch that compacted once or more set the kernel's time, 296 K
        // cycles against 190 K for the rest, when room for a whole tile -- 16 rows per lane -- was demanded after every tile); the
        // usual tile appends ONE key per lane, so room for one is what the next tile needs, and a walk makes room for its own 16.
        auto make_room = [&](int need) __attribute__((always_inline)) {
            cmax = wave_max(cnt);
            if (cmax > CAP - need) {
                compact_in_sweep(need);
                cmax = wave_max(cnt);
            }
        };
        if (d.any2 == 0ull) {                               // the usual hit: one element per lane, its maximum, appended in the MFMA gaps
            if (!d.appended && d.m >= thr) append_bits(__float_as_uint(d.m), base);
            cmax += 1;
        } else {                                            // rare (~1 % of the tiles): every passing element of the lane but the maximum, which is in
            const uint32_t mb = __float_as_uint(d.m);       // (an element carries its row in its fraction bits: no two of a lane are equal)
            bool have_m = d.appended;
            if (cmax > CAP - 16) {                          // might not fit: count what each lane is about to append, compact only if some column really overflows
                int nl = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) nl += (acc[r] >= thr && !(have_m && __float_as_uint(acc[r]) == mb)) ? 1 : 0;
                if (__ballot(cnt + nl > CAP) != 0ull) {
                    // A compaction tightens the threshold to "strictly better than the k-th kept key" because every LATER item has
                    // a larger id -- true at a tile boundary only.  This tile's maximum went in already (the element with the
                    // LARGEST row of the lane's ties), and the rows still to come tie it with smaller ids: take it back (it is the
                    // lane's last key), compact what the earlier tiles left, then append every passing element of this tile.
                    // (tests/test_hip_hamming_mfma.py::test_columns_that_fill_up_during_the_sweep found the other order wrong.)
                    if (have_m && d.m >= thr) --cnt;
                    have_m = false;
                    compact_in_sweep(16);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (acc[r] >= thr && !(have_m && __float_as_uint(acc[r]) == mb)) append_bits(__float_as_uint(acc[r]), base);
            cmax += 16;
        }
        if (cmax > CAP - 1) make_room(1);
    };
    // A second hit of a lane in this tile (exact, see hamming_mfma_kernel): two distinct rows differ in r / 3 or in r % 3, so it
    // exists iff the second-largest maximum over the triples {3 j ..} (sg) or over the residue classes {r % 3 = j} (sh) passes.
#define PS_MAX3(A, B, C) fmaxf(fmaxf(A, B), C)
    auto detect_all = [&](const v16f &x, Det &d) __attribute__((always_inline)) {
        const float a0 = PS_MAX3(x[0], x[1], x[2]), a1 = PS_MAX3(x[3], x[4], x[5]), a2 = PS_MAX3(x[6], x[7], x[8]);
        const float a3 = PS_MAX3(x[9], x[10], x[11]), a4 = PS_MAX3(x[12], x[13], x[14]), a5 = x[15];
        const float mu = PS_MAX3(a0, a1, a2), mv = PS_MAX3(a3, a4, a5);
        d.m = fmaxf(mu, mv);
        d.appended = false;
        if (MODE == 0) { d.any = __ballot(d.m > best[KM - 1]); d.any2 = 0ull; return; }
        d.any = __ballot(d.m >= thr);
        const float sg = PS_MAX3(fminf(mu, mv), __builtin_amdgcn_fmed3f(a0, a1, a2), __builtin_amdgcn_fmed3f(a3, a4, a5));
        const float h0 = fmaxf(PS_MAX3(x[0], x[3], x[6]), PS_MAX3(x[9], x[12], x[15]));
        const float h1 = PS_MAX3(PS_MAX3(x[1], x[4], x[7]), x[10], x[13]);
        const float h2 = PS_MAX3(PS_MAX3(x[2], x[5], x[8]), x[11], x[14]);
        d.any2 = __ballot(fmaxf(sg, __builtin_amdgcn_fmed3f(h0, h1, h2)) >= thr);
    };
    // the MFMA chain of one tile into `w`, the detection of the previous tile `x` in its gaps: 23 vector instructions behind the
    // first four MFMAs (the gap of a 32 x 32 x 64 fp4 MFMA leaves room for about six).  sched_barrier(0) pins the interleaving;
    // PIN: an empty volatile asm that names the accumulator keeps each MFMA at its place -- the results are not needed before the
    // next tile's detection, and without it the compiler sinks the whole chain below the branches of `finish`.
#define PS_PIN(W) asm volatile("" : "+v"(W))
    // The usual hit (one passing element per lane: its maximum) is appended right there, predicated, between the third and the
    // fourth MFMA: every wave of the workgroup then runs the same instructions tile after tile whether it has a hit or not, and
    // nobody arrives late at the next barrier (r04 measurement with the append behind a branch after the chain: sweep with the
    // usual exit only 82 us, complete 142 us -- 2/3 of the wave-tiles have a hit, so every barrier interval had a late wave).
    // Fragments: four registers sets of 16 bytes; 512-bit codes (KS = 8) request the second half into the same registers once the
    // first four MFMAs have been issued (32 fragment registers beside two accumulator sets, the row constants and 32 query registers
    // do not fit the 128 of a four-waves-per-SIMD kernel).
    auto pipe_tile = [&](v16f &w, const unsigned char *tb, const v16f &x, Det &d, int xi) __attribute__((always_inline)) {
        v4i av[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) av[s] = *reinterpret_cast<const v4i *>(tb + s * 1024);
        __builtin_amdgcn_sched_barrier(0);
        w = MODE == 1 ? sign_mfma_first(av[0], bq[0], cinit, 0x7f7f7f7f) : sign_mfma(av[0], bq[0], cinit);
        PS_PIN(w);
        const float a0 = PS_MAX3(x[0], x[1], x[2]), a1 = PS_MAX3(x[3], x[4], x[5]), a2 = PS_MAX3(x[6], x[7], x[8]);
        const float a3 = PS_MAX3(x[9], x[10], x[11]), a4 = PS_MAX3(x[12], x[13], x[14]), a5 = x[15];
        __builtin_amdgcn_sched_barrier(0);
        w = sign_mfma(av[1], bq[1], w);
        PS_PIN(w);
        const float mu = PS_MAX3(a0, a1, a2), mv = PS_MAX3(a3, a4, a5);
        d.m = fmaxf(mu, mv);
        d.any = MODE == 1 ? __ballot(d.m >= thr) : __ballot(d.m > best[KM - 1]);
        d.any2 = 0ull;
        d.appended = MODE == 1;
        float sg = 0.f;
        if (MODE == 1) sg = PS_MAX3(fminf(mu, mv), __builtin_amdgcn_fmed3f(a0, a1, a2), __builtin_amdgcn_fmed3f(a3, a4, a5));
        __builtin_amdgcn_sched_barrier(0);
        w = sign_mfma(av[2], bq[2], w);
        PS_PIN(w);
        if (MODE == 1) {
            if (d.m >= thr) append_bits(__float_as_uint(d.m), (uint32_t)xi << 4);
        }
        __builtin_amdgcn_sched_barrier(0);
        w = sign_mfma(av[3], bq[3], w);
        PS_PIN(w);
        if constexpr (KS == 8) {                            // the second half of the fragments into the same registers
#pragma unroll
            for (int s = 0; s < 4; ++s) av[s] = *reinterpret_cast<const v4i *>(tb + (4 + s) * 1024);
        }
        if (MODE == 1) {                                    // the residue classes' maxima: eight instructions, three live values
            const float h0 = fmaxf(PS_MAX3(x[0], x[3], x[6]), PS_MAX3(x[9], x[12], x[15]));
            const float h1 = PS_MAX3(PS_MAX3(x[1], x[4], x[7]), x[10], x[13]);
            const float h2 = PS_MAX3(PS_MAX3(x[2], x[5], x[8]), x[11], x[14]);
            d.any2 = __ballot(fmaxf(sg, __builtin_amdgcn_fmed3f(h0, h1, h2)) >= thr);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (KS == 8) {
#pragma unroll
            for (int s = 0; s < 4; ++s) { w = sign_mfma(av[s], bq[4 + s], w); PS_PIN(w); }
        }
    };
#undef PS_PIN
#undef PS_MAX3

    __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0): query fragments / bounds are in; from here on the ring counts
    // Ring of NB single tiles, PD = NB - 1 requested ahead.  Step t: own piece of tile t landed -> barrier: everybody's have, and
    // everybody is past the MFMA chain of tile t - 1, i.e. done with its fragments' buffer, which the request for tile t + PD
    // overwrites -> the tile's KS fragments are requested and the MFMA chain runs behind them (each MFMA waits for its own fragment
    // only), with the detection of tile t - 1 in its gaps -> whatever tile t - 1 left to do.
    // (r04 variants measured at 256 bit, sweep with the usual exit only / complete: two-tile entries read and awaited in front of the
    // entry's MFMAs 80 / 136 us; single tiles with the next tile's fragments requested one step ahead into a second register set
    // 85 / 152 us -- the LDS round trip was not what the waves wait for, and its 16 registers are what the steady-state loop needs.)
#pragma unroll
    for (int j = 0; j < PD; ++j)
        if (j < nt) prefetch(j);
    v16f accA, accB;                                        // accumulators of the even / odd tiles
#pragma unroll
    for (int r = 0; r < 16; ++r) { accA[r] = MODE == 1 ? 0.f : NO_DOT; accB[r] = MODE == 1 ? 0.f : NO_DOT; }   // "tile -1": no hit, no list change
    // tile `t` readable: own piece landed (at most the PD - 1 younger requests still in flight), barrier, next request, fragments
    auto open_tile = [&](int t) __attribute__((always_inline)) -> const unsigned char * {
        if (nt - 1 - t >= PD - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PD - 1) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + PD < nt) prefetch(t + PD);
        return smem + (t % NB) * TILE_BYTES + lane * 16;
    };
    auto mask_padding = [&](v16f &w, int i) __attribute__((always_inline)) {
        const int64_t left64 = a.N - (t0 + i) * 32;         // valid rows of this tile (<= 0: a padding tile of the table's end)
        const int left = (int)(left64 < 0 ? 0 : left64 > 32 ? 32 : left64) - 4 * lh;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if ((r & 3) + 8 * (r >> 2) >= left) w[r] = MODE == 1 ? 0.f : NO_DOT;
    };
    Det d;
    int t = 0;
    // Steady state, two tiles per trip (one per accumulator set): the vmcnt literal is a constant, the plane table is addressed by a
    // scalar base that advances one tile per step (LDS-DMA in its scalar-base form: no vector address arithmetic) and the ring
    // slots by two scalar offsets that wrap by compare-and-select -- nothing is computed modulo NB.  (The generic loops below spend
    // ~28 scalar and 5 vector instructions per tile on that bookkeeping.  A version that unrolled 2 NB steps with constant slots
    // was no faster and grew the kernel to 70 KB of code, more than the instruction cache two CUs share: half of the workgroups
    // took 185 K cycles instead of 135 K, tools/hm_times.py.)
    {
        int n_steady = n_main < nt - PD ? n_main : nt - PD;
        n_steady = n_steady > 0 ? n_steady & ~1 : 0;
        const uint32_t voff = (uint32_t)(wv * 1024 + lane * 16);          // this lane's 16 bytes inside a tile of the plane table
        const unsigned char *gnext = a.dbplanes + ((t0 + PD) * (int64_t)KS) * 1024;      // tile t + PD
        const uint32_t lds_w = lds_base + (uint32_t)wv * 1024u;
        const uint32_t frag = lds_base + (uint32_t)lane * 16u;
        uint32_t rd = 0, fill = (uint32_t)((PD % NB) * TILE_BYTES);        // ring offsets of tile t and of tile t + PD
        auto steady_step = [&](v16f &w, const v16f &x, int xi) __attribute__((always_inline)) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PD - 1) : "memory");
            __builtin_amdgcn_s_barrier();
            if (loader)
                asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(gnext), "s"(lds_w + fill) : "memory", "m0");
            gnext += TILE_BYTES;
            const unsigned char *tb = reinterpret_cast<const unsigned char *>(smem) + ((frag + rd) - lds_base);
            rd = rd + TILE_BYTES == (uint32_t)(NB * TILE_BYTES) ? 0u : rd + TILE_BYTES;
            fill = fill + TILE_BYTES == (uint32_t)(NB * TILE_BYTES) ? 0u : fill + TILE_BYTES;
            pipe_tile(w, tb, x, d, xi);
            finish(x, d, xi);
        };
        for (; t < n_steady; t += 2) {
            steady_step(accA, accB, t - 1);
            steady_step(accB, accA, t);
        }
    }
    for (; t < n_main; t += 2) {                            // tiles t, t + 1 < n_main <= nt
        pipe_tile(accA, open_tile(t), accB, d, t - 1);      // tile t, detection of tile t - 1
        finish(accB, d, t - 1);
        pipe_tile(accB, open_tile(t + 1), accA, d, t);      // tile t + 1, detection of tile t
        finish(accA, d, t);
    }
    // the slice's last tiles: those that hold padding rows are masked before anybody looks at them (a few per launch; same ring
    // protocol, the masking is what the loops above are spared)
    for (; t < nt; t += 2) {
        pipe_tile(accA, open_tile(t), accB, d, t - 1);
        if (t >= last_rel) mask_padding(accA, t);
        finish(accB, d, t - 1);
        if (t + 1 < nt) {
            pipe_tile(accB, open_tile(t + 1), accA, d, t);
            if (t + 1 >= last_rel) mask_padding(accB, t + 1);
            finish(accA, d, t);
        } else {                                            // an odd tile count: tile t is the last one
            detect_all(accA, d);
            finish(accA, d, t);
#pragma unroll
            for (int r = 0; r < 16; ++r) accB[r] = MODE == 1 ? 0.f : NO_DOT;     // nothing pending
        }
    }
    if (nt > 0) {                                           // the tile still in flight (none when the count was odd: accB is blank)
        detect_all(accB, d);
        finish(accB, d, nt - 1);
    }

    // ---- results (as hamming_mfma_kernel) ----
    // (the query index is derived again from a thread id the compiler cannot connect with the one of the prologue: kept alive across
    // the sweep it was the one value that no longer fitted the 128 registers at 512 bit -- and a spill is scratch traffic in a kernel
    // whose vmcnt literals count every vector-memory request)
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    const int li2 = tid2 & 31, lh2 = (tid2 >> 5) & 1, lane2 = tid2 & 63;
    const int64_t q2 = qtile * 32 + li2;
    const bool q2_ok = q2 < a.nq;
    if (MODE == 0) {
        if (q2_ok) {
            int32_t *dst = a.bl + ((int64_t)q2 * (a.slices * 2) + slice * 2 + lh2) * KM;
#pragma unroll
            for (int j = 0; j < KM; ++j) dst[j] = best[j] == NO_DOT ? 0x7fffffff : (a.nbits - (int)best[j]) >> 1;
        }
        return;
    }
#if PS_HM_DEBUG & 512
    const unsigned ps_dbg_app_sum = (unsigned)ps_wave_sum_i32(ps_dbg_appends);
    if (MODE == 1 && lane2 == 0 && blockIdx.x < 4096)       // the busiest wave's appends / in-sweep compactions (bits 32.., 56..)
        atomicMax(&ps_hm_times[4 * blockIdx.x + 2], ((unsigned long long)(ps_dbg_app_sum & 0xffffffu) << 32) | ((unsigned long long)(unsigned)ps_dbg_compactions << 56));
    if (MODE == 1 && tid == 0 && blockIdx.x < 4096) {
        ps_hm_times[4 * blockIdx.x] = ps_t_start; ps_hm_times[4 * blockIdx.x + 1] = __builtin_readcyclecounter();
        atomicOr(&ps_hm_times[4 * blockIdx.x + 2], (unsigned long long)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 4));      // HW_ID
        ps_hm_times[4 * blockIdx.x + 3] = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) | ((unsigned long long)slice << 32) | ((unsigned long long)qb << 48);   // XCC_ID, slice, query block
    }
#endif
    compact();
    for (int p = 0; p < cnt; ++p) {                         // (distance << shift | row id in the slice) keys: comparable across the two lanes
        const uint32_t key = cand[p * 64 + lane2];
        const int dot = 2047 - (int)(key >> 21) - 1024;
        const uint32_t r = key & 15u, tile = (key >> 4) & 0x1ffffu;
        cand[p * 64 + lane2] = ((uint32_t)((a.nbits - dot) >> 1) << a.shift) | (tile * 32u + 4u * lh2 + (r & 3u) + 8u * (r >> 2));
    }
    ps_wave_lds_sync();
    const int cnt_hi = __shfl(cnt, li2 + 32, 64);
    if (q2_ok && lh2 == 0) {
        const uint32_t idmask = (1u << a.shift) - 1u;
        const int64_t o = ((int64_t)(a.list_base + slice) * a.nq + q2) * a.k;
        int ia = 0, ib = 0;
        for (int p = 0; p < a.k; ++p) {
            const uint32_t ka = ia < cnt ? cand[ia * 64 + lane2] : EMPTY_KEY;
            const uint32_t kb = ib < cnt_hi ? cand[ib * 64 + lane2 + 32] : EMPTY_KEY;
            const uint32_t key = ka < kb ? ka : kb;
            if (ka < kb) ++ia; else ++ib;
            const bool has = key != EMPTY_KEY;
            a.out_d[o + p] = has ? (int32_t)(key >> a.shift) : 0x7fffffff;
            a.out_r[o + p] = has ? (int32_t)((key & idmask) + (uint32_t)(t0 * 32)) : -1;
        }
    }
}

// Final merge of the per-slice lists: 16 lanes per query, lane j = the head of slice j's sorted list; k rounds of
// "row minimum of the 64-bit keys distance << 32 | table row" (4 DPP exchange steps, no LDS), the winner advances.
// One wave = 4 queries.  ids = row + id_offset; missing entries (-1, INT32_MAX) like faiss.
__device__ __forceinline__ uint64_t row_min_u64(uint64_t v) {
#define PS_STEP(ctrl)                                                                                         \
    {                                                                                                         \
        const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, ctrl, 0xf, 0xf, true); \
        const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), ctrl, 0xf, 0xf, true); \
        const uint64_t o = ((uint64_t)hi << 32) | lo;                                                         \
        v = o < v ? o : v;                                                                                    \
    }
    PS_STEP(0xB1) PS_STEP(0x4E) PS_STEP(0x141) PS_STEP(0x140)
#undef PS_STEP
    return v;
}

__global__ __launch_bounds__(256) void slice_merge_kernel(const int32_t *__restrict__ din, const int32_t *__restrict__ rin,
                                                          int P, int64_t nq, int k, int64_t id_offset,
                                                          int32_t *__restrict__ dout, int64_t *__restrict__ iout) {
    const int lane = threadIdx.x & 63, sub = lane & 15;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t q = wave * 4 + (lane >> 4);
    const bool mine = q < nq && sub < P;
    const int64_t base = mine ? ((int64_t)sub * nq + q) * k : 0;
    constexpr uint64_t NONE = ~0ull;
    auto head = [&](int ptr) -> uint64_t {
        if (!mine || ptr >= k) return NONE;
        const int32_t r = rin[base + ptr];
        return r < 0 ? NONE : ((uint64_t)(uint32_t)din[base + ptr] << 32) | (uint32_t)r;
    };
    int ptr = 0;
    uint64_t key = head(0);
    for (int r = 0; r < k; ++r) {
        const uint64_t m = row_min_u64(key);
        if (sub == 0 && q < nq) {
            dout[q * k + r] = m == NONE ? 0x7fffffff : (int32_t)(m >> 32);
            iout[q * k + r] = m == NONE ? -1 : (int64_t)(uint32_t)m + id_offset;
        }
        if (key == m && m != NONE) key = head(++ptr);
    }
}

// the same merge for up to 64 lists (few queries over a large table are cut into more slices, so that they still fill the
// chip): one wave per query, lane j = the head of list j, wave minimum by DPP + v_readlane
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
#define PS_STEP(ctrl, rows)                                                                                     \
    {                                                                                                           \
        const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)(uint32_t)v, ctrl, rows, 0xf, false); \
        const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)(uint32_t)(v >> 32), ctrl, rows, 0xf, false); \
        const uint64_t o = ((uint64_t)hi << 32) | lo;                                                           \
        v = o < v ? o : v;                                                                                      \
    }
    PS_STEP(0xB1, 0xf) PS_STEP(0x4E, 0xf) PS_STEP(0x141, 0xf) PS_STEP(0x140, 0xf) PS_STEP(0x142, 0xa) PS_STEP(0x143, 0xc)
#undef PS_STEP
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63);
    return ((uint64_t)hi << 32) | lo;
}
__global__ __launch_bounds__(256) void slice_merge64_kernel(const int32_t *__restrict__ din, const int32_t *__restrict__ rin,
                                                            int P, int64_t nq, int k, int64_t id_offset,
                                                            int32_t *__restrict__ dout, int64_t *__restrict__ iout) {
    const int lane = threadIdx.x & 63;
    const int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (q >= nq) return;
    const bool mine = lane < P;
    const int64_t base = mine ? ((int64_t)lane * nq + q) * k : 0;
    constexpr uint64_t NONE = ~0ull;
    auto head = [&](int ptr) -> uint64_t {
        if (!mine || ptr >= k) return NONE;
        const int32_t r = rin[base + ptr];
        return r < 0 ? NONE : ((uint64_t)(uint32_t)din[base + ptr] << 32) | (uint32_t)r;
    };
    int ptr = 0;
    uint64_t key = head(0);
    for (int r = 0; r < k; ++r) {
        const uint64_t m = wave_min_u64(key);
        if (lane == 0) {
            dout[q * k + r] = m == NONE ? 0x7fffffff : (int32_t)(m >> 32);
            iout[q * k + r] = m == NONE ? -1 : (int64_t)(uint32_t)m + id_offset;
        }
        if (key == m && m != NONE) key = head(++ptr);
    }
}

// one wave per query: thr0 = the k-th smallest of the query's group minima (all bound lists), or nbits (admit
// everything) when the sample held fewer than k groups
__global__ __launch_bounds__(256) void bound_select_kernel(const int32_t *__restrict__ bl, int64_t nq, int nvals, int k,
                                                           int nbits, int32_t *__restrict__ thr0) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    if (nvals <= 128) {                                      // the usual case (KM = 4 lists): two values per lane, two ballots per probe
        for (int64_t q = wave; q < nq; q += nw) {
            const int v0 = lane < nvals ? bl[q * nvals + lane] : 0x7fffffff;
            const int v1 = lane + 64 < nvals ? bl[q * nvals + lane + 64] : 0x7fffffff;
            int lo = 0, hi = nbits;                          // smallest d with #(v <= d) >= k, else nbits
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (__popcll(__ballot(v0 <= mid)) + __popcll(__ballot(v1 <= mid)) >= k) hi = mid; else lo = mid + 1;
            }
            if (lane == 0) thr0[q] = lo;
        }
        return;
    }
    for (int64_t q = wave; q < nq; q += nw) {
        int v[16];                                           // nvals <= 1024 (make_plan)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = r * 64 + lane;
            v[r] = c < nvals ? bl[q * nvals + c] : 0x7fffffff;
        }
        int lo = 0, hi = nbits;                              // smallest d with #(v <= d) >= k, else nbits
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            int c = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) c += __popcll(__ballot(v[r] <= mid));
            if (c >= k) hi = mid; else lo = mid + 1;
        }
        if (lane == 0) thr0[q] = lo;
    }
}

int key_shift_bits(int nbits) {
    int dbits = 1;
    while ((1 << (dbits - 1)) < nbits) ++dbits;
    return 32 - dbits;
}

struct Plan {
    bool ok;
    int KS, IT, nqb, slices, bslices, shift, km, cap;
    bool db;                // collect pass: one workgroup per CU with two fragment sets (any k) or two per CU (k <= 12)
    int64_t tiles, tiles_per_slice, sample_tiles, btiles_per_slice;
    size_t off_thr, off_bl, off_i, off_d, off_qp, total;
};

int env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

Plan make_plan(int64_t nq, int64_t N, int cs, int k) {
    Plan p{};
    p.ok = false;
    if (cs % 4 != 0) return p;
    if (cs % 8 != 0) return p;
    p.KS = cs / 8;                                         // 64-bit K steps
    if (!(p.KS == 1 || p.KS == 2 || p.KS == 4 || p.KS == 8)) return p;
    if (k <= 0 || k > 32) return p;
    if (nq < 64 || N < 4096) return p;                    // small problems: the popcount kernel has no tile padding
    const int IT = p.KS >= 4 ? 2 : 4;                      // EntryTiles<KS>
    p.IT = IT;
    p.db = k > 12 || p.KS <= 2 || env_int("PS_HAMMING_MFMA_DB", 0) != 0;   // (KS <= 2: four tiles per entry do not fit 128 VGPRs)
    p.cap = p.db ? (k <= 16 ? 32 : 48) : 28;               // a lane's column: k kept + one tile's 16 rows
    p.shift = key_shift_bits(cs * 8);
    p.tiles = (N + 31) >> 5;
    const int64_t nqt = (nq + 31) >> 5;
    p.nqb = (int)((nqt + WAVES - 1) / WAVES);
    const int slots = env_int("PS_HAMMING_MFMA_SLOTS", 256);      // one 8-wave workgroup per CU (bound pass; collect pass with db)
    int64_t s = (p.db ? slots : 2 * slots) / p.nqb;
    if (s < 1) s = 1;
    if (s > 64) s = 64;                                    // merge fan-in: 16 lanes per query up to 16 lists, a wave beyond
    while (s > 1 && p.tiles / s < 32) --s;                 // a slice is at least 32 tiles (1024 items)
    const int64_t cap_tiles = ((int64_t)1 << p.shift) >> 5;
    if (p.tiles > s * cap_tiles) s = (p.tiles + cap_tiles - 1) / cap_tiles;   // slice-local ids must fit under the distance bits
    p.tiles_per_slice = ((p.tiles + s - 1) / s + IT - 1) / IT * IT;
    p.slices = (int)((p.tiles + p.tiles_per_slice - 1) / p.tiles_per_slice);
    if (p.slices > 64) return p;                           // merge fan-in: one list per slice, one lane per list
    // bound pass: 1/5 of the table (measured on MI355X, 10 000 x 59 047 x 512 bit: 5 % 0.67 ms, 10 % 0.48, 20 % 0.44,
    // 30 % 0.45: every candidate that passes the bound costs ~180 SIMD cycles in the collect pass), at least 16 groups
    // per lane so that a lane's list can hold KM real minima
    int64_t st = env_int("PS_HAMMING_MFMA_SAMPLE_TILES", 0);
    if (st <= 0) st = p.tiles / 5;
    if (st < 64) st = 64;
    if (st > p.tiles) st = p.tiles;
    int64_t bs = (p.db ? slots : 2 * slots) / p.nqb;
    if (bs < 1) bs = 1;
    if (bs > 16) bs = 16;                                  // <= 2 bs lists of KM <= 32 values per query (bound_select: 1024)
    while (bs > 1 && st / bs < 16) --bs;
    // values a lane keeps: 2 bs lanes see a query's sample; with >= 3 k values among them four per lane are enough for the
    // k-th smallest to be (nearly always) the true one, and ANY k distinct groups give a valid bound
    p.btiles_per_slice = ((st + bs - 1) / bs + IT - 1) / IT * IT;
    p.bslices = (int)((st + p.btiles_per_slice - 1) / p.btiles_per_slice);
    p.km = (2 * p.bslices * 4 >= 3 * k) ? 4 : (k <= 16 ? 16 : 32);
    // 512-bit codes keep at most 16 values per lane: with KM = 32 the two fragment sets, 32 query registers and the list do not
    // fit 256 VGPRs (52 bytes of scratch per lane in r03 = vector-memory traffic inside the counted-vmcnt ring).  Any k distinct
    // groups bound the k-th distance, and 2 bslices x 16 >= 32 >= k values are always there; only the tightness of the bound for
    // k > 16 with fewer than three list entries per wanted neighbour changes, never a result.
    if (p.KS == 8 && p.km == 32) p.km = 16;
    p.sample_tiles = st;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    p.off_thr = take((size_t)nq * sizeof(int32_t));
    p.off_bl = take((size_t)nq * p.bslices * 2 * p.km * sizeof(int32_t));
    p.off_i = take((size_t)p.slices * nq * k * sizeof(int32_t));
    p.off_d = take((size_t)p.slices * nq * k * sizeof(int32_t));
    p.off_qp = take(ps_lsh_planes_bytes(nq, cs));          // query planes, when packed query codes meet a kernel that reads planes
    p.total = off + 256;
    p.ok = true;
    return p;
}

// dynamic LDS beyond 64 KiB has to be allowed per kernel (idempotent; remembered per device in an atomic: PsPerDevice)
template <typename K>
bool allow_lds(K kernel, size_t bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) ==
           hipSuccess;
}

template <int KS>
int launch_passes(const Plan &p, HArgs a, hipStream_t st, int32_t *thr0, int32_t *bl, int64_t nq, void *qp_ws) {
    constexpr int IT = EntryTiles<KS>::value;
    constexpr int IT1 = KS >= 8 ? 1 : KS == 4 ? 2 : 4;      // entry of the two-workgroups-per-CU collect kernel
    const size_t tiles_lds = (size_t)NBUF * IT * KS * 1024;
    const size_t lds = p.db ? tiles_lds + (size_t)WAVES * p.cap * 64 * sizeof(uint32_t)
                            : (size_t)3 * IT1 * KS * 1024 + (size_t)WAVES * p.cap * 64 * sizeof(uint32_t);
    // once per (kernel, device): one process may drive several GPUs
    static PsPerDevice lds_done;
    int devid = 0;
    if (hipGetDevice(&devid) != hipSuccess || devid < 0 || devid >= 64) return PS_ELAUNCH;
    bool lds_ok = lds_done.get(devid) != 0;
    if (!lds_ok) {
        lds_ok = allow_lds(hamming_mfma_kernel<KS, 0, 4, true>, 160 * 1024) && allow_lds(hamming_mfma_kernel<KS, 0, 16, true>, 160 * 1024) &&
                 allow_lds(hamming_mfma_kernel<KS, 1, 4, true>, 160 * 1024);
        if constexpr (KS < 8) lds_ok = lds_ok && allow_lds(hamming_mfma_kernel<KS, 0, 32, true>, 160 * 1024);
        // two workgroups per CU: 256- and 512-bit codes only (make_plan: KS <= 2 always takes the one-workgroup form, whose
        // four-tile entries do not fit 128 VGPRs -- those instantiations spilled and were never launched: not built any more)
        if constexpr (KS >= 4)
            lds_ok = lds_ok && allow_lds(hamming_mfma_kernel<KS, 1, 4, false>, 80 * 1024) && allow_lds(hamming_mfma_kernel<KS, 0, 4, false>, 80 * 1024);
        if constexpr (PipeServed<KS>::value)
            lds_ok = lds_ok && allow_lds(hamming_pipe_kernel<KS, 1>, 80 * 1024) && allow_lds(hamming_pipe_kernel<KS, 0>, 80 * 1024);
        lds_done.set(devid, lds_ok ? 1 : 0);
    }
    if (!lds_ok) return PS_ELAUNCH;
    a.nqb = p.nqb;
    a.cap = p.cap;
    // bound
    HArgs b = a;
    b.tile_begin = 0; b.tile_end = p.sample_tiles; b.tiles_per_slice = p.btiles_per_slice; b.slices = p.bslices; b.bl = bl;
    const unsigned gb = (unsigned)(p.nqb * p.bslices);
    bool launched = false;
    // 0: the r03 kernels (cross-check / experiments); the pipelined kernels' keys hold a 17-bit tile index per slice
    const bool pipe = PipeServed<KS>::value && env_int("PS_HAMMING_PIPE", 1) != 0 && p.tiles_per_slice <= (1 << 17) && p.btiles_per_slice <= (1 << 17);
    // packed query codes are expanded by the pipelined kernels' workgroups themselves; when either pass runs a kernel that reads
    // planes (k > 12, 64- / 128-bit codes, PS_HAMMING_PIPE=0) the launcher builds them in the workspace first
    if (a.qplanes == nullptr && !(pipe && !p.db && p.km == 4)) {
        const int64_t pieces = (((nq + 31) / 32 + PAD_TILES - 1) / PAD_TILES * PAD_TILES) * KS * 64;
        int64_t ge = ps_cdiv(pieces, 256);
        if (ge > 256 * 64) ge = 256 * 64;
        hipLaunchKernelGGL(lsh_expand_kernel, dim3((unsigned)ge), dim3(256), 0, st, a.qcodes, nq, KS, pieces, reinterpret_cast<uint4 *>(qp_ws));
        PS_CHECK_LAUNCH();
        a.qplanes = reinterpret_cast<const unsigned char *>(qp_ws);
        a.qcodes = nullptr;
        b.qplanes = a.qplanes;
        b.qcodes = nullptr;
    }
    if constexpr (PipeServed<KS>::value) {
        if (p.km == 4 && !p.db && pipe) {
            hipLaunchKernelGGL((hamming_pipe_kernel<KS, 0>), dim3(gb), dim3(512), (size_t)PipeRing<KS>::value * KS * 1024, st, b);
            launched = true;
        }
    }
    if constexpr (KS >= 4) {
        if (!launched && p.km == 4 && !p.db) {
            hipLaunchKernelGGL((hamming_mfma_kernel<KS, 0, 4, false>), dim3(gb), dim3(512), (size_t)3 * IT1 * KS * 1024, st, b);
            launched = true;
        }
    }
    if constexpr (KS < 8) {
        if (!launched && p.km == 32) {
            hipLaunchKernelGGL((hamming_mfma_kernel<KS, 0, 32, true>), dim3(gb), dim3(512), tiles_lds, st, b);
            launched = true;
        }
    }
    if (!launched && p.km == 4) hipLaunchKernelGGL((hamming_mfma_kernel<KS, 0, 4, true>), dim3(gb), dim3(512), tiles_lds, st, b);
    else if (!launched && p.km == 16) hipLaunchKernelGGL((hamming_mfma_kernel<KS, 0, 16, true>), dim3(gb), dim3(512), tiles_lds, st, b);
    else if (!launched) return PS_EINVAL;
    PS_CHECK_LAUNCH();
    int64_t gs = ps_cdiv(nq, 4);
    if (gs > 4096) gs = 4096;
    hipLaunchKernelGGL(bound_select_kernel, dim3((unsigned)gs), dim3(256), 0, st, bl, nq, p.bslices * 2 * p.km, a.k, a.nbits, thr0);
    PS_CHECK_LAUNCH();
    // collect
    a.tile_begin = 0; a.tile_end = p.tiles; a.tiles_per_slice = p.tiles_per_slice; a.slices = p.slices; a.thr0 = thr0;
    a.list_base = 0;
    const unsigned gc = (unsigned)(p.nqb * p.slices);
    bool collected = false;
    if constexpr (PipeServed<KS>::value) {
        if (!p.db && pipe) {
            hipLaunchKernelGGL((hamming_pipe_kernel<KS, 1>), dim3(gc), dim3(512), (size_t)PipeRing<KS>::value * KS * 1024 + (size_t)WAVES * p.cap * 64 * sizeof(uint32_t), st, a);
            collected = true;
        }
    }
    if constexpr (KS >= 4) {
        if (!collected && !p.db) {
            hipLaunchKernelGGL((hamming_mfma_kernel<KS, 1, 4, false>), dim3(gc), dim3(512), lds, st, a);
            collected = true;
        }
    }
    if (!collected) hipLaunchKernelGGL((hamming_mfma_kernel<KS, 1, 4, true>), dim3(gc), dim3(512), lds, st, a);
    PS_CHECK_LAUNCH();
    return PS_OK;
}

}  // namespace

extern "C" size_t ps_lsh_planes_bytes(int64_t n, int cs) {
    if (n <= 0 || cs <= 0 || cs % 8 != 0) return 0;
    return (size_t)(((n + 31) / 32 + PAD_TILES - 1) / PAD_TILES * PAD_TILES) * (size_t)(cs / 8) * 1024;   // whole ring entries
}

extern "C" int ps_lsh_expand(const uint8_t *codes, int64_t n, int cs, void *planes, ps_stream_t stream) {
    if (n < 0 || cs <= 0) return PS_EINVAL;
    if (cs % 8 != 0) return PS_EUNSUPPORTED;
    if (n == 0) return PS_OK;
    if (!codes || !planes || reinterpret_cast<size_t>(codes) % 4 != 0 || reinterpret_cast<size_t>(planes) % 16 != 0)
        return PS_EINVAL;
    const int KS = cs / 8;
    const int64_t pieces = (((n + 31) / 32 + PAD_TILES - 1) / PAD_TILES * PAD_TILES) * KS * 64;   // padding tiles are zero-filled
    int64_t grid = ps_cdiv(pieces, 256);
    if (grid > 256 * 64) grid = 256 * 64;
    hipLaunchKernelGGL(lsh_expand_kernel, dim3((unsigned)grid), dim3(256), 0, ps_stream(stream),
                       reinterpret_cast<const uint32_t *>(codes), n, KS, pieces, reinterpret_cast<uint4 *>(planes));
    PS_CHECK_LAUNCH();
    return PS_OK;
}

extern "C" size_t ps_hamming_topk_mfma_workspace_bytes(int64_t nq, int64_t N, int cs, int k) {
    const Plan p = make_plan(nq, N, cs, k);
    return p.ok ? p.total : 0;                                // 0 = shape not served by the MFMA path
}

static int hamming_topk_mfma(const void *qplanes, const uint8_t *qcodes, int64_t nq, const void *dbplanes, int64_t N, int cs, int k,
                             int64_t id_offset, int32_t *dist, int64_t *ids, void *workspace, size_t workspace_bytes,
                             ps_stream_t stream) {
    if (nq < 0 || N < 0 || cs <= 0 || k <= 0) return PS_EINVAL;
    const Plan p = make_plan(nq, N, cs, k);
    if (!p.ok) return PS_EUNSUPPORTED;
    if ((!qplanes && !qcodes) || !dbplanes || !dist || !ids) return PS_EINVAL;
    if ((reinterpret_cast<size_t>(qplanes) | reinterpret_cast<size_t>(dbplanes)) % 16 != 0) return PS_EINVAL;
    if (reinterpret_cast<size_t>(qcodes) % 4 != 0) return PS_EINVAL;
    if (!workspace || workspace_bytes < p.total) return PS_EWORKSPACE;
    char *base = reinterpret_cast<char *>((reinterpret_cast<size_t>(workspace) + 255) / 256 * 256);
    int32_t *thr0 = reinterpret_cast<int32_t *>(base + p.off_thr);
    int32_t *bl = reinterpret_cast<int32_t *>(base + p.off_bl);
    int32_t *cr = reinterpret_cast<int32_t *>(base + p.off_i);
    int32_t *cd = reinterpret_cast<int32_t *>(base + p.off_d);
    hipStream_t st = ps_stream(stream);
    HArgs a{};
    a.qplanes = reinterpret_cast<const unsigned char *>(qplanes);
    a.qcodes = qplanes ? nullptr : reinterpret_cast<const uint32_t *>(qcodes);
    a.dbplanes = reinterpret_cast<const unsigned char *>(dbplanes);
    if (N >= ((int64_t)1 << 31)) return PS_EUNSUPPORTED;     // table rows travel as int32 between the passes
    a.nq = nq; a.N = N; a.k = k; a.nbits = cs * 8; a.shift = p.shift; a.id_offset = id_offset;
    a.out_d = cd; a.out_r = cr;
    int rc;
    switch (p.KS) {
        case 1: rc = launch_passes<1>(p, a, st, thr0, bl, nq, base + p.off_qp); break;
        case 2: rc = launch_passes<2>(p, a, st, thr0, bl, nq, base + p.off_qp); break;
        case 4: rc = launch_passes<4>(p, a, st, thr0, bl, nq, base + p.off_qp); break;
        case 8: rc = launch_passes<8>(p, a, st, thr0, bl, nq, base + p.off_qp); break;
        default: return PS_EUNSUPPORTED;
    }
    if (rc != PS_OK) return rc;
    if (p.slices <= 16)
        hipLaunchKernelGGL(slice_merge_kernel, dim3((unsigned)ps_cdiv(nq, 16)), dim3(256), 0, st, cd, cr, p.slices, nq, k, id_offset,
                           dist, ids);
    else
        hipLaunchKernelGGL(slice_merge64_kernel, dim3((unsigned)ps_cdiv(nq, 4)), dim3(256), 0, st, cd, cr, p.slices, nq, k, id_offset,
                           dist, ids);
    PS_CHECK_LAUNCH();
    return PS_OK;
}

extern "C" int ps_hamming_topk_mfma(const void *qplanes, int64_t nq, const void *dbplanes, int64_t N, int cs, int k,
                                    int64_t id_offset, int32_t *dist, int64_t *ids, void *workspace,
                                    size_t workspace_bytes, ps_stream_t stream) {
    if (!qplanes) return PS_EINVAL;
    return hamming_topk_mfma(qplanes, nullptr, nq, dbplanes, N, cs, k, id_offset, dist, ids, workspace, workspace_bytes, stream);
}

extern "C" int ps_hamming_topk_mfma_codes(const uint8_t *qcodes, int64_t nq, const void *dbplanes, int64_t N, int cs, int k,
                                          int64_t id_offset, int32_t *dist, int64_t *ids, void *workspace,
                                          size_t workspace_bytes, ps_stream_t stream) {
    if (!qcodes) return PS_EINVAL;
    return hamming_topk_mfma(nullptr, qcodes, nq, dbplanes, N, cs, k, id_offset, dist, ids, workspace, workspace_bytes, stream);
}
