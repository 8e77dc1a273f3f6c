/* pinsage_hip.h -- C ABI of libpinsage_hip.so, the MI355X (gfx950) implementation of the
 * PinSage hot path of anisanazim/Movie-Recommendation-Engine.
 *
 * The reference has no FFI: its boundary for this path is the Python class surface
 * (utils/random_walk.py, model/pinsage.py, model/aggregators.py, utils/nearest_neighbors.py).
 * Each entry point below names the reference code (file:line, relative to the reference
 * root) whose work it replaces; the Python classes of the same names under
 * movie-recommendation-engine_amd/{utils,model}/ bind these symbols with ctypes
 * (see INTEGRATION.md for the stub a maintainer of the reference would add).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (HBM) unless the parameter name starts with `h_`;
 *    buffers are caller-allocated and caller-owned; inputs are never written.
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing
 *    synchronises, nothing allocates, no state a caller can observe: calls are re-entrant,
 *    callable from several host threads and hipGraph-capturable.  (The only process-wide
 *    data are per-device memos of idempotent one-time queries -- a kernel's dynamic-LDS
 *    attribute has been set, a persistent kernel's resident-workgroup count -- held in
 *    atomics: csrc/ps_common.h PsPerDevice.)
 *  - return value: PS_OK (0) or a negative PS_E* code; `ps_error_string` names it.
 *  - node ids are int32 on the device (V < 2^31), edge offsets int64.
 */
#ifndef PINSAGE_HIP_H
#define PINSAGE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PS_OK 0
#define PS_EINVAL (-1)      /* bad argument / unsupported shape            */
#define PS_ELAUNCH (-2)     /* HIP launch or runtime error                 */
#define PS_EWORKSPACE (-3)  /* workspace too small                         */
#define PS_EUNSUPPORTED (-4)

#define PS_RNG_STREAM 0     /* uniforms[] holds the numpy legacy MT19937 stream */
#define PS_RNG_PHILOX 1     /* Philox4x32-10, counter (node, walk, step/2, call): one block = two steps  */
#define PS_RNG_STREAM_RAW 2 /* ps_walk_sample / ps_walk_sample_layers only: `uniforms` holds the stream as raw MT19937 words
                               (ps_mt19937_raw_stream): uniform i = genrand_res53(temper(word 2i), temper(word 2i+1)) */

#define PS_RNG_STREAM_WALKS 3 /* ps_walk_sample only: PS_RNG_STREAM with ONE stream position per walk, uoff int64[B * W]: step s of
                               walk w of start i reads uniforms[uoff[i * W + w] + s].  For graphs with reachable sinks, where a
                               walk that stops early consumes fewer than L uniforms (utils/random_walk.py:65-69) and the positions
                               are the running count of uniforms consumed by every earlier walk (host: sampling.sink_walk_offsets) */

#define PS_WALK_HALF_BUCKETS 0x100 /* OR-ed into rng_mode of ps_walk_sample / ps_walk_sample_layers: `buckets` holds the 32-byte half
                                      records of ps_bucket_build_half instead of the 64-byte records of ps_bucket_build */

typedef void *ps_stream_t;

int ps_abi_version(void);
const char *ps_error_string(int code);

/* ---- a1: RandomWalkSampler._prepare_adjacency_list (utils/random_walk.py:33-50) -----------
 * adj_list[src].append((dst, w)) in edge-column order  ==  CSR stably sorted by src.
 * src/dst int64[E] (the two rows of edge_index), w float[E] or NULL (=> 1.0, :45-48).
 * Out: rowptr int64[V+1], col int32[E], wsorted double[E] (fp32 weight widened exactly). */
size_t ps_csr_build_workspace_bytes(int64_t E, int64_t V);
int ps_csr_build(const int64_t *src, const int64_t *dst, const float *w, int64_t E, int64_t V,
                 int64_t *rowptr, int32_t *col, double *wsorted,
                 void *workspace, size_t workspace_bytes, ps_stream_t stream);

/* Per-row CDF, the arithmetic `np.random.choice(dest, p=w/w.sum())` performs per step
 * (utils/random_walk.py:76,79): p = w / numpy_sum(w); cdf = cumsum(p); cdf /= cdf[-1], fp64,
 * same operation order (bit-exact).  cdf double[E]. */
int ps_cdf_build(const int64_t *rowptr, const double *wsorted, int64_t V, double *cdf, ps_stream_t stream);

/* Acceleration records for the walk kernels (exact, results unchanged):
 *  nodeinfo uint32[2V] : (row start, out-degree) of every node in one 8-byte record;
 *  guide    int32[E]   : guide[lo_v + j] = #{k : cdf_v[k] <= (j/deg_v)(1 - 2^-50)}, j = 0..deg_v-1 -- a bucket table
 *  for the inverse-CDF lookup: searchsorted(cdf_v, u, 'right') starts at guide[lo_v + floor(u*deg_v)] and
 *  scans forward (usually 1-2 entries) instead of probing log2(deg) cache lines; the threshold sits a hair
 *  below j/deg so that fp64 rounding of u*deg can never put the start past the answer. */
int ps_guide_build(const int64_t *rowptr, const double *cdf, int64_t V, uint32_t *nodeinfo, int32_t *guide,
                   ps_stream_t stream);

/* Packed edge blocks: every 8 consecutive edges e = 8b..8b+7 become one 128-byte line
 *   [ 8 x fp64 cdf | 8 x int32 col | 8 x int32 guide ]   (packed: 128 * ceil(E/8) bytes, 128-B aligned)
 * so the bucket lookup, the CDF scan and the destination read of one walk step usually hit the same line. */
int ps_pack_edges(const int32_t *col, const double *cdf, const int32_t *guide, int64_t E, void *packed,
                  ps_stream_t stream);

/* Bucket records: one 64-byte record per bucket e = lo_v + j of the guide table,
 *   [ c0 c1 | c2 c3 | k0 k1 k2 k3 | c4 k4 - ]   ci fp64 = cdf[lo_v + guide[e] + i], ki int32 = col[same]
 * (past the row end: ci = 2.0, ki = the row's last destination), so searchsorted(cdf_v, u, 'right') = first i with
 * ci > u is answered from ONE 64-byte sector per walk step; a lane that needs more than five candidates repeats the
 * search through the packed blocks.  buckets: 64 * E bytes, 64-B aligned.  Optional (4x the adjacency's memory). */
int ps_bucket_build(const int64_t *rowptr, const int32_t *col, const double *cdf, const int32_t *guide, int64_t V,
                    int64_t E, void *buckets, ps_stream_t stream);

/* The same table at half the size for graphs whose 64-byte records do not fit (BASELINE config 5: 2 x 10^9 edges = 64 GB instead of
 * 128): record lo[v] + j = [lo32(cdf[g]) .. lo32(cdf[g+3]) | col[g] .. col[g+3]] with g the bucket's guide position and lo32 = the
 * fp64 value rounded DOWN to fp32 (cdf lies in [lo32, next float): u < lo32 proves u < cdf, u >= the next float proves u >= cdf);
 * positions past the row end as in ps_bucket_build (2.0f / the row's last destination).  A step is answered from its 32-byte
 * record when these bounds prove one of the four candidates to be the searchsorted pick; a u inside a rounding sliver, or beyond
 * the fourth candidate, makes the walk kernel repeat the search through the packed blocks.  Same results bit for bit.
 * buckets: 32 * E bytes, 64-B aligned; pass PS_WALK_HALF_BUCKETS with the RNG mode. */
int ps_bucket_build_half(const int64_t *rowptr, const int32_t *col, const double *cdf, const int32_t *guide, int64_t V,
                         int64_t E, void *buckets, ps_stream_t stream);

/* Destination records: dest_info[e] = the nodeinfo record (row start, degree) of col[e], 8 bytes per edge, 8-B aligned.  Read
 * contiguously with a start row (the walk kernels stage it into LDS next to the row's blocks), it tells a walk the row of the
 * node it has just picked without a gather of its own: step 1 of a walk is then ONE dependent round trip to memory (the bucket
 * record) instead of two.  Pays where the node records themselves are not cache resident (V * 8 bytes beyond the L2 / MALL:
 * BASELINE config 5, 110 M nodes); optional. */
int ps_dest_info_build(const int32_t *col, const uint32_t *nodeinfo, int64_t E, int64_t V, void *dest_info, ps_stream_t stream);

/* flags[0] = 1 iff some edge points at a node with out-degree 0 (a reachable sink: the
 * reference's walk then breaks early, utils/random_walk.py:68-69, and its RNG consumption
 * becomes data dependent); flags[1] = max out-degree.  flags int64[2]. */
int ps_graph_stats(const int64_t *rowptr, const int32_t *col, int64_t E, int64_t V, int64_t *flags,
                   ps_stream_t stream);

/* ---- a2-a4: _single_walk / sample_neighbors / batch_sample_neighbors (utils/random_walk.py:52-142)
 * For each start node: W walks of L weighted steps (pick = searchsorted(cdf, u, 'right')),
 * visit counts over walk[1:], top-T by (count desc, first-visit order), one wave per start node.
 * rng_mode PS_RNG_STREAM: step (walk w, step s) of start i reads uniforms[uoff[i] + w*L + s]
 *   (uoff int64[B]; the numpy call order on a graph without reachable sinks).
 * rng_mode PS_RNG_STREAM_WALKS: uoff int64[B * W], one position per walk (graphs with reachable sinks; see the define).
 * rng_mode PS_RNG_PHILOX: uniforms/uoff ignored; u = philox(seed; node, w, s/2, call) words (0,1) for even s, (2,3) for odd s.
 * Out: ids int32[B,T] (-1 pad), counts int32[B,T] (0 pad), nvalid int32[B].
 * The reference's weights are counts[i,j] / sum_j counts[i,:nvalid[i]] (:113-115).
 * nodeinfo/guide (both or neither; from ps_guide_build) select the bucket-table lookup; NULL = plain
 * binary search over the CDF row.  packed (from ps_pack_edges; needs nodeinfo) reads cdf/col/guide from
 * the interleaved 128-byte blocks instead of the three arrays; start rows of up to ~40 blocks are searched in LDS.
 * With packed, col / cdf / guide may be NULL (the blocks hold the same values: graphs that fill the GPU drop the plain arrays).
 * buckets (from ps_bucket_build; needs nodeinfo) answers every other step from one 64-byte record; with PS_WALK_HALF_BUCKETS
 * OR-ed into rng_mode it is the 32-byte form of ps_bucket_build_half.
 * dest_info (from ps_dest_info_build; needs packed; may be NULL): the (row start, degree) record of every edge's destination. */
int ps_walk_sample(const int64_t *rowptr, const int32_t *col, const double *cdf, int64_t V,
                   const int64_t *starts, int64_t B, int W, int L, int T,
                   int rng_mode, const double *uniforms, const int64_t *uoff,
                   uint64_t seed, uint32_t call, const uint32_t *nodeinfo, const int32_t *guide,
                   const void *packed, const void *buckets, const void *dest_info,
                   int32_t *ids, int32_t *counts, int32_t *nvalid, ps_stream_t stream);

/* PinSage.get_embeddings draws one fresh sample per GCN layer for the SAME start nodes (model/pinsage.py:271-275:
 * `for layer in range(num_layers): batch_sample_neighbors(nodes, num_neighbors)`).  This entry point runs `layers`
 * consecutive samples of every start node in one wave: the start row is brought into LDS once and the per-node fixed
 * chain (start id -> row bounds -> row) is paid once (few start nodes -- a rank's shard -- get one wave per (node, layer)
 * instead: shorter waves fill the last generation of resident waves better).  Results are exactly those of `layers` ps_walk_sample calls:
 * PS_RNG_PHILOX uses call, call + 1, ...; PS_RNG_STREAM reads layer r's uniforms at r * layer_stride + uoff[i] + w*L + s
 * (layer_stride = the uniforms one whole batch consumes = total[0] of ps_uniform_offsets, i.e. the reference's order:
 * all of layer 0's draws, then all of layer 1's).  ids/counts int32[layers, B, T], nvalid int32[layers, B]. */
int ps_walk_sample_layers(const int64_t *rowptr, const int32_t *col, const double *cdf, int64_t V,
                          const int64_t *starts, int64_t B, int W, int L, int T,
                          int rng_mode, const double *uniforms, const int64_t *uoff, int64_t layer_stride,
                          uint64_t seed, uint32_t call, const uint32_t *nodeinfo, const int32_t *guide,
                          const void *packed, const void *buckets, const void *dest_info, int layers,
                          int32_t *ids, int32_t *counts, int32_t *nvalid, ps_stream_t stream);

/* _single_walk (utils/random_walk.py:52-83), batched: one walk of L steps per start node, one lane
 * per walk.  paths int32[B,L]: the visited nodes after the start (-1 once the walk hit a sink).
 * PS_RNG_STREAM: walk i, step s reads uniforms[uoff[i] + s]; PS_RNG_PHILOX: philox(seed; node, w, s/2, call)
 * with walk id w = i (walk_mod == 0) or i % walk_mod (replays walk w of ps_walk_sample when the start
 * nodes are repeated W = walk_mod times). */
int ps_walk_paths(const int64_t *rowptr, const int32_t *col, const double *cdf, int64_t V,
                  const int64_t *starts, int64_t B, int L, int rng_mode, const double *uniforms,
                  const int64_t *uoff, uint64_t seed, uint32_t call, int walk_mod,
                  const uint32_t *nodeinfo, const int32_t *guide, int32_t *paths, ps_stream_t stream);

/* Offsets into the numpy stream: uoff[i] = W*L * #{j < i : outdeg(starts[j]) > 0};
 * total[0] = uniforms consumed by the whole batch. */
int ps_uniform_offsets(const int64_t *rowptr, int64_t V, const int64_t *starts, int64_t B, int W, int L,
                       int64_t *uoff, int64_t *total, ps_stream_t stream);

/* numpy legacy MT19937 `random_sample` on the device: state uint32[624] + pos (0..624) in; `skip` doubles are
 * skipped, then n doubles written; the state after skip + n draws comes back (what np.random.set_state needs).
 * Replaces the global-RNG draws of np.random.choice at utils/random_walk.py:79.
 * jump_polys uint32[jump_levels, 624] (row m = t^(2^m) mod phi over GF(2), from pinsage_hip/mtjump.py) enables
 * the parallel path (2^c-word chunks, c = ps_mt19937_chunk_log2(), generated by independent workgroups, windows by
 * jump-ahead); with NULL polynomials / workspace a single workgroup generates the stream serially (skip must be 0).
 * radix_polys uint32[radix_levels, 31, 624] (entry (i, j-1) = t^(j * 2^(c + 5i)) mod phi, optional): the chunk windows
 * are then produced in radix-32 rounds instead of by doubling.
 * window_polys uint32[n_window, 624] (row j-1 = t^(j * 2^c + s) mod phi, s = ps_mt19937_window_shift() -- the generator
 * expands the sequence s words back from the first window as well as forwards; optional; mtjump.window_polynomials):
 * requests of 33 .. n_window + 1 chunks get ALL their windows in one product round from the first window.
 * ranges (ps_mt19937_raw_stream only; a HOST array int64[n_ranges][2], n_ranges <= 3, NULL = everything): runs [lo, hi) of
 * uniform indices the caller will read -- a rank of an item-sharded job passes the stream positions of ITS start nodes
 * (one run per GCN layer); only those words of `raw` (and the state hand-back) are generated, the rest of the buffer is
 * left unwritten.  The returned state is the post-whole-stream state either way.  Needs window_polys (else ignored). */
/* ps_mt19937_raw_stream: the same stream (skip = 0) left as untempered 32-bit state words in raw uint32[2n + 1248] for
 * PS_RNG_STREAM_RAW -- the walk kernel tempers and combines the two words of a uniform itself, which saves the conversion
 * pass (65 us and 190 MB of traffic per 23.6 M doubles).  Needs the jump polynomials (n >= 2^17). */
int ps_mt19937_chunk_log2(void);
int ps_mt19937_window_shift(void);
int ps_mt19937_raw_stream(const uint32_t *state_in, int pos_in, int64_t n, uint32_t *raw, uint32_t *state_out,
                          int32_t *pos_out, const uint32_t *jump_polys, int jump_levels, const uint32_t *radix_polys,
                          int radix_levels, const uint32_t *window_polys, int n_window, const int64_t *ranges,
                          int n_ranges, void *workspace, size_t workspace_bytes, ps_stream_t stream);
size_t ps_mt19937_workspace_bytes(int64_t skip, int64_t n);
int ps_mt19937_random_sample(const uint32_t *state_in, int pos_in, int64_t skip, int64_t n, double *out,
                             uint32_t *state_out, int32_t *pos_out, const uint32_t *jump_polys, int jump_levels,
                             const uint32_t *radix_polys, int radix_levels, const uint32_t *window_polys, int n_window,
                             void *workspace, size_t workspace_bytes, ps_stream_t stream);

/* ---- a5 / a9: ImportancePooling.forward (model/pinsage.py:101-150); Weighted/Mean/Importance
 * aggregators' gather + weighted reduce (model/aggregators.py:13-91,233-287).
 * x float[N,H]; ids/counts int32[B,T]; nvalid int32[B]; out float[B,H].
 * Row i: keep j < nvalid[i] with 0 <= ids[i,j] <= max_idx (:123-129); w_j = fp32(count_j/total);
 * w /= sum(w) if > 0 (:140-143); out = sum_j w_j * x[ids_j] (:146); zeros if none (:115-117,:132-134).
 * If wts != NULL (float[B,T]) it supplies w_j directly instead of counts (list API with arbitrary
 * weights; aggregators).  renorm = 0 skips the `w /= sum(w)` step (weights already final). */
int ps_importance_pool(const float *x, int64_t N, int H, const int32_t *ids, const int32_t *counts,
                       const float *wts, const int32_t *nvalid, int64_t B, int T, int64_t max_idx,
                       int renorm, float *out, ps_stream_t stream);

/* ---- a6: the dense layers of PinSage.forward (model/pinsage.py:202,235-240,248-249) ---------
 * y[M,N] = epilogue( x[M,K] W[N,K]^T (+ x2[M,K2] W2[N,K2]^T) + b ), fp32 MFMA, k-ordered fma chain.
 * x2/W2 eliminate torch.cat([h_self, h_neigh]) (:238): W = lin_update.weight[:, :H], W2 = [:, H:]
 * (ldw / ldw2 = row strides of W / W2 in floats).  flags: PS_RELU, PS_L2NORM (F.normalize, eps 1e-12), PS_WPERM: W (and W2)
 * are stored in the order the kernel stages them (ps_permute_k; needs 16-byte aligned operands and K % 32 == 0, else PS_EINVAL):
 * same results bit for bit, 32 fewer vector instructions per 64 MFMAs -- for weights that are multiplied many times. */
#define PS_RELU 1
#define PS_L2NORM 2
#define PS_WPERM 4
/* out float[rows, K] (dense) = W with every group of eight k as k 0 2 4 6 1 3 5 7 (K % 8 == 0; ld = row stride of W). */
int ps_permute_k(const float *W, int64_t rows, int K, int ld, float *out, ps_stream_t stream);
int ps_linear(const float *x, int64_t M, int K, const float *W, int ldw, const float *b, int N,
              const float *x2, int K2, const float *W2, int ldw2, int flags, float *y, ps_stream_t stream);

/* ---- a10: LSHIndex.build/search (utils/nearest_neighbors.py:28-68 -> faiss.IndexLSH) ---------
 * codes[n, nbits/8] : bit j = ( x . A[j,:] >= 0 ), LSB-first (faiss fvec2bitvec); A float[nbits,D].
 * flags: 0 or PS_WPERM (A stored by ps_permute_k). */
int ps_lsh_encode(const float *x, int64_t N, int D, const float *A, int nbits, uint8_t *codes, int flags, ps_stream_t stream);

/* Hamming k-NN over all codes: k smallest by (distance, id), ascending; id = row + id_offset.
 * dist int32[nq,k] (INT32_MAX pad), ids int64[nq,k] (-1 pad).  cs = bytes per code (multiple of 4). */
size_t ps_hamming_topk_workspace_bytes(int64_t nq, int64_t N, int cs, int k);
int ps_hamming_topk(const uint8_t *qcodes, int64_t nq, const uint8_t *codes, int64_t N, int cs, int k,
                    int64_t id_offset, int32_t *dist, int64_t *ids,
                    void *workspace, size_t workspace_bytes, ps_stream_t stream);

/* The same search as an exact contraction on the matrix cores (csrc/hamming_mfma.hip): every code is expanded
 * ONCE into "sign planes" -- bit j of a code -> an fp4 (e2m1) +1 / -1, laid out tile by tile (32 codes x 64 bits = 1 KiB)
 * in the order v_mfma_scale_f32_32x32x64_f8f6f4 consumes them -- and dot(q, x) = nbits - 2 * hamming(q, x) exactly (the
 * products are +-1, the partial sums integers far below 2^24).
 * Replaces the same reference lines as ps_hamming_topk (utils/nearest_neighbors.py:47-68 -> faiss hammings_knn_hc)
 * and returns bit-identical (dist, ids).
 *   ps_lsh_planes_bytes(n, cs)      : size of the plane table of n codes = 4 x the packed codes (0 if cs % 8 != 0)
 *   ps_lsh_expand(codes, n, cs, pl) : builds it (16-B aligned); done at LSHIndex.build time for the table (and per call for
 *                                     the queries by callers of ps_hamming_topk_mfma)
 *   ps_hamming_topk_mfma_codes      : the same search with the QUERIES given as packed codes (4-B aligned, cs bytes each):
 *                                     every workgroup of the scan expands its own 32 queries in registers, which saves
 *                                     the ps_lsh_expand launch in front of a search (the table's planes are still built
 *                                     once per index)
 *   ps_hamming_topk_mfma_workspace_bytes : 0 when the shape is not served (k > 32, code size not 8 / 16 / 32 / 64 bytes,
 *                                     fewer than 64 queries or 4096 items): callers use ps_hamming_topk then;
 *                                     ps_hamming_topk_mfma itself returns PS_EUNSUPPORTED for such shapes. */
size_t ps_lsh_planes_bytes(int64_t n, int cs);
int ps_lsh_expand(const uint8_t *codes, int64_t n, int cs, void *planes, ps_stream_t stream);
size_t ps_hamming_topk_mfma_workspace_bytes(int64_t nq, int64_t N, int cs, int k);
int ps_hamming_topk_mfma(const void *qplanes, int64_t nq, const void *dbplanes, int64_t N, int cs, int k,
                         int64_t id_offset, int32_t *dist, int64_t *ids,
                         void *workspace, size_t workspace_bytes, ps_stream_t stream);
int ps_hamming_topk_mfma_codes(const uint8_t *qcodes, int64_t nq, const void *dbplanes, int64_t N, int cs, int k,
                               int64_t id_offset, int32_t *dist, int64_t *ids,
                               void *workspace, size_t workspace_bytes, ps_stream_t stream);

/* Merge P sorted candidate lists per query (multi-GPU shards: all-gathered [P, nq, k]) into the
 * global k best by (distance, id). */
int ps_topk_merge(const int32_t *dist_in, const int64_t *ids_in, int P, int64_t nq, int k,
                  int32_t *dist, int64_t *ids, ps_stream_t stream);
/* The same merge over candidate lists that are not packed back to back: shard p's [nq, k] distances start at
 * dist_in + p * dist_stride, its ids at ids_in + p * ids_stride (strides in elements, >= nq * k).  Lets every rank write
 * (ids | dist) into ONE record that a single all-gather exchanges (pinsage_hip/shard.py: [P, nq*k*8 B ids | nq*k*4 B dist]). */
int ps_topk_merge_strided(const int32_t *dist_in, int64_t dist_stride, const int64_t *ids_in, int64_t ids_stride,
                          int P, int64_t nq, int k, int32_t *dist, int64_t *ids, ps_stream_t stream);

/* ---- a11: exact search (inference.py:112-118, utils/evaluation.py:106-132) --------------------
 * sim = E[q] . E^T ; optionally sim[q] = -inf ; top-k descending.  vals float[nq,k], ids int64[nq,k].  Any k >= 1
 * (k > 32: further sweeps over the similarity row, each admitting the keys after the last one emitted; k > N pads with
 * (-inf, -1)); the same holds for ps_l2_topk. */
size_t ps_dot_topk_workspace_bytes(int64_t nq, int64_t N, int D, int k);
int ps_dot_topk(const float *E, int64_t N, int D, const int64_t *qidx, int64_t nq, int k, int exclude_self,
                float *vals, int64_t *ids, void *workspace, size_t workspace_bytes, ps_stream_t stream);

/* ---- next row: exact L2 / IVF search behind WeakANDIndex and benchmark_search_methods
 * (utils/nearest_neighbors.py:70-139, 176: faiss.IndexFlatL2 / faiss.IndexIVFFlat).
 * dist(q, x) = |q|^2 + |x|^2 - 2 q.x in fp32; k smallest by (distance, id), ascending; dist float[nq,k]
 * (FLT_MAX pad), ids int64[nq,k] (-1 pad).  With assign int32[N] (inverted-list id of every item) and
 * probe uint32[nq, words] (bit l set = list l is probed by that query) only probed lists are visible. */
size_t ps_l2_topk_workspace_bytes(int64_t nq, int64_t N, int D, int k);
int ps_l2_topk(const float *X, int64_t N, int D, const float *Q, int64_t nq, int k, const int32_t *assign,
               const uint32_t *probe, int words, float *dist, int64_t *ids, void *workspace,
               size_t workspace_bytes, ps_stream_t stream);

/* The inverted-file form of the same scan (faiss.IndexIVFFlat, utils/nearest_neighbors.py:88-93; nprobe = min(nlist, 20), :134):
 * X float[N,D] holds the items SORTED BY LIST (list l = rows [list_ptr[l], list_ptr[l+1]), list_ptr int64[nlist+1]),
 * max_list = the longest list, item_ids int64[N] = the original id of every sorted row, probes int32[nq, nprobe] = the lists
 * each query visits (entries outside [0, nlist) are skipped).  The (query, list) pairs are grouped by list on the device,
 * one grouped fp32-MFMA product multiplies every list's queries with that list's rows only, and a query's top-k sweeps its
 * nprobe result rows: nq * nprobe * (list length) dot products instead of nq * N.  Same arithmetic per (query, item) as
 * ps_l2_topk: identical (dist, ids) to the masked form.
 * Precondition: list_ptr non-decreasing with list_ptr[0] = 0, list_ptr[nlist] = N, and max_list >= the longest list (the
 * workspace is sized from it).  A list that breaks it is clamped ON THE DEVICE to [0, N) and to max_list rows -- its tail is not
 * searched -- so no access ever leaves the workspace or X. */
size_t ps_ivf_topk_workspace_bytes(int64_t nq, int64_t N, int D, int k, int nlist, int nprobe, int64_t max_list);
int ps_ivf_topk(const float *X, int64_t N, int D, const int64_t *list_ptr, int nlist, int64_t max_list,
                const int64_t *item_ids, const float *Q, int64_t nq, const int32_t *probes, int nprobe, int k,
                float *dist, int64_t *ids, void *workspace, size_t workspace_bytes, ps_stream_t stream);

/* ---- next row: the aggregation of GraphConv.propagate (model/pinsage.py:53-54, 70-92; PyG
 * MessagePassing, aggr='add', flow source->target): out[r] = sum over the edges e of CSR row r of val[e] * x[col[e]].
 * rowptr int64[V+1] / col int32[E] = edges grouped by TARGET node (ps_csr_build with the edge_index rows swapped),
 * val float[E] = edge_weight * importance_weight in that order, or NULL (= 1); x float[N,H]; out float[V,H]
 * (rowptr[V] == E).  fp32 accumulation in edge order inside a 512-edge slice; rows cut by a slice are combined with
 * atomics, so long rows are reproducible only up to fp32 re-association, like PyG's scatter-add. */
int ps_spmm_csr(const int64_t *rowptr, const int32_t *col, const float *val, const float *x, int64_t N, int H,
                int64_t V, int64_t E, float *out, ps_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PINSAGE_HIP_H */
