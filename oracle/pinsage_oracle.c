/* CPU oracle, plain-C restatement of the reference PinSage hot path.
 *
 * TEST INFRASTRUCTURE ONLY: linked/loaded by tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg -- never by the product path.
 *
 * It follows oracle/pinsage_oracle.py function by function (which is pinned to the
 * reference by tests/golden/reference_golden.npz) and is itself checked against that
 * numpy restatement and the goldens in tests/test_oracle_golden.py / test_c_oracle.py.
 * Reference citations are relative to the reference repo root.
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off: no FMA contraction, so the
 * fp64 CDF arithmetic is the IEEE sequence numpy executes).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_OK 0
#define ORC_EINVAL -1
#define ORC_ENOMEM -2
#define ORC_EUNIFORMS -3

/* ---------- numpy ndarray.sum() for contiguous fp64 (utils/random_walk.py:76) ------ */
static double pairwise_block(const double *a, int64_t n) {
    if (n < 8) {
        double res = 0.0;
        for (int64_t i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; j++) r[j] = a[j];
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise_block(a, n2) + pairwise_block(a + n2, n - n2);
    }
}

double orc_np_sum(const double *a, int64_t n) {
    double res = 0.0;                      /* ufunc buffer: 8192-element chunks, sequential */
    for (int64_t i = 0; i < n; i += 8192) {
        int64_t m = n - i < 8192 ? n - i : 8192;
        res += pairwise_block(a + i, m);
    }
    return res;
}

/* ---------- a1: adjacency -> CSR (utils/random_walk.py:33-50), stable by src -------- */
int orc_csr_build(const int64_t *src, const int64_t *dst, const float *w /*nullable*/,
                  int64_t E, int64_t V, int64_t *rowptr, int32_t *col, double *wsorted) {
    memset(rowptr, 0, sizeof(int64_t) * (size_t)(V + 1));
    for (int64_t e = 0; e < E; e++) {
        if (src[e] < 0 || src[e] >= V || dst[e] < 0 || dst[e] >= V) return ORC_EINVAL;
        rowptr[src[e] + 1]++;
    }
    for (int64_t v = 0; v < V; v++) rowptr[v + 1] += rowptr[v];
    int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * (size_t)(V > 0 ? V : 1));
    if (!cur) return ORC_ENOMEM;
    memcpy(cur, rowptr, sizeof(int64_t) * (size_t)V);
    for (int64_t e = 0; e < E; e++) {
        int64_t p = cur[src[e]]++;
        col[p] = (int32_t)dst[e];
        wsorted[p] = w ? (double)w[e] : 1.0;      /* :45-48 */
    }
    free(cur);
    return ORC_OK;
}

/* p = w / w.sum(); cdf = p.cumsum(); cdf /= cdf[-1]  (random_walk.py:76 + RandomState.choice) */
int orc_cdf_build(const int64_t *rowptr, const double *w, int64_t V, double *cdf, int threads) {
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 256) num_threads(threads > 0 ? threads : 1)
#endif
    for (int64_t v = 0; v < V; v++) {
        int64_t lo = rowptr[v], hi = rowptr[v + 1];
        if (hi == lo) continue;
        double S = orc_np_sum(w + lo, hi - lo);
        double acc = 0.0;
        for (int64_t i = lo; i < hi; i++) {
            double p = w[i] / S;
            acc = (i == lo) ? p : acc + p;
            cdf[i] = acc;
        }
        double last = cdf[hi - 1];
        for (int64_t i = lo; i < hi; i++) cdf[i] = cdf[i] / last;
    }
    return ORC_OK;
}

/* ---------- Philox4x32-10 ---------------------------------------------------------- */
static inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    philox4x32_10(c, key[0], key[1]);
    memcpy(out, c, sizeof(c));
}

static inline double philox_uniform(uint64_t seed, uint32_t call, uint32_t node, uint32_t walk, uint32_t step) {
    /* one block serves two consecutive steps: words 0,1 for the even step, 2,3 for the odd one */
    uint32_t c[4] = {node, walk, step >> 1, call};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint32_t a = (step & 1) ? c[2] : c[0], b = (step & 1) ? c[3] : c[1];
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

/* searchsorted(cdf[lo:hi], u, side='right') */
static inline int64_t upper_bound(const double *cdf, int64_t lo, int64_t hi, double u) {
    while (lo < hi) {
        int64_t mid = lo + ((hi - lo) >> 1);
        if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
    }
    return lo;
}

/* ---------- a2-a4: walks + visit-count top-T (utils/random_walk.py:52-142) ----------
 * rng_mode 0: numpy stream.  uoff == NULL -> strictly sequential consumption (exact for
 *   any graph, incl. sinks, random_walk.py:68-69), single thread.  uoff != NULL ->
 *   uoff[i] is start node i's offset into `uniforms` (valid when no sink is reachable:
 *   every walk takes all L steps), parallel over start nodes.
 * rng_mode 1: philox(seed, call).
 * Outputs: ids int64[B,T] (-1 pad), counts int32[B,T] (0 pad), nvalid int32[B],
 *   weights fp64[B,T] = count / sum(top counts) (random_walk.py:113-115), *consumed. */
int orc_walk_sample(const int64_t *rowptr, const int32_t *col, const double *cdf, int64_t V,
                    const int64_t *starts, int64_t B, int W, int L, int T,
                    int rng_mode, const double *uniforms, int64_t n_uniforms, const int64_t *uoff,
                    uint64_t seed, uint32_t call,
                    int64_t *ids, int32_t *counts, int32_t *nvalid, double *weights,
                    int64_t *consumed, int64_t *probes /*nullable: sum of ceil(log2(d+1))*/, int threads) {
    int P = W * L;
    int hs = 16;
    while (hs < 2 * P) hs <<= 1;
    int64_t seqpos = 0, nprobes = 0;
    int rc = ORC_OK;
    int par = (rng_mode == 1 || uoff != NULL) ? (threads > 0 ? threads : 1) : 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(par) reduction(+ : nprobes)
#endif
    {
        int32_t *visited = (int32_t *)malloc(sizeof(int32_t) * (size_t)P);
        int32_t *hkey = (int32_t *)malloc(sizeof(int32_t) * (size_t)hs);
        int32_t *hidx = (int32_t *)malloc(sizeof(int32_t) * (size_t)hs);
        int32_t *uid = (int32_t *)malloc(sizeof(int32_t) * (size_t)P);
        int32_t *ucnt = (int32_t *)malloc(sizeof(int32_t) * (size_t)P);
        char *taken = (char *)malloc((size_t)P);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
        for (int64_t i = 0; i < B; i++) {
            int64_t s = starts[i];
            int64_t *oid = ids + i * T;
            int32_t *ocn = counts + i * T;
            double *ow = weights + i * T;
            for (int t = 0; t < T; t++) { oid[t] = -1; ocn[t] = 0; ow[t] = 0.0; }
            nvalid[i] = 0;
            if (s < 0 || s >= V) { rc = ORC_EINVAL; continue; }   /* reference: IndexError */
            int nv = 0;
            int64_t upos = uoff ? uoff[i] : 0;
            for (int w = 0; w < W; w++) {
                int64_t cur = s;
                for (int st = 0; st < L; st++) {
                    int64_t lo = rowptr[cur], hi = rowptr[cur + 1];
                    if (hi == lo) break;                               /* :68-69 */
                    double u;
                    if (rng_mode == 1) u = philox_uniform(seed, call, (uint32_t)s, (uint32_t)w, (uint32_t)st);
                    else {
                        int64_t idx = uoff ? upos++ : seqpos++;
                        if (idx >= n_uniforms) { rc = ORC_EUNIFORMS; u = 0.0; } else u = uniforms[idx];
                    }
                    int64_t d = hi - lo, lg = 0;
                    while (((int64_t)1 << lg) < d + 1) lg++;
                    nprobes += lg;
                    int64_t k = upper_bound(cdf, lo, hi, u);
                    cur = col[k];
                    visited[nv++] = (int32_t)cur;
                }
            }
            if (nv == 0) continue;                                      /* :109-110 */
            /* Counter in first-visit order (:101-104) */
            for (int h = 0; h < hs; h++) hkey[h] = -1;
            int D = 0;
            for (int p = 0; p < nv; p++) {
                uint32_t h = ((uint32_t)visited[p] * 2654435761u) & (uint32_t)(hs - 1);
                while (hkey[h] != -1 && hkey[h] != visited[p]) h = (h + 1) & (uint32_t)(hs - 1);
                if (hkey[h] == -1) { hkey[h] = visited[p]; hidx[h] = D; uid[D] = visited[p]; ucnt[D] = 0; D++; }
                ucnt[hidx[h]]++;
            }
            /* stable sort by count desc, keep T (:107) */
            memset(taken, 0, (size_t)D);
            int K = D < T ? D : T;
            int64_t tot = 0;
            for (int t = 0; t < K; t++) {
                int best = -1;
                for (int j = 0; j < D; j++)
                    if (!taken[j] && (best < 0 || ucnt[j] > ucnt[best])) best = j;
                taken[best] = 1;
                oid[t] = uid[best];
                ocn[t] = ucnt[best];
                tot += ucnt[best];
            }
            for (int t = 0; t < K; t++) ow[t] = (double)ocn[t] / (double)tot;   /* :113-115 */
            nvalid[i] = K;
        }
        free(visited); free(hkey); free(hidx); free(uid); free(ucnt); free(taken);
    }
    if (consumed) *consumed = seqpos;
    if (probes) *probes = nprobes;
    return rc;
}

/* _single_walk (random_walk.py:52-83): out[0..len) incl. start; returns len via *outlen */
int orc_single_walk(const int64_t *rowptr, const int32_t *col, const double *cdf, int64_t start, int L,
                    const double *uniforms, int64_t *pos, int64_t *out, int *outlen) {
    int n = 0;
    int64_t cur = start;
    out[n++] = start;
    for (int st = 0; st < L; st++) {
        int64_t lo = rowptr[cur], hi = rowptr[cur + 1];
        if (hi == lo) break;
        double u = uniforms[(*pos)++];
        cur = col[upper_bound(cdf, lo, hi, u)];
        out[n++] = cur;
    }
    *outlen = n;
    return ORC_OK;
}

/* ---------- a5: ImportancePooling (model/pinsage.py:101-150) ------------------------ */
int orc_importance_pool(const float *x, int64_t N, int H, const int64_t *ids, const int32_t *counts,
                        const int32_t *nvalid, int64_t B, int T, float *out, int threads) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
    for (int64_t i = 0; i < B; i++) {
        float *o = out + i * H;
        for (int h = 0; h < H; h++) o[h] = 0.f;
        int k = nvalid[i];
        if (k == 0) continue;
        int64_t tot = 0;
        for (int j = 0; j < k; j++) tot += counts[i * T + j];
        float wsum = 0.f;
        for (int j = 0; j < k; j++)
            if (ids[i * T + j] <= N - 1) wsum += (float)((double)counts[i * T + j] / (double)tot);
        for (int j = 0; j < k; j++) {
            int64_t id = ids[i * T + j];
            if (id > N - 1) continue;                                   /* :123-129 */
            float wj = (float)((double)counts[i * T + j] / (double)tot); /* torch.tensor(fp64) -> fp32 :140 */
            if (wsum > 0.f) wj = wj / wsum;                               /* :141-143 */
            const float *r = x + id * H;
            for (int h = 0; h < H; h++) o[h] += r[h] * wj;               /* :146 */
        }
    }
    return ORC_OK;
}

/* ---------- dense: y = act(x W^T + b), optional row L2 normalise (F.normalize eps 1e-12) --
 * k-ordered fp32 fma chain from 0, bias added last (what the MFMA kernel computes).  */
int orc_linear(const float *x, const float *W, const float *b, int64_t M, int K, int N,
               const float *x2, const float *W2, int K2, int relu, int l2norm, float *y, int threads) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
    for (int64_t m = 0; m < M; m++) {
        float *o = y + m * N;
        for (int n = 0; n < N; n++) {
            float acc = 0.f;
            for (int k = 0; k < K; k++) acc = fmaf(x[m * K + k], W[(int64_t)n * K + k], acc);
            if (x2) for (int k = 0; k < K2; k++) acc = fmaf(x2[m * K2 + k], W2[(int64_t)n * K2 + k], acc);
            if (b) acc += b[n];
            if (relu && acc < 0.f) acc = 0.f;
            o[n] = acc;
        }
        if (l2norm) {
            float ss = 0.f;
            for (int n = 0; n < N; n++) ss += o[n] * o[n];
            float nrm = sqrtf(ss);
            if (nrm < 1e-12f) nrm = 1e-12f;
            for (int n = 0; n < N; n++) o[n] = o[n] / nrm;
        }
    }
    return ORC_OK;
}

/* ---------- a10: LSH encode (faiss IndexLSH.sa_encode restated; parity unpinned) ------ */
int orc_lsh_encode(const float *x, int64_t N, int D, const float *A, int nbits, uint8_t *codes, int threads) {
    int cs = (nbits + 7) / 8;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
    for (int64_t i = 0; i < N; i++) {
        uint8_t *c = codes + i * cs;
        memset(c, 0, (size_t)cs);
        for (int j = 0; j < nbits; j++) {
            float acc = 0.f;
            for (int k = 0; k < D; k++) acc = fmaf(x[i * D + k], A[(int64_t)j * D + k], acc);
            if (acc >= 0.f) c[j >> 3] |= (uint8_t)(1u << (j & 7));     /* fvec2bitvec: >= 0, LSB first */
        }
    }
    return ORC_OK;
}

/* Hamming k-NN, k smallest by (dist, id), ascending (faiss hammings_knn_hc + reorder) */
int orc_hamming_topk(const uint8_t *q, int64_t nq, const uint8_t *codes, int64_t N, int cs, int k,
                     int64_t id_offset, float *dist, int64_t *ids, int threads) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
    for (int64_t i = 0; i < nq; i++) {
        int32_t *bd = (int32_t *)malloc(sizeof(int32_t) * (size_t)k);
        int64_t *bi = (int64_t *)malloc(sizeof(int64_t) * (size_t)k);
        int n = 0;
        for (int64_t j = 0; j < N; j++) {
            int32_t d = 0;
            const uint8_t *a = q + i * cs, *b = codes + j * cs;
            int w = 0;
            for (; w + 8 <= cs; w += 8) {
                uint64_t xa, xb;
                memcpy(&xa, a + w, 8); memcpy(&xb, b + w, 8);
                d += __builtin_popcountll(xa ^ xb);
            }
            for (; w < cs; w++) d += __builtin_popcount((unsigned)(a[w] ^ b[w]));
            if (n < k || d < bd[n - 1]) {                               /* strict: ids ascend */
                int p = n < k ? n : k - 1;
                while (p > 0 && bd[p - 1] > d) { bd[p] = bd[p - 1]; bi[p] = bi[p - 1]; p--; }
                bd[p] = d; bi[p] = j + id_offset;
                if (n < k) n++;
            }
        }
        for (int t = 0; t < k; t++) {
            dist[i * k + t] = t < n ? (float)bd[t] : 2147483647.0f;
            ids[i * k + t] = t < n ? bi[t] : -1;
        }
        free(bd); free(bi);
    }
    return ORC_OK;
}

/* a11 exact: sim = q . E^T (k-ordered fma), optional self exclusion, top-k by (sim desc, id asc) */
int orc_dot_topk(const float *E, int64_t N, int D, const int64_t *qidx, int64_t nq, int k, int exclude_self,
                 float *vals, int64_t *ids, int threads) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
    for (int64_t i = 0; i < nq; i++) {
        float *bv = (float *)malloc(sizeof(float) * (size_t)k);
        int64_t *bi = (int64_t *)malloc(sizeof(int64_t) * (size_t)k);
        int n = 0;
        const float *q = E + qidx[i] * D;
        for (int64_t j = 0; j < N; j++) {
            float acc = 0.f;
            for (int d = 0; d < D; d++) acc = fmaf(q[d], E[j * D + d], acc);
            if (exclude_self && j == qidx[i]) acc = -INFINITY;
            if (n < k || acc > bv[n - 1]) {
                int p = n < k ? n : k - 1;
                while (p > 0 && bv[p - 1] < acc) { bv[p] = bv[p - 1]; bi[p] = bi[p - 1]; p--; }
                bv[p] = acc; bi[p] = j;
                if (n < k) n++;
            }
        }
        for (int t = 0; t < k; t++) { vals[i * k + t] = t < n ? bv[t] : -INFINITY; ids[i * k + t] = t < n ? bi[t] : -1; }
        free(bv); free(bi);
    }
    return ORC_OK;
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
