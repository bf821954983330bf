/*
 * wtp_oracle_impl.h — type-generic body of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 * Included twice by wtp_oracle.c with REAL = float / double and SUF = f32 / f64.
 *
 * This is a restatement, not a copy: the reference is Julia and its k-NN arithmetic lives in
 * NearestNeighbors.jl / Distances.jl (not vendored).  Each function cites the reference
 * file:line it follows (paths relative to the reference checkout).
 */

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

/* ---- canonical squared distance ------------------------------------------------------
 * Distances.jl Euclidean as called from NearestNeighbors (reference call sites
 * src/repel.jl:259, src/topology.jl:80-81): s = 0; s += (a[i]-b[i])^2 in coordinate
 * order, sqrt applied last.  ((dx*dx + dy*dy) + dz*dz) in REAL, no FMA contraction
 * (the file is compiled with -ffp-contract=off). */
static inline REAL FN(d2)(const REAL* a, const REAL* b, int dim) {
    REAL dx = a[0] - b[0];
    REAL dy = a[1] - b[1];
    REAL s = dx * dx + dy * dy;
    if (dim == 3) {
        REAL dz = a[2] - b[2];
        s = s + dz * dz;
    }
    return s;
}

/* canonical total order on (d2, index) */
static inline int FN(lt)(REAL da, int32_t ia, REAL db, int32_t ib) {
    return (da < db) || (da == db && ia < ib);
}

/* ---- brute-force k-NN -----------------------------------------------------------------
 * _build_knn_neighbors (src/topology.jl:79-84): k+1 sorted hits, self dropped — here
 * self is removed BY INDEX (include_self == 0) or kept (include_self == 1, the raw
 * `search` result of src/neighbors.jl:9-14).  O(n^2); the pin for every faster path. */
int FN(wtpo_knn_brute)(const REAL* xyz, int64_t n, int dim, int k, int include_self,
                       int32_t* idx_out, REAL* dist_out) {
    if (n < 1 || k < 1 || (dim != 2 && dim != 3)) return 1;
    if ((int64_t)k > n - (include_self ? 0 : 1)) return 1;
#pragma omp parallel
    {
        REAL* bd = (REAL*)malloc(sizeof(REAL) * (size_t)k);
        int32_t* bi = (int32_t*)malloc(sizeof(int32_t) * (size_t)k);
#pragma omp for schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            int m = 0;
            const REAL* q = xyz + i * dim;
            for (int64_t j = 0; j < n; ++j) {
                if (!include_self && j == i) continue;
                REAL d = FN(d2)(q, xyz + j * dim, dim);
                if (m == k && !FN(lt)(d, (int32_t)j, bd[k - 1], bi[k - 1])) continue;
                int p = (m < k) ? m++ : k - 1;
                while (p > 0 && FN(lt)(d, (int32_t)j, bd[p - 1], bi[p - 1])) {
                    bd[p] = bd[p - 1];
                    bi[p] = bi[p - 1];
                    --p;
                }
                bd[p] = d;
                bi[p] = (int32_t)j;
            }
            for (int t = 0; t < k; ++t) {
                idx_out[i * k + t] = bi[t];
                if (dist_out) dist_out[i * k + t] = SQRT(bd[t]);
            }
        }
        free(bd);
        free(bi);
    }
    return 0;
}

/* ---- brute-force radius search -----------------------------------------------------------
 * _build_radius_neighbors (src/topology.jl:91-97): inrange is inclusive (d <= r, compared
 * as d2 <= r*r with r*r formed in REAL), self removed by index (filter(!=(i), n)).
 * Reference row order is tree-traversal order (unspecified); the canonical order here is
 * ascending (d2, index). */
int FN(wtpo_radius_count_brute)(const REAL* xyz, int64_t n, int dim, REAL r, int32_t* counts) {
    if (n < 1 || (dim != 2 && dim != 3) || !(r >= 0)) return 1;
    REAL r2 = r * r;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        int32_t c = 0;
        for (int64_t j = 0; j < n; ++j)
            if (j != i && FN(d2)(xyz + i * dim, xyz + j * dim, dim) <= r2) ++c;
        counts[i] = c;
    }
    return 0;
}

int FN(wtpo_radius_fill_brute)(const REAL* xyz, int64_t n, int dim, REAL r,
                               const int64_t* offsets, int32_t* idx_out) {
    if (n < 1 || (dim != 2 && dim != 3) || !(r >= 0)) return 1;
    REAL r2 = r * r;
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < n; ++i) {
        int32_t* row = idx_out + offsets[i];
        int64_t cap = offsets[i + 1] - offsets[i];
        REAL* rd = (REAL*)malloc(sizeof(REAL) * (size_t)(cap > 0 ? cap : 1));
        int64_t m = 0;
        for (int64_t j = 0; j < n; ++j) {
            if (j == i) continue;
            REAL d = FN(d2)(xyz + i * dim, xyz + j * dim, dim);
            if (!(d <= r2) || m >= cap) continue;
            int64_t p = m++;
            while (p > 0 && FN(lt)(d, (int32_t)j, rd[p - 1], row[p - 1])) {
                rd[p] = rd[p - 1];
                row[p] = row[p - 1];
                --p;
            }
            rd[p] = d;
            row[p] = (int32_t)j;
        }
        free(rd);
    }
    return 0;
}

/* ---- kd-tree (the CPU "port" of the reference's search structure) --------------------------
 * NearestNeighbors.jl KDTree as the reference uses it (src/repel.jl:218,252;
 * Meshes KNearestSearch behind src/topology.jl:80): points reordered into leaf buckets,
 * split on the widest extent, exact search with a size-k bounded heap and
 * hyper-rectangle lower bounds.  The canonical (d2,index) order replaces the package's
 * traversal-order tie rule (parity unpinned there). */
typedef struct FN(kdnode) {
    int32_t lo, hi;      /* point range [lo,hi) in the reordered arrays */
    int32_t left, right; /* children, -1 for a leaf */
    int32_t sdim;
    REAL split;
} FN(kdnode);

typedef struct FN(kdtree) {
    int64_t n;
    int dim;
    REAL* pts;     /* reordered n x dim */
    int32_t* perm; /* reordered slot -> original index */
    FN(kdnode)* nodes;
    int32_t n_nodes, cap_nodes;
    REAL bbmin[3], bbmax[3];
} FN(kdtree);

#define KD_LEAF 10

static void FN(kd_select)(REAL* pts, int32_t* perm, int dim, int32_t lo, int32_t hi, int32_t nth,
                          int sd) {
    /* quickselect on coordinate sd, ties by original index for determinism */
    REAL tmp[3];
    while (hi - lo > 1) {
        int32_t mid = lo + (hi - lo) / 2;
        REAL pv = pts[(int64_t)mid * dim + sd];
        int32_t pi = perm[mid];
        int32_t i = lo, j = hi - 1;
        while (i <= j) {
            while (pts[(int64_t)i * dim + sd] < pv ||
                   (pts[(int64_t)i * dim + sd] == pv && perm[i] < pi))
                ++i;
            while (pts[(int64_t)j * dim + sd] > pv ||
                   (pts[(int64_t)j * dim + sd] == pv && perm[j] > pi))
                --j;
            if (i <= j) {
                for (int d = 0; d < dim; ++d) {
                    tmp[d] = pts[(int64_t)i * dim + d];
                    pts[(int64_t)i * dim + d] = pts[(int64_t)j * dim + d];
                    pts[(int64_t)j * dim + d] = tmp[d];
                }
                int32_t t = perm[i];
                perm[i] = perm[j];
                perm[j] = t;
                ++i;
                --j;
            }
        }
        if (nth <= j)
            hi = j + 1;
        else if (nth >= i)
            lo = i;
        else
            return;
    }
}

static int32_t FN(kd_build_rec)(FN(kdtree)* t, int32_t lo, int32_t hi) {
    if (t->n_nodes == t->cap_nodes) {
        t->cap_nodes *= 2;
        t->nodes = (FN(kdnode)*)realloc(t->nodes, sizeof(FN(kdnode)) * (size_t)t->cap_nodes);
    }
    int32_t id = t->n_nodes++;
    t->nodes[id].lo = lo;
    t->nodes[id].hi = hi;
    t->nodes[id].left = t->nodes[id].right = -1;
    t->nodes[id].sdim = 0;
    t->nodes[id].split = 0;
    if (hi - lo <= KD_LEAF) return id;
    int dim = t->dim, sd = 0;
    REAL best = -1;
    for (int d = 0; d < dim; ++d) {
        REAL mn = t->pts[(int64_t)lo * dim + d], mx = mn;
        for (int32_t i = lo + 1; i < hi; ++i) {
            REAL v = t->pts[(int64_t)i * dim + d];
            if (v < mn) mn = v;
            if (v > mx) mx = v;
        }
        if (mx - mn > best) {
            best = mx - mn;
            sd = d;
        }
    }
    int32_t mid = lo + (hi - lo) / 2;
    FN(kd_select)(t->pts, t->perm, dim, lo, hi, mid, sd);
    REAL split = t->pts[(int64_t)mid * dim + sd];
    int32_t l = FN(kd_build_rec)(t, lo, mid);
    int32_t r = FN(kd_build_rec)(t, mid, hi);
    t->nodes[id].left = l;
    t->nodes[id].right = r;
    t->nodes[id].sdim = sd;
    t->nodes[id].split = split;
    return id;
}

static FN(kdtree)* FN(kd_build)(const REAL* xyz, int64_t n, int dim) {
    FN(kdtree)* t = (FN(kdtree)*)calloc(1, sizeof(FN(kdtree)));
    t->n = n;
    t->dim = dim;
    t->pts = (REAL*)malloc(sizeof(REAL) * (size_t)(n * dim));
    t->perm = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    memcpy(t->pts, xyz, sizeof(REAL) * (size_t)(n * dim));
    for (int64_t i = 0; i < n; ++i) t->perm[i] = (int32_t)i;
    t->cap_nodes = (int32_t)(2 * (n / (KD_LEAF / 2) + 8));
    t->nodes = (FN(kdnode)*)malloc(sizeof(FN(kdnode)) * (size_t)t->cap_nodes);
    for (int d = 0; d < dim; ++d) {
        REAL mn = xyz[d], mx = xyz[d];
        for (int64_t i = 1; i < n; ++i) {
            REAL v = xyz[i * dim + d];
            if (v < mn) mn = v;
            if (v > mx) mx = v;
        }
        t->bbmin[d] = mn;
        t->bbmax[d] = mx;
    }
    FN(kd_build_rec)(t, 0, (int32_t)n);
    return t;
}

static void FN(kd_free)(FN(kdtree)* t) {
    if (!t) return;
    free(t->pts);
    free(t->perm);
    free(t->nodes);
    free(t);
}

typedef struct FN(kdq) {
    const FN(kdtree)* t;
    const REAL* q;
    int k, m;
    int32_t skip; /* original index to ignore, or -1 */
    REAL* bd;     /* sorted ascending (d2,idx), length m */
    int32_t* bi;
} FN(kdq);

/* Lower bounds are formed in long double, i.e. (nearly) exactly — but a candidate's d2 is evaluated
 * in REAL and may round BELOW its exact value by a few ulp.  On inputs full of exact ties
 * (lattices) an exact bound therefore pruned subtrees holding points whose REAL d2 ties or beats
 * the current k-th.  The bound is relaxed by 8 REAL ulps before it is compared. */
#define KD_SLACK (1.0L - 8.0L * (long double)REAL_EPS)
static void FN(kd_search)(FN(kdq)* s, int32_t node, long double lb, long double off[3]) {
    const FN(kdtree)* t = s->t;
    const FN(kdnode)* nd = &t->nodes[node];
    if (s->m == s->k && lb > (long double)s->bd[s->k - 1]) return;
    if (nd->left < 0) {
        int dim = t->dim;
        for (int32_t p = nd->lo; p < nd->hi; ++p) {
            int32_t j = t->perm[p];
            if (j == s->skip) continue;
            REAL d = FN(d2)(s->q, t->pts + (int64_t)p * dim, dim);
            if (s->m == s->k && !FN(lt)(d, j, s->bd[s->k - 1], s->bi[s->k - 1])) continue;
            int pos = (s->m < s->k) ? s->m++ : s->k - 1;
            while (pos > 0 && FN(lt)(d, j, s->bd[pos - 1], s->bi[pos - 1])) {
                s->bd[pos] = s->bd[pos - 1];
                s->bi[pos] = s->bi[pos - 1];
                --pos;
            }
            s->bd[pos] = d;
            s->bi[pos] = j;
        }
        return;
    }
    int sd = nd->sdim;
    long double diff = (long double)s->q[sd] - (long double)nd->split;
    int32_t near = diff < 0 ? nd->left : nd->right;
    int32_t far = diff < 0 ? nd->right : nd->left;
    FN(kd_search)(s, near, lb, off);
    long double old = off[sd];
    long double nlb = lb - old * old + diff * diff;
    if (fabsl(diff) < fabsl(old)) nlb = lb; /* never tighten below the accumulated bound */
    off[sd] = fabsl(diff) > fabsl(old) ? diff : old;
    FN(kd_search)(s, far, nlb * KD_SLACK, off);
    off[sd] = old;
}

static void FN(kd_knn)(const FN(kdtree)* t, const REAL* q, int k, int32_t skip, REAL* bd,
                       int32_t* bi, int* m_out) {
    FN(kdq) s;
    s.t = t;
    s.q = q;
    s.k = k;
    s.m = 0;
    s.skip = skip;
    s.bd = bd;
    s.bi = bi;
    long double off[3] = {0, 0, 0};
    long double lb = 0;
    for (int d = 0; d < t->dim; ++d) {
        long double o = 0;
        if (q[d] < t->bbmin[d]) o = (long double)t->bbmin[d] - q[d];
        if (q[d] > t->bbmax[d]) o = (long double)q[d] - t->bbmax[d];
        off[d] = o;
        lb += o * o;
    }
    FN(kd_search)(&s, 0, lb * KD_SLACK, off);
    *m_out = s.m;
}

int FN(wtpo_knn_kdtree)(const REAL* xyz, int64_t n, int dim, int k, int include_self,
                        int32_t* idx_out, REAL* dist_out) {
    if (n < 1 || k < 1 || (dim != 2 && dim != 3)) return 1;
    if ((int64_t)k > n - (include_self ? 0 : 1)) return 1;
    FN(kdtree)* t = FN(kd_build)(xyz, n, dim);
#pragma omp parallel
    {
        REAL* bd = (REAL*)malloc(sizeof(REAL) * (size_t)k);
        int32_t* bi = (int32_t*)malloc(sizeof(int32_t) * (size_t)k);
#pragma omp for schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            int m;
            FN(kd_knn)(t, xyz + i * dim, k, include_self ? -1 : (int32_t)i, bd, bi, &m);
            for (int j = 0; j < k; ++j) {
                idx_out[i * k + j] = bi[j];
                if (dist_out) dist_out[i * k + j] = SQRT(bd[j]);
            }
        }
        free(bd);
        free(bi);
    }
    FN(kd_free)(t);
    return 0;
}

/* inrange on the same tree: counts (fill == NULL) or rows sorted by (d2,index) */
static void FN(kd_range)(const FN(kdtree)* t, int32_t node, const REAL* q, REAL r2, int32_t skip,
                         long double lb, long double off[3], int32_t* row, REAL* rd, int64_t* m,
                         int64_t cap) {
    const FN(kdnode)* nd = &t->nodes[node];
    if (lb > (long double)r2) return;
    if (nd->left < 0) {
        int dim = t->dim;
        for (int32_t p = nd->lo; p < nd->hi; ++p) {
            int32_t j = t->perm[p];
            if (j == skip) continue;
            REAL d = FN(d2)(q, t->pts + (int64_t)p * dim, dim);
            if (!(d <= r2)) continue;
            if (row) {
                if (*m >= cap) continue;
                int64_t pos = (*m)++;
                while (pos > 0 && FN(lt)(d, j, rd[pos - 1], row[pos - 1])) {
                    rd[pos] = rd[pos - 1];
                    row[pos] = row[pos - 1];
                    --pos;
                }
                rd[pos] = d;
                row[pos] = j;
            } else {
                ++(*m);
            }
        }
        return;
    }
    int sd = nd->sdim;
    long double diff = (long double)q[sd] - (long double)nd->split;
    int32_t near = diff < 0 ? nd->left : nd->right;
    int32_t far = diff < 0 ? nd->right : nd->left;
    FN(kd_range)(t, near, q, r2, skip, lb, off, row, rd, m, cap);
    long double old = off[sd];
    long double nlb = lb - old * old + diff * diff;
    if (fabsl(diff) < fabsl(old)) nlb = lb;
    off[sd] = fabsl(diff) > fabsl(old) ? diff : old;
    FN(kd_range)(t, far, q, r2, skip, nlb * KD_SLACK, off, row, rd, m, cap);
    off[sd] = old;
}

int FN(wtpo_radius_kdtree)(const REAL* xyz, int64_t n, int dim, REAL r, int32_t* counts,
                           const int64_t* offsets, int32_t* idx_out) {
    if (n < 1 || (dim != 2 && dim != 3) || !(r >= 0)) return 1;
    REAL r2 = r * r;
    FN(kdtree)* t = FN(kd_build)(xyz, n, dim);
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < n; ++i) {
        const REAL* q = xyz + i * dim;
        long double off[3] = {0, 0, 0};
        int64_t m = 0;
        if (idx_out) {
            int64_t cap = offsets[i + 1] - offsets[i];
            REAL* rd = (REAL*)malloc(sizeof(REAL) * (size_t)(cap > 0 ? cap : 1));
            FN(kd_range)(t, 0, q, r2, (int32_t)i, 0, off, idx_out + offsets[i], rd, &m, cap);
            free(rd);
        } else {
            FN(kd_range)(t, 0, q, r2, (int32_t)i, 0, off, NULL, NULL, &m, 0);
            counts[i] = (int32_t)m;
        }
    }
    FN(kd_free)(t);
    return 0;
}

/* ---- force laws ---------------------------------------------------------------------------
 * src/repel_forces.jl:37 (inverse distance), :57-60 (spacing equilibrium), :96-100
 * (clipped spacing, the default), :124-127 (strong spacing).  Parameters are held in REAL
 * (ClippedSpacingForce(0.5f0) stays Float32: test/repel.jl:175-182). */
REAL FN(wtpo_force)(int kind, REAL beta, REAL u0, REAL gamma, REAL u) {
    REAL u2 = u * u;
    switch (kind) {
        case 0: {
            REAL d = u2 + beta;
            return 1 / (d * d);
        }
        case 1: {
            REAL d = u2 + beta;
            return (1 - u2) / (d * d);
        }
        case 2: {
            REAL d = u2 + beta;
            REAL f = (u0 * u0 - u2) / (d * d);
            return f > 0 ? f : 0;
        }
        default:
            return (1 - u2) / POW(u2 + beta, gamma);
    }
}

/* deterministic stand-in for the randn direction of _safe_direction when r == 0
 * (src/repel.jl:358-364): the reference draws from an unseeded task-local RNG, so this
 * branch is unpinned by construction; parity inputs avoid coincident points. */
static void FN(fallback_dir)(int64_t i, int64_t j, int dim, REAL* out) {
    uint64_t z = (uint64_t)i * 0x9E3779B97F4A7C15ull + (uint64_t)j * 0xBF58476D1CE4E5B9ull + 1;
    REAL v[3];
    REAL nn = 0;
    for (int d = 0; d < dim; ++d) {
        z ^= z >> 30;
        z *= 0xBF58476D1CE4E5B9ull;
        z ^= z >> 27;
        z *= 0x94D049BB133111EBull;
        z ^= z >> 31;
        v[d] = (REAL)((double)(z >> 11) * (1.0 / 9007199254740992.0)) - (REAL)0.5;
        nn += v[d] * v[d];
    }
    nn = SQRT(nn);
    if (!(nn > 0)) {
        v[0] = 1;
        nn = 1;
        for (int d = 1; d < dim; ++d) v[d] = 0;
    }
    for (int d = 0; d < dim; ++d) out[d] = v[d] / nn;
}

/* ---- one Jacobi sweep of _relax! ---------------------------------------------------------
 * src/repel.jl:256-292.  snap: n x dim search snapshot (fixed head, movable tail mirroring
 * p_old when the tree is fresh); p_old: n_move x dim; spacings: n values at snapshot
 * positions (repel.jl:209,251) — `s = spacing(xi)` (repel.jl:260) is spacings[id+n_fixed].
 * Outputs: p_new, forces, nn_dist (REAL), nn_id (0-based snapshot index, -1 if none).
 * Arithmetic order follows the Julia expressions term by term, all in REAL. */
int FN(wtpo_relax_sweep)(const REAL* snap, int64_t n, int64_t n_fixed, int dim,
                         const REAL* p_old, const REAL* spacings, int force_kind, REAL beta,
                         REAL u0, REAL gamma, int k, REAL alpha_lo, REAL alpha_max, REAL* p_new,
                         REAL* forces, REAL* nn_dist, int32_t* nn_id) {
    if (n < 1 || n_fixed < 0 || n_fixed > n || k < 1 || (dim != 2 && dim != 3)) return 1;
    int64_t n_move = n - n_fixed;
    int kk = (int64_t)k < n ? k : (int)n; /* repel.jl:208 */
    FN(kdtree)* t = FN(kd_build)(snap, n, dim);
#pragma omp parallel
    {
        REAL* bd = (REAL*)malloc(sizeof(REAL) * (size_t)kk);
        int32_t* bi = (int32_t*)malloc(sizeof(int32_t) * (size_t)kk);
#pragma omp for schedule(static)
        for (int64_t id = 0; id < n_move; ++id) {
            const REAL* xi = p_old + id * dim;
            int m;
            FN(kd_knn)(t, xi, kk, -1, bd, bi, &m); /* knn!(ids, dists, tree, xi, kk, true) :259 */
            REAL s = spacings[id + n_fixed];       /* :260 */
            int32_t self = (int32_t)(id + n_fixed);/* :266 */
            int32_t nid = -1;
            REAL ndist = REAL_MAX;
            REAL F[3] = {0, 0, 0};
            for (int j = 0; j < m; ++j) { /* :270-280 */
                if (bi[j] == self) continue;
                REAL r = SQRT(bd[j]);
                if (nid < 0) {
                    nid = bi[j];
                    ndist = r;
                }
                const REAL* xj = snap + (int64_t)bi[j] * dim;
                REAL f = FN(wtpo_force)(force_kind, beta, u0, gamma, r / s);
                REAL dir[3] = {0, 0, 0};
                if (r > 0) {
                    for (int d = 0; d < dim; ++d) dir[d] = (xi[d] - xj[d]) / r; /* :360 */
                } else {
                    FN(fallback_dir)(self, bi[j], dim, dir);
                }
                for (int d = 0; d < dim; ++d) F[d] = F[d] + f * dir[d];
            }
            REAL Fn = F[0] * F[0] + F[1] * F[1];
            if (dim == 3) Fn = Fn + F[2] * F[2];
            Fn = SQRT(Fn);         /* :282 */
            forces[id] = Fn * s;   /* :283 */
            REAL a = 1 / (Fn + (REAL)1.0e-30); /* :285 */
            if (a < alpha_lo) a = alpha_lo;
            if (a > alpha_max) a = alpha_max;
            REAL sa = s * a; /* s * α_i * repel_force parses as (s*α_i)*F  :286 */
            REAL disp[3] = {0, 0, 0};
            for (int d = 0; d < dim; ++d) disp[d] = sa * F[d];
            REAL dn = disp[0] * disp[0] + disp[1] * disp[1];
            if (dim == 3) dn = dn + disp[2] * disp[2];
            dn = SQRT(dn); /* :287 */
            if (dn > s) {  /* :288-290 */
                REAL sc = s / dn;
                for (int d = 0; d < dim; ++d) disp[d] = disp[d] * sc;
            }
            for (int d = 0; d < dim; ++d) p_new[id * dim + d] = xi[d] + disp[d]; /* :291 */
            nn_dist[id] = ndist;
            nn_id[id] = nid;
        }
        free(bd);
        free(bi);
    }
    FN(kd_free)(t);
    return 0;
}

/* _dnn_cv (src/repel.jl:374-386): serial sums in the promoted type (REAL here). */
REAL FN(wtpo_dnn_cv)(const REAL* nn_dist, const REAL* spacings, int64_t n_move, int64_t n_fixed,
                     double* sum_u, double* sum_u2) {
    REAL s1 = 0, s2 = 0;
    double d1 = 0, d2 = 0;
    for (int64_t i = 0; i < n_move; ++i) {
        REAL u = nn_dist[i] / spacings[i + n_fixed];
        s1 += u;
        s2 += u * u;
        d1 += (double)u;
        d2 += (double)u * (double)u;
    }
    if (sum_u) *sum_u = d1;
    if (sum_u2) *sum_u2 = d2;
    REAL mu = s1 / (REAL)n_move;
    REAL var = s2 / (REAL)n_move - mu * mu;
    if (var < 0) var = 0;
    return SQRT(var) / mu;
}

/* _closest_pair (src/repel.jl:396-403): argmin(nn_dist) takes the FIRST minimum. */
void FN(wtpo_closest_pair)(const REAL* nn_dist, const int32_t* nn_id, const REAL* spacings,
                           int64_t n_move, int64_t n_fixed, REAL* r, REAL* s, int64_t* idx_a,
                           int64_t* idx_b) {
    int64_t i = 0;
    for (int64_t t = 1; t < n_move; ++t)
        if (nn_dist[t] < nn_dist[i]) i = t;
    int64_t j = nn_id[i];
    int64_t ig = i + n_fixed;
    *r = nn_dist[i];
    *s = (spacings[ig] + spacings[j >= 0 ? j : ig]) / 2;
    *idx_a = ig < j ? ig : j;
    *idx_b = ig < j ? j : ig;
}

/* ---- the whole _relax! loop (src/repel.jl:243-334) with its stop rules ------------------
 * kick / trace / deposit! are host-side extras and stay out.  spacing is constant, a per-point
 * array (static during the loop), or — `refresh` != NULL — a callable of the movable points:
 *   refresh(positions n_move x dim, n_move, out n_move values, user)
 * evaluated like the reference does: at the current positions on every rebuild for the array the CV
 * monitor reads (src/repel.jl:251), and at p_old in EVERY sweep for the force / step (s = spacing(xi),
 * src/repel.jl:260).  `spacings` is then updated in place (n values, fixed head kept).
 * Returns the number of conv entries written (<= max_iters); p (n_move x dim) is updated in
 * place; *stop_reason: 0 max_iters, 1 tol, 2 cv_target (p reverted), 3 stall. */
typedef void (*FN(wtpo_spacing_cb))(const REAL* pos, int64_t n_move, REAL* out, void* user);
int FN(wtpo_relax_loop_cb)(REAL* snap, int64_t n, int64_t n_fixed, int dim, REAL* p,
                           REAL* spacings, int force_kind, REAL beta, REAL u0, REAL gamma, int k,
                           REAL alpha_lo, REAL alpha_max, int max_iters, double tol, int rebuild_every,
                           int stall_after, double cv_target, REAL* conv, int* stop_reason,
                           FN(wtpo_spacing_cb) refresh, void* user);
int FN(wtpo_relax_loop)(REAL* snap, int64_t n, int64_t n_fixed, int dim, REAL* p,
                        const REAL* spacings, int force_kind, REAL beta, REAL u0, REAL gamma, int k,
                        REAL alpha_lo, REAL alpha_max, int max_iters, double tol, int rebuild_every,
                        int stall_after, double cv_target, REAL* conv, int* stop_reason) {
    return FN(wtpo_relax_loop_cb)(snap, n, n_fixed, dim, p, (REAL*)spacings, force_kind, beta, u0, gamma, k, alpha_lo,
                                  alpha_max, max_iters, tol, rebuild_every, stall_after, cv_target, conv, stop_reason,
                                  NULL, NULL);
}
int FN(wtpo_relax_loop_cb)(REAL* snap, int64_t n, int64_t n_fixed, int dim, REAL* p,
                           REAL* spacings, int force_kind, REAL beta, REAL u0, REAL gamma, int k,
                           REAL alpha_lo, REAL alpha_max, int max_iters, double tol, int rebuild_every,
                           int stall_after, double cv_target, REAL* conv, int* stop_reason,
                           FN(wtpo_spacing_cb) refresh, void* user) {
    if (rebuild_every < 1) return -1; /* ArgumentError, repel.jl:74 */
    int64_t n_move = n - n_fixed;
    REAL* s_sweep = NULL; /* spacing(x_i) at p_old, every sweep (:260); the CV monitor keeps the rebuild-time array (:251) */
    if (refresh) {
        s_sweep = (REAL*)malloc(sizeof(REAL) * (size_t)(n + 1));
        memcpy(s_sweep, spacings, sizeof(REAL) * (size_t)n);
    }
    REAL* p_old = (REAL*)malloc(sizeof(REAL) * (size_t)(n_move * dim + 1));
    REAL* forces = (REAL*)malloc(sizeof(REAL) * (size_t)(n_move + 1));
    REAL* nn_dist = (REAL*)malloc(sizeof(REAL) * (size_t)(n_move + 1));
    int32_t* nn_id = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_move + 1));
    double best_cv = INFINITY; /* typemax(U) :238 (held in double: a REAL value converts exactly) */
    int last_impr = 0, nconv = 0, i = 1;
    *stop_reason = 0;
    while (i <= max_iters) {
        memcpy(p_old, p, sizeof(REAL) * (size_t)(n_move * dim)); /* :244 */
        if ((i - 1) % rebuild_every == 0) {                      /* :245-253 */
            memcpy(snap + n_fixed * dim, p, sizeof(REAL) * (size_t)(n_move * dim));
            if (refresh && i > 1) refresh(p, n_move, spacings + n_fixed, user); /* spacings at the current positions :251 */
        }
        if (refresh) refresh(p_old, n_move, s_sweep + n_fixed, user);          /* s = spacing(xi) :260 */
        FN(wtpo_relax_sweep)(snap, n, n_fixed, dim, p_old, refresh ? s_sweep : spacings, force_kind, beta, u0, gamma,
                             k, alpha_lo, alpha_max, p, forces, nn_dist, nn_id);
        REAL mx = 0; /* maximum(forces; init = 0) :293 */
        for (int64_t t = 0; t < n_move; ++t)
            if (forces[t] > mx) mx = forces[t];
        conv[nconv++] = mx;
        if ((stall_after > 0 || cv_target > 0) && n_move > 0) { /* :305-327 */
            double su = 0, su2 = 0;
            double cv = (double)FN(wtpo_dnn_cv)(nn_dist, spacings, n_move, n_fixed, &su, &su2);
            if (g_cv_double) { /* test switch (wtp_oracle.c): the rule on double sums, as the device evaluates it */
                const double mu = su / (double)n_move;
                double var = su2 / (double)n_move - mu * mu;
                var = var > 0 ? var : 0;
                cv = sqrt(var) / mu;
            }
            if (cv_target > 0 && (double)cv <= cv_target) {
                memcpy(p, p_old, sizeof(REAL) * (size_t)(n_move * dim)); /* :314 */
                *stop_reason = 2;
                break;
            }
            if (stall_after > 0) {
                if ((double)cv < (double)best_cv * (1 - 1.0e-3)) { /* :319 */
                    best_cv = cv;
                    last_impr = i;
                } else if (i - last_impr >= stall_after) {
                    *stop_reason = 3;
                    break;
                }
            }
        }
        if ((double)mx < tol) { /* :329-332 */
            *stop_reason = 1;
            break;
        }
        ++i;
    }
    free(p_old);
    free(forces);
    free(nn_dist);
    free(nn_id);
    free(s_sweep);
    return nconv;
}

/* Wall time of one kd-tree build (the serial part of every oracle iteration, like NearestNeighbors.jl's
 * KDTree(coords) at src/repel.jl:252): bench.py reports its share of the CPU baseline. */
double FN(wtpo_kd_build_seconds)(const REAL* xyz, int64_t n, int dim) {
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    FN(kdtree)* t = FN(kd_build)(xyz, n, dim);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    FN(kd_free)(t);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* _near_duplicate_keep_mask (src/repel.jl:565-580): greedy, order-preserving; the ball
 * search at ratio*max(spacings) sees every point inside the threshold, so a plain scan of
 * all pairs within thr gives the same mask. */
int FN(wtpo_cull_mask)(const REAL* xyz, int64_t n, int dim, const REAL* spacings, REAL ratio,
                       uint8_t* keep) {
    for (int64_t i = 0; i < n; ++i) keep[i] = 1;
    if (ratio <= 0 || n < 2) return 0;
    for (int64_t i = 0; i < n; ++i) {
        if (!keep[i]) continue;
        REAL thr = ratio * spacings[i];
        for (int64_t j = 0; j < n; ++j) {
            if (j == i || !keep[j]) continue;
            REAL d = SQRT(FN(d2)(xyz + j * dim, xyz + i * dim, dim));
            if (d < thr) keep[j] = 0;
        }
    }
    return 0;
}

/* ---- variable spacings (src/discretization/spacings.jl) ----------------------------------
 * _min_distance (:19-23) = 1-NN distance to the boundary points (brute force here).
 * LogLike (:67-72): h0*x/(a+x), a = h0*(1-(g-1)).  BoundaryLayerSpacing (:121-133):
 * h_w + (h_b-h_w)/(1+exp(-(d-δ/2)/(δ/6))). */
static REAL FN(min_dist)(const REAL* q, const REAL* bnd, int64_t nb, int dim) {
    REAL best = REAL_MAX;
    for (int64_t j = 0; j < nb; ++j) {
        REAL d = FN(d2)(q, bnd + j * dim, dim);
        if (d < best) best = d;
    }
    return SQRT(best);
}

void FN(wtpo_spacing_loglike)(const REAL* xyz, int64_t n, int dim, const REAL* bnd, int64_t nb,
                              REAL base_size, REAL growth_rate, REAL* out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        REAL x = FN(min_dist)(xyz + i * dim, bnd, nb, dim);
        REAL inv_growth = 1 - (growth_rate - 1);
        REAL a = base_size * inv_growth;
        out[i] = base_size * x / (a + x);
    }
}

void FN(wtpo_spacing_boundary_layer)(const REAL* xyz, int64_t n, int dim, const REAL* bnd,
                                     int64_t nb, REAL at_wall, REAL bulk, REAL thickness,
                                     REAL* out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        REAL d = FN(min_dist)(xyz + i * dim, bnd, nb, dim);
        REAL center = thickness / 2;
        REAL width = thickness / 6;
        REAL sig = 1 / (1 + EXP(-(d - center) / width));
        out[i] = at_wall + (bulk - at_wall) * sig;
    }
}


/* ---- isinside: the post-filter of the volume-only repel (src/repel.jl:90) --------------------
 * 3-D (src/isinside.jl:86-106): g = sum over boundary elements of
 *   ((area * dist) . normal) / norm(dist)^3,  dist = testpoint - point,
 * inside iff g < -2*pi (the comparison promotes g to Float64).  A test point coincident with an
 * element gives 0/0 = NaN, NaN < x is false: outside, as in the reference.  The reference reduces
 * with tmapreduce (order unspecified); here the sum is sequential, in T. */
void FN(wtpo_isinside_greens)(const REAL* test, int64_t n, const REAL* pts, const REAL* normals,
                              const REAL* areas, int64_t m, REAL* g_out, uint8_t* inside) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const REAL* x = test + 3 * i;
        REAL g = 0;
        for (int64_t j = 0; j < m; ++j) {
            REAL dx = x[0] - pts[3 * j], dy = x[1] - pts[3 * j + 1], dz = x[2] - pts[3 * j + 2];
            REAL a = areas[j];
            REAL dot = ((a * dx) * normals[3 * j] + (a * dy) * normals[3 * j + 1]) + (a * dz) * normals[3 * j + 2];
            REAL r = SQRT((dx * dx + dy * dy) + dz * dz);
            g = g + dot / ((r * r) * r);
        }
        if (g_out) g_out[i] = g;
        inside[i] = ((double)g < -2.0 * 3.14159265358979323846) ? 1 : 0;
    }
}

/* 2-D (src/isinside.jl:17-33): true when the test point coincides with a polygon point
 * (r < 100 eps); else the sum of the signed angles ∠(p_i, x, p_{i+1}) around the closed,
 * ordered polygon (Meshes' 2-D ∠(u, v) = atan(u x v, u . v)); inside iff |sum| >= 1e3 eps(T). */
void FN(wtpo_isinside_winding)(const REAL* test, int64_t n, const REAL* poly, int64_t m, REAL* sum_out,
                               uint8_t* inside) {
    const REAL eps = REAL_EPS;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const REAL x = test[2 * i], y = test[2 * i + 1];
        int coincident = 0;
        REAL sum = 0;
        for (int64_t j = 0; j < m; ++j) {
            const int64_t jn = j + 1 < m ? j + 1 : 0;
            REAL ux = poly[2 * j] - x, uy = poly[2 * j + 1] - y;
            REAL vx = poly[2 * jn] - x, vy = poly[2 * jn + 1] - y;
            if (SQRT(ux * ux + uy * uy) < (REAL)1.0e2 * eps) coincident = 1;
            sum = sum + ATAN2(ux * vy - uy * vx, ux * vx + uy * vy);
        }
        if (sum_out) sum_out[i] = sum;
        REAL as = sum < 0 ? -sum : sum;
        inside[i] = coincident ? 1 : (as < (REAL)1.0e3 * eps ? 0 : 1);
    }
}


/* ======================================================================================
 * Triangle-mesh geometry index: the queries the octree repel method makes
 * (src/repel.jl:122-181 wall rule :448-469, projection :522-537).
 *
 *   closest point + feature   src/octree/geometric_utils.jl:68-136 (Ericson region tests)
 *   face normals, angle-weighted edge / vertex pseudonormals keyed by exact coordinates
 *                             src/octree/triangle_octree.jl:221-277, :297-311
 *   nearest triangle          :532-553 (strict d2 < best; which of two equidistant triangles wins
 *                             depends on the octree traversal there: parity unpinned.  Canonical
 *                             rule here: smallest (d2, triangle index), by brute force)
 *   signed distance           :583-607 (sign of (p - closest) . pseudonormal of the closest feature)
 *   isinside                  :71-99  (mesh bbox test, then sd < 0; the leaf-class cache of the
 *                             reference is an accelerator for the same predicate)
 *   project_to_boundary       src/repel.jl:522-537 (closest point - offset * face normal)
 * Vector arithmetic is written out left to right (dot = (a1 b1 + a2 b2) + a3 b3).
 * ====================================================================================== */
static inline REAL FN(dot3)(const REAL* a, const REAL* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

/* returns the feature code 0 face, 1..3 vertex, 4 e12, 5 e13, 6 e23 */
static int FN(tri_closest)(const REAL* p, const REAL* a, const REAL* b, const REAL* c, REAL* out) {
    REAL ab[3], ac[3], ap[3], bp[3], cp[3];
    for (int i = 0; i < 3; ++i) { ab[i] = b[i] - a[i]; ac[i] = c[i] - a[i]; ap[i] = p[i] - a[i]; }
    const REAL d1 = FN(dot3)(ab, ap), d2 = FN(dot3)(ac, ap);
    if (d1 <= 0 && d2 <= 0) { for (int i = 0; i < 3; ++i) out[i] = a[i]; return 1; }
    for (int i = 0; i < 3; ++i) bp[i] = p[i] - b[i];
    const REAL d3 = FN(dot3)(ab, bp), d4 = FN(dot3)(ac, bp);
    if (d3 >= 0 && d4 <= d3) { for (int i = 0; i < 3; ++i) out[i] = b[i]; return 2; }
    const REAL vc = d1 * d4 - d3 * d2;
    if (vc <= 0 && d1 >= 0 && d3 <= 0) {
        const REAL v = d1 / (d1 - d3);
        for (int i = 0; i < 3; ++i) out[i] = a[i] + v * ab[i];
        return 4;
    }
    for (int i = 0; i < 3; ++i) cp[i] = p[i] - c[i];
    const REAL d5 = FN(dot3)(ab, cp), d6 = FN(dot3)(ac, cp);
    if (d6 >= 0 && d5 <= d6) { for (int i = 0; i < 3; ++i) out[i] = c[i]; return 3; }
    const REAL vb = d5 * d2 - d1 * d6;
    if (vb <= 0 && d2 >= 0 && d6 <= 0) {
        const REAL w = d2 / (d2 - d6);
        for (int i = 0; i < 3; ++i) out[i] = a[i] + w * ac[i];
        return 5;
    }
    const REAL va = d3 * d6 - d5 * d4;
    if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) {
        const REAL w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        for (int i = 0; i < 3; ++i) out[i] = b[i] + w * (c[i] - b[i]);
        return 6;
    }
    const REAL denom = (REAL)1 / ((va + vb) + vc);
    const REAL v = vb * denom, w = vc * denom;
    for (int i = 0; i < 3; ++i) out[i] = (a[i] + ab[i] * v) + ac[i] * w;
    return 0;
}

void FN(wtpo_tri_closest)(const REAL* p, const REAL* a, const REAL* b, const REAL* c, REAL* out, int32_t* feature) {
    *feature = FN(tri_closest)(p, a, b, c, out);
}

typedef struct {
    REAL key[6];   /* vertex: 3 coords (+3 zero); edge: the lexicographically smaller end first */
    int64_t order; /* triangle * 3 + slot: accumulation order of the reference's loops */
    REAL add[3];
} FN(PnTerm);

static int FN(pn_cmp)(const void* pa, const void* pb) {
    const FN(PnTerm)* a = (const FN(PnTerm)*)pa;
    const FN(PnTerm)* b = (const FN(PnTerm)*)pb;
    const int c = memcmp(a->key, b->key, sizeof(a->key)); /* exact-coordinate identity (bit patterns) */
    if (c) return c;
    return a->order < b->order ? -1 : (a->order > b->order ? 1 : 0);
}

static int FN(lex_less)(const REAL* a, const REAL* b) { /* SVector isless: lexicographic */
    for (int i = 0; i < 3; ++i) {
        if (a[i] < b[i]) return 1;
        if (a[i] > b[i]) return 0;
    }
    return 0;
}

static REAL FN(corner_angle)(const REAL* vc, const REAL* va, const REAL* vb) {
    REAL u[3], w[3];
    for (int i = 0; i < 3; ++i) { u[i] = va[i] - vc[i]; w[i] = vb[i] - vc[i]; }
    const REAL den = SQRT(FN(dot3)(u, u) * FN(dot3)(w, w));
    if (den < REAL_EPS) return 0;
    REAL c = FN(dot3)(u, w) / den;
    c = c < (REAL)-1 ? (REAL)-1 : (c > (REAL)1 ? (REAL)1 : c);
    return ACOS(c);
}

/* pn: nt x 7 x 3 = {face, vertex 1..3, edge 12, 13, 23} pseudonormals per triangle (not normalised:
 * only their sign against (p - closest) is used, as in the reference). */
void FN(wtpo_mesh_pseudonormals)(const REAL* verts, const int32_t* tris, int64_t nt, REAL* pn) {
    FN(PnTerm)* vt = (FN(PnTerm)*)calloc((size_t)(3 * nt > 0 ? 3 * nt : 1), sizeof(FN(PnTerm)));
    FN(PnTerm)* et = (FN(PnTerm)*)calloc((size_t)(3 * nt > 0 ? 3 * nt : 1), sizeof(FN(PnTerm)));
    for (int64_t t = 0; t < nt; ++t) {
        const REAL* v[3] = {verts + 3 * tris[3 * t], verts + 3 * tris[3 * t + 1], verts + 3 * tris[3 * t + 2]};
        REAL e1[3], e2[3], nr[3];
        for (int i = 0; i < 3; ++i) { e1[i] = v[1][i] - v[0][i]; e2[i] = v[2][i] - v[0][i]; }
        nr[0] = e1[1] * e2[2] - e1[2] * e2[1];
        nr[1] = e1[2] * e2[0] - e1[0] * e2[2];
        nr[2] = e1[0] * e2[1] - e1[1] * e2[0];
        const REAL mag = SQRT(FN(dot3)(nr, nr));
        REAL* f = pn + 21 * t;
        for (int i = 0; i < 3; ++i) f[i] = mag < REAL_EPS * 100 ? (REAL)0 : nr[i] / mag;
        static const int ea[3] = {0, 1, 2}, eb[3] = {1, 2, 0}; /* (v1,v2), (v2,v3), (v3,v1) */
        for (int s = 0; s < 3; ++s) {
            FN(PnTerm)* e = et + 3 * t + s;
            const REAL* a = v[ea[s]];
            const REAL* b = v[eb[s]];
            const int ab = FN(lex_less)(a, b);
            for (int i = 0; i < 3; ++i) { e->key[i] = ab ? a[i] : b[i]; e->key[3 + i] = ab ? b[i] : a[i]; e->add[i] = f[i]; }
            e->order = 3 * t + s;
            FN(PnTerm)* q = vt + 3 * t + s; /* corner s with its two neighbours in cyclic order */
            const REAL ang = FN(corner_angle)(v[s], v[(s + 1) % 3], v[(s + 2) % 3]);
            for (int i = 0; i < 3; ++i) { q->key[i] = v[s][i]; q->key[3 + i] = 0; q->add[i] = ang * f[i]; }
            q->order = 3 * t + s;
        }
    }
    for (int pass = 0; pass < 2; ++pass) {
        FN(PnTerm)* terms = pass ? et : vt;
        qsort(terms, (size_t)(3 * nt), sizeof(FN(PnTerm)), FN(pn_cmp));
        int64_t i = 0;
        while (i < 3 * nt) {
            int64_t j = i;
            REAL sum[3] = {0, 0, 0};
            while (j < 3 * nt && memcmp(terms[j].key, terms[i].key, sizeof(terms[i].key)) == 0) {
                for (int c = 0; c < 3; ++c) sum[c] = sum[c] + terms[j].add[c];
                ++j;
            }
            for (int64_t q = i; q < j; ++q) {
                const int64_t t = terms[q].order / 3;
                const int s = (int)(terms[q].order % 3);
                /* vertex slot s -> feature 1 + s; edge slots (12),(23),(31) -> features 4, 6, 5 */
                const int feat = pass ? (s == 0 ? 4 : (s == 1 ? 6 : 5)) : 1 + s;
                for (int c = 0; c < 3; ++c) pn[21 * t + 3 * feat + c] = sum[c];
            }
            i = j;
        }
    }
    free(vt);
    free(et);
}

/* Brute force over every triangle: canonical minimum of (d2, triangle index). */
void FN(wtpo_mesh_nearest)(const REAL* verts, const int32_t* tris, int64_t nt, const REAL* pts, int64_t n,
                           REAL* d2_out, int32_t* tri_out, REAL* cp_out, int32_t* feat_out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const REAL* p = pts + 3 * i;
        REAL best = REAL_MAX, bc[3] = {p[0], p[1], p[2]};
        int32_t bt = -1, bf = 0;
        for (int64_t t = 0; t < nt; ++t) {
            REAL cp[3], dv[3];
            const int f = FN(tri_closest)(p, verts + 3 * tris[3 * t], verts + 3 * tris[3 * t + 1],
                                          verts + 3 * tris[3 * t + 2], cp);
            for (int c = 0; c < 3; ++c) dv[c] = p[c] - cp[c];
            const REAL d2 = FN(dot3)(dv, dv);
            if (d2 < best) { best = d2; bt = (int32_t)t; bf = f; bc[0] = cp[0]; bc[1] = cp[1]; bc[2] = cp[2]; }
        }
        d2_out[i] = best;
        tri_out[i] = bt;
        feat_out[i] = bf;
        for (int c = 0; c < 3; ++c) cp_out[3 * i + c] = bc[c];
    }
}

/* sd (src/octree/triangle_octree.jl:583-607), inside (:71-99), projection (src/repel.jl:522-537)
 * from the nearest-triangle data; bbox = {min xyz, max xyz} of the vertices (:279-291). */
void FN(wtpo_mesh_classify)(const REAL* pts, int64_t n, const REAL* d2, const int32_t* tri, const REAL* cp,
                            const int32_t* feat, const REAL* pn, const REAL* bbox, REAL offset, REAL* sd_out,
                            uint8_t* inside_out, REAL* proj_out) {
    for (int64_t i = 0; i < n; ++i) {
        const REAL* p = pts + 3 * i;
        if (tri[i] < 0) {
            sd_out[i] = REAL_MAX;
            inside_out[i] = 0;
            for (int c = 0; c < 3; ++c) proj_out[3 * i + c] = p[c];
            continue;
        }
        const REAL* nrm = pn + 21 * (int64_t)tri[i] + 3 * feat[i];
        REAL dv[3];
        for (int c = 0; c < 3; ++c) dv[c] = p[c] - cp[3 * i + c];
        const REAL s = FN(dot3)(dv, nrm);
        const REAL dist = SQRT(d2[i]);
        const REAL sd = s < 0 ? -dist : (s > 0 ? dist : (REAL)0);
        sd_out[i] = sd;
        int out_of_box = 0;
        for (int c = 0; c < 3; ++c) out_of_box |= (p[c] < bbox[c]) || (p[c] > bbox[3 + c]);
        /* classify_point with tol 0: |sd| <= 0 is BOUNDARY, sd < 0 INTERIOR */
        const REAL asd = sd < 0 ? -sd : sd;
        inside_out[i] = (!out_of_box && !(asd <= 0) && sd < 0) ? 1 : 0;
        const REAL* f = pn + 21 * (int64_t)tri[i];
        for (int c = 0; c < 3; ++c) proj_out[3 * i + c] = cp[3 * i + c] - offset * f[c];
    }
}


/* ======================================================================================
 * Consumers of the k-NN rows.
 *   _compute_normal    src/normals.jl:65-69: eigen(Symmetric(cov(v))) -> Q[:, 1], the eigenvector of the
 *                      smallest eigenvalue of the covariance of the k nearest points (self included).
 *                      LAPACK's eigen() is not in the reference tree; restated with a cyclic Jacobi
 *                      iteration (any symmetric eigen solver gives the same vector up to sign and
 *                      rounding).  Sign: unpinned in the reference (orient_normals! fixes it later);
 *                      canonical here: the component of largest magnitude is positive.
 *   _gradient_limit_field   src/discretization/algorithms/octree.jl:677-717
 * ====================================================================================== */
static void FN(jacobi_rot)(REAL* app, REAL* aqq, REAL* apq, REAL* arp, REAL* arq, REAL* vp, REAL* vq) {
    if (*apq == 0) return;
    const REAL theta = (*aqq - *app) / ((REAL)2 * *apq);
    const REAL at = theta < 0 ? -theta : theta;
    REAL t = (REAL)1 / (at + SQRT(theta * theta + (REAL)1));
    t = theta < 0 ? -t : t;
    const REAL c = (REAL)1 / SQRT(t * t + (REAL)1), sn = t * c;
    *app = *app - t * *apq;
    *aqq = *aqq + t * *apq;
    *apq = 0;
    const REAL rp = *arp, rq = *arq;
    *arp = c * rp - sn * rq;
    *arq = sn * rp + c * rq;
    for (int i = 0; i < 3; ++i) {
        const REAL a = vp[i], b = vq[i];
        vp[i] = c * a - sn * b;
        vq[i] = sn * a + c * b;
    }
}

void FN(wtpo_pca_normals)(const REAL* xyz, int64_t n, int dim, const int32_t* rows, int k, REAL* out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const int32_t* row = rows + i * k;
        REAL m[3] = {0, 0, 0};
        for (int j = 0; j < k; ++j)
            for (int c = 0; c < dim; ++c) m[c] = m[c] + xyz[(int64_t)row[j] * dim + c];
        const REAL inv = (REAL)1 / (REAL)k;
        for (int c = 0; c < 3; ++c) m[c] = m[c] * inv;
        REAL xx = 0, xy = 0, xz = 0, yy = 0, yz = 0, zz = 0;
        for (int j = 0; j < k; ++j) {
            const REAL* p = xyz + (int64_t)row[j] * dim;
            const REAL dx = p[0] - m[0], dy = p[1] - m[1], dz = dim == 3 ? p[2] - m[2] : (REAL)0;
            xx = xx + dx * dx, xy = xy + dx * dy, xz = xz + dx * dz;
            yy = yy + dy * dy, yz = yz + dy * dz, zz = zz + dz * dz;
        }
        REAL vx[3] = {1, 0, 0}, vy[3] = {0, 1, 0}, vz[3] = {0, 0, 1};
        for (int sweep = 0; sweep < 8; ++sweep) {
            FN(jacobi_rot)(&xx, &yy, &xy, &xz, &yz, vx, vy);
            if (dim == 3) {
                FN(jacobi_rot)(&xx, &zz, &xz, &xy, &yz, vx, vz);
                FN(jacobi_rot)(&yy, &zz, &yz, &xy, &xz, vy, vz);
            }
        }
        const REAL* v = vx;
        REAL lam = xx;
        if (yy < lam) { lam = yy; v = vy; }
        if (dim == 3 && zz < lam) { lam = zz; v = vz; }
        int big = 0;
        REAL ab = v[0] < 0 ? -v[0] : v[0];
        for (int c = 1; c < dim; ++c) {
            const REAL a = v[c] < 0 ? -v[c] : v[c];
            if (a > ab) { ab = a; big = c; }
        }
        const REAL sg = v[big] < 0 ? (REAL)-1 : (REAL)1;
        for (int c = 0; c < dim; ++c) out[i * dim + c] = sg * v[c];
    }
}

/* rows / dist: n x k (self included); returns the sweeps applied */
int FN(wtpo_gradient_limit)(const int32_t* rows, const REAL* dist, int64_t n, int k, const REAL* h0, REAL g, double tol,
                            int max_sweeps, REAL* out) {
    REAL* h = (REAL*)malloc(sizeof(REAL) * (size_t)(n > 0 ? n : 1));
    REAL* hn = (REAL*)malloc(sizeof(REAL) * (size_t)(n > 0 ? n : 1));
    memcpy(h, h0, sizeof(REAL) * (size_t)n);
    int sweeps = 0;
    for (int s = 0; s < max_sweeps; ++s) {
#pragma omp parallel for schedule(static)
        for (int64_t a = 0; a < n; ++a) {
            REAL hi = h[a];
            for (int t = 0; t < k; ++t) {
                const REAL cand = h[rows[a * k + t]] + g * dist[a * k + t];
                if (cand < hi) hi = cand;
            }
            hn[a] = hi;
        }
        REAL maxrel = 0;
        for (int64_t a = 0; a < n; ++a) {
            const REAL d = hn[a] - h[a];
            const REAL rel = (d < 0 ? -d : d) / h[a];
            if (rel > maxrel) maxrel = rel;
        }
        memcpy(h, hn, sizeof(REAL) * (size_t)n);
        ++sweeps;
        if ((double)maxrel < tol) break;
    }
    memcpy(out, h, sizeof(REAL) * (size_t)n);
    free(h);
    free(hn);
    return sweeps;
}

#undef KD_LEAF
#undef KD_SLACK
#undef FN
#undef CAT
#undef CAT_
