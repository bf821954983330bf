/*
 * wtp.h — C ABI of libwtp, the MI355X (gfx950) neighbour/stencil engine behind
 * WhatsThePoint.jl's `set_topology` / `repel` hot path.
 *
 * Every entry point below replaces the body of one reference function (citations are
 * relative to the reference checkout, JuliaMeshless/WhatsThePoint.jl v0.3.1).  The
 * reference has no FFI seam of its own: the seam is the set of Julia functions whose
 * bodies a maintainer swaps for `ccall`s (INTEGRATION.md shows the Julia side).
 *
 * Conventions
 *  - plain C, plain pointers and sizes; no exceptions or signals cross the ABI;
 *  - every function returns a wtp_status (0 = OK); wtp_last_error(ctx) gives the text;
 *  - host entry points (`xyz`, `idx_out`, ...) take HOST pointers, block until the
 *    result is on the host; `_dev` entry points take DEVICE pointers on the context's
 *    GPU and only enqueue + synchronise the context's stream;
 *  - coordinates are AoS `n x dim` contiguous (the in-memory layout of Julia's
 *    Vector{SVector{D,T}} / Vector{Point}, src/repel.jl:216), dim in {2,3};
 *  - indices are int32, 0-based, in the caller's (snapshot-global) numbering:
 *    boundary points first, then volume (src/cloud.jl:235-237);
 *  - canonical neighbour order: ascending (d2, index) with
 *    d2 = ((dx*dx + dy*dy) + dz*dz) evaluated in the cloud's float type, no FMA
 *    contraction; distances returned are sqrt(d2) in that type;
 *  - a context is not thread-safe (one caller at a time); several contexts may coexist.
 */
#ifndef WTP_H
#define WTP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wtp_ctx wtp_ctx;

typedef enum wtp_status {
    WTP_OK = 0,
    WTP_ERR_ARG = 1,      /* bad argument: the Julia shim maps it to ArgumentError (src/repel.jl:74) */
    WTP_ERR_OOM = 2,      /* device or host allocation failed */
    WTP_ERR_HIP = 3,      /* HIP runtime / kernel error */
    WTP_ERR_STATE = 4,    /* call out of order (e.g. relax_step before relax_init) */
    WTP_ERR_NO_DEVICE = 5 /* no usable gfx950 device: the library never falls back to a CPU path */
} wtp_status;

typedef enum wtp_dtype { WTP_F32 = 0, WTP_F64 = 1 } wtp_dtype;

/* Force laws of src/repel_forces.jl:37,57-60,96-100,124-127 (u = r/s). */
typedef enum wtp_force_kind {
    WTP_FORCE_INVERSE_DISTANCE = 0,    /* 1/(u^2+beta)^2                       */
    WTP_FORCE_SPACING_EQUILIBRIUM = 1, /* (1-u^2)/(u^2+beta)^2                 */
    WTP_FORCE_CLIPPED_SPACING = 2,     /* max((u0^2-u^2)/(u^2+beta)^2, 0)  (default) */
    WTP_FORCE_STRONG_SPACING = 3       /* (1-u^2)/(u^2+beta)^gamma             */
} wtp_force_kind;

typedef struct wtp_force_desc {
    int32_t kind;  /* wtp_force_kind */
    double beta;   /* softening, > 0 */
    double u0;     /* support radius (clipped law only) */
    double gamma;  /* core strength (strong law only) */
} wtp_force_desc;

/* Spacing callables of src/discretization/spacings.jl invoked inside the sweep
 * (src/repel.jl:209,251,260). */
typedef enum wtp_spacing_kind {
    WTP_SPACING_CONSTANT = 0,  /* ConstantSpacing: spacings.jl:35-39 */
    WTP_SPACING_PER_POINT = 1, /* host-evaluated s[n] in snapshot order (any callable); refresh with
                                  wtp_relax_set_spacing */
    WTP_SPACING_LOGLIKE = 2,   /* LogLike: spacings.jl:67-72, p0 = base_size, p1 = growth_rate       */
    WTP_SPACING_BOUNDARY_LAYER = 3 /* BoundaryLayerSpacing: spacings.jl:121-133, p0 = at_wall,
                                  p1 = bulk, p2 = layer_thickness                                 */
} wtp_spacing_kind;

typedef struct wtp_spacing_desc {
    int32_t kind;            /* wtp_spacing_kind */
    double constant;         /* CONSTANT: the spacing (unitless, as ustrip gives it) */
    const void* per_point;   /* PER_POINT: host array of n values of the cloud's dtype */
    double p0, p1, p2;       /* LOGLIKE / BOUNDARY_LAYER parameters (unitless)          */
    const void* boundary_xyz;/* LOGLIKE / BOUNDARY_LAYER: host array, n_boundary x dim of the cloud's dtype:
                                the points the law measures its distance to (spacings.jl:17-28); they
                                are evaluated on the device at every point's current position, every
                                sweep (src/repel.jl:251,260) */
    int64_t n_boundary;
} wtp_spacing_desc;

/* Scalars the host-side stop logic of src/repel.jl:293-334 needs after one sweep. */
typedef struct wtp_step_stats {
    double max_force;  /* maximum(forces), forces[id] = |F|*s         (repel.jl:283,293) */
    double sum_u;      /* sum_i nn_dist[i]/spacings[i+n_fixed]        (repel.jl:374-386) */
    double sum_u2;     /* sum_i (nn_dist[i]/spacings[i+n_fixed])^2                        */
    int64_t n_move;    /* number of movable points the sums run over                     */
    int64_t argmin_i;  /* closest pair (repel.jl:396-403): snapshot-global index of the  */
    int64_t argmin_j;  /*   movable point with the smallest nn_dist, and its neighbour;  */
    double argmin_r;   /*   -1/-1/inf when n_move == 0                                    */
    int64_t n_fallback;/* queries that left the 27-cell fast path (diagnostic)           */
    int64_t n_uncovered;/* sharded sessions: queries whose neighbourhood reaches past the
                          covered range (wtp_relax_set_coverage); 0 when unlimited        */
    int64_t n_escaped; /* octree method: volume points the wall rule sent back this sweep
                          (repel.jl:466-467); 0 without wtp_relax_set_wall                */
} wtp_step_stats;

/* ---- lifetime ------------------------------------------------------------------ */

/* Create a context on one GPU (n_dev must be 1: multi-GPU runs use one context per
 * process/GPU, sharded by the host driver — SURVEY.md §8e). */
int wtp_create(const int* device_ordinals, int n_dev, wtp_ctx** out);
int wtp_destroy(wtp_ctx* ctx);
/* UTF-8 text of the last non-zero status on this context ("" if none).  ctx may be
 * NULL: then the text of the last failed wtp_create on this thread. */
const char* wtp_last_error(const wtp_ctx* ctx);
/* Library version string, for the binding's sanity check. */
const char* wtp_version(void);

/* ---- KNNTopology / KNearestSearch ------------------------------------------------ */

/* Replaces _build_knn_neighbors (src/topology.jl:79-84) and the KNearestSearch +
 * search/searchdists wrappers (src/neighbors.jl:1-21).
 * include_self = 0: row i = the k nearest OTHER points (self removed by index), as
 *                   set_topology stores them (k+1 query, first hit dropped);
 * include_self = 1: row i = the k nearest points including i itself (raw `search`
 *                   result: self first, test/neighbors.jl:54-56).
 * idx_out: n*k int32 (row-major); dist_out: n*k values of dtype, or NULL.
 * Errors: k < 1, or k > n - (include_self ? 0 : 1)  (the reference's kd-tree throws
 * for k+1 > n), dim not in {2,3}, n < 1. */
int wtp_knn(wtp_ctx* ctx, const void* xyz, int64_t n, int dim, int dtype, int k,
            int include_self, int32_t* idx_out, void* dist_out);
/* Same, device-resident input/output (bench and the sharded driver use this). */
int wtp_knn_dev(wtp_ctx* ctx, const void* d_xyz, int64_t n, int dim, int dtype, int k,
                int include_self, int32_t* d_idx_out, void* d_dist_out);

/* ---- RadiusTopology / BallSearch --------------------------------------------------- */

/* Replaces _build_radius_neighbors (src/topology.jl:91-97): all j != i with
 * d2(i,j) <= r*r (inclusive).  Two-phase CSR so the caller allocates exactly:
 * count fills counts_out[n]; the caller builds offsets[n+1] (exclusive scan, int64)
 * and calls fill, which writes rows sorted by ascending (d2, index) into idx_out.
 * fill refers to the cloud passed to the preceding count on the same context. */
int wtp_radius_count(wtp_ctx* ctx, const void* xyz, int64_t n, int dim, int dtype,
                     double r, int32_t* counts_out);
int wtp_radius_fill(wtp_ctx* ctx, const int64_t* offsets, int32_t* idx_out);
/* The same first phase with the exclusive scan done on the device: offsets_out[n+1] (int64) is what the
 * caller needs to allocate idx_out (offsets_out[n] entries); the offsets also stay resident, so the fill
 * that follows may pass offsets = NULL.  Saves the counts' trip to the host and the offsets' trip back.  */
int wtp_radius_offsets(wtp_ctx* ctx, const void* xyz, int64_t n, int dim, int dtype, double r,
                       int64_t* offsets_out);

/* ---- repel: _relax! ------------------------------------------------------------------ */

/* Replaces the setup of _relax! (src/repel.jl:207-241).  snap_xyz: the search snapshot,
 * n points, the first n_fixed static (the boundary wall), the tail movable.
 * k is the reference's `k` (self slot included: kk = min(k, n), repel.jl:208,259).
 * alpha_lo/alpha_max: step bounds (repel.jl:85,226). */
int wtp_relax_init(wtp_ctx* ctx, const void* snap_xyz, int64_t n, int64_t n_fixed, int dim,
                   int dtype, const wtp_spacing_desc* spacing, const wtp_force_desc* force,
                   int k, double alpha_lo, double alpha_max);

/* Same with the snapshot already on the context's GPU (spacing->per_point stays a host array). */
int wtp_relax_init_dev(wtp_ctx* ctx, const void* d_snap_xyz, int64_t n, int64_t n_fixed, int dim,
                       int dtype, const wtp_spacing_desc* spacing, const wtp_force_desc* force,
                       int k, double alpha_lo, double alpha_max);

/* One pass of src/repel.jl:244-293 plus the reductions of :293,374-403.
 * rebuild != 0 refreshes the search snapshot from the current positions first
 * (repel.jl:245-253); rebuild == 0 sweeps against the stale snapshot. */
int wtp_relax_step(wtp_ctx* ctx, int rebuild, wtp_step_stats* stats);

/* n_iters passes with no host round trip of coordinates; iteration i (0-based)
 * rebuilds iff i % rebuild_every == 0.  conv_out[n_iters] receives max_force per
 * iteration (may be NULL); last receives the final iteration's stats (may be NULL). */
int wtp_relax_run(wtp_ctx* ctx, int n_iters, int rebuild_every, double* conv_out,
                  wtp_step_stats* last);

/* The loop of `_relax!` WITH its stop rules (src/repel.jl:305-334), evaluated on the device after every sweep in the
 * reference's order: cv_target (positions are then reverted to p_old, :314), stall_after on the CV of d_NN / s
 * (:316-326), tol on max |F| s (:329-332).  Sweeps are enqueued in batches of 16 without any host round trip; once a
 * rule fires the rest of the batch does nothing, so the session ends in the state of the last iteration that ran.
 * The CV is evaluated from sums of u = d_NN / s and u^2 accumulated in DOUBLE (u itself is rounded in the cloud's
 * type).  The reference's `_dnn_cv` (src/repel.jl:374-386) sums serially in the cloud's type T: for Float64 clouds the
 * two agree to rounding and the loop stops at the reference's iteration (tests/test_gpu_stop_rules.py: same count,
 * reason and positions as the oracle's loop); for Float32 clouds a serial Float32 sum over >= 10^4 points carries
 * 1e-4 .. 1e-3 relative error in the variance — the size of the stall rule's own 1e-3 margin — so the reference's
 * stall / cv_target rule can fire at ANOTHER iteration than this one (which uses the better number).  tol is
 * unaffected (a maximum).  conv_out[0 .. *n_done) = max |F| s per sweep; *reason: 0 max_iters reached, 1 tol,
 * 2 cv_target, 3 stall.
 * Not for sessions with the octree wall rule (wtp_relax_set_wall), kicks, traces or a caller-evaluated spacing:
 * those need the host between two sweeps (wtp_relax_step).  */
int wtp_relax_run_until(wtp_ctx* ctx, int max_iters, int rebuild_every, double tol, int stall_after, double cv_target,
                        double* conv_out, int* n_done, int* reason, wtp_step_stats* last);

/* Current movable points, (n - n_fixed) x dim of dtype, snapshot order. */
int wtp_relax_get(wtp_ctx* ctx, void* xyz_out);
/* Same into device memory on the context's GPU (sharded driver: no host round trip). */
int wtp_relax_get_dev(wtp_ctx* ctx, void* d_xyz_out);
/* Per-point outputs of the last sweep, each n - n_fixed long (any may be NULL):
 * forces (|F|*s), nn_dist (dtype), nn_id (int32 snapshot-global, -1 if none). */
int wtp_relax_get_point_data(wtp_ctx* ctx, void* forces_out, void* nn_dist_out,
                             int32_t* nn_id_out);
/* Overwrite movable point i (0-based within the movable tail): _maybe_kick!'s write
 * (src/repel.jl:431). */
int wtp_relax_set(wtp_ctx* ctx, int64_t i, const void* xyz);
/* p .= p_old (src/repel.jl:314): undo the last sweep. */
int wtp_relax_revert(wtp_ctx* ctx);
/* wtp_relax_set for m movable points in one pass: idx strictly increasing, xyz m x dim of the session's
 * dtype (the deposition pass of the octree method may land thousands of points in one iteration).  */
int wtp_relax_set_batch(wtp_ctx* ctx, const int64_t* idx, const void* xyz, int64_t m);
/* Refresh the PER_POINT spacing array (n values, snapshot order; repel.jl:251). */
int wtp_relax_set_spacing(wtp_ctx* ctx, const void* spacing);
/* Release the relax state (device buffers stay pooled in the context). */
int wtp_relax_end(wtp_ctx* ctx);

/* The spacings the session currently holds, one value per snapshot point in snapshot order
 * (`spacings` of src/repel.jl:209,251 — the kick and the trace read it).  Host array of n values. */
int wtp_relax_get_spacing(wtp_ctx* ctx, void* spacing_out);

/* A variable spacing law evaluated at arbitrary points (spacing.(points), e.g. the default
 * alpha = minimum(spacing.(to(cloud)))/20 of src/repel.jl:61, or the metrics): LOGLIKE /
 * BOUNDARY_LAYER descriptors only.  xyz: host n x dim of dtype, out: host n values.  */
int wtp_spacing_eval(wtp_ctx* ctx, const wtp_spacing_desc* spacing, const void* xyz, int64_t n, int dim,
                     int dtype, void* out);

/* ---- isinside: the post-filter of the volume-only repel (src/repel.jl:90) -----------------
 * 3-D, replaces isinside(testpoint, cloud|boundary) + _greens (src/isinside.jl:86-106) for a whole
 * array of test points: g_i = sum_j ((area_j (x_i - p_j)) . n_j) / |x_i - p_j|^3 over the m boundary
 * elements (centroid p, unit normal n, area), inside_out[i] = (g_i < -2 pi); a test point that
 * coincides with an element gives NaN and is reported outside, as in the reference.  Host arrays:
 * test_xyz n x 3, elem_xyz / elem_normal m x 3, elem_area m, all of dtype; g_out n values or NULL.
 * g is evaluated with fused multiply-adds and a 1-ulp rsqrt: it agrees with a term-by-term
 * evaluation to ~1e-6 relative, so only points with |g + 2 pi| below that can flip.  */
int wtp_isinside_greens(wtp_ctx* ctx, const void* test_xyz, int64_t n, const void* elem_xyz,
                        const void* elem_normal, const void* elem_area, int64_t m, int dtype,
                        uint8_t* inside_out, void* g_out);
/* 2-D, replaces isinside(testpoint, pts) (src/isinside.jl:17-33): winding sum of the signed angles
 * around the ordered, closed polygon poly_xy (m x 2); inside iff |sum| >= 1e3 eps(dtype), or the
 * point coincides (r < 100 eps) with a polygon point.  The ordering check of
 * _validate_polygon_ordering (:36-69) is the caller's (it needs the polygon only).  m < 3 is an
 * argument error.  sum_out: n values or NULL.  */
int wtp_isinside_winding(wtp_ctx* ctx, const void* test_xy, int64_t n, const void* poly_xy, int64_t m,
                         int dtype, uint8_t* inside_out, void* sum_out);

/* ---- triangle-mesh geometry index: the octree method of repel --------
 * Replaces TriangleIndex(T, mesh) + the TriangleOctree queries repel makes
 * (src/octree/triangle_octree.jl:22-31,221-277; queries :71-99,532-607; src/repel.jl:522-537).
 * vertices: nv x 3 of dtype (the index's machine type T); triangles: nt x 3 int32, 0-based.
 * The library derives unit face normals and the angle-weighted edge / vertex pseudonormals
 * (keyed by exact coordinates, so triangle soup with duplicated vertices needs no welding) and
 * builds its own search tree; one mesh per context, replaced by the next call.  */
int wtp_mesh_set(wtp_ctx* ctx, const void* vertices, int64_t nv, const int32_t* triangles, int64_t nt,
                 int dtype);
int wtp_mesh_clear(wtp_ctx* ctx);
/* nt x 3 unit face normals (zero for degenerate triangles), as doubles.  */
int wtp_mesh_face_normals(wtp_ctx* ctx, double* normals_out);
/* {min xyz, max xyz} of the vertices, flat axes widened (_compute_bbox_raw, :279-291).  */
int wtp_mesh_bounds(wtp_ctx* ctx, double bbox_out[6]);
/* Per query point (host array n x 3 of dtype, converted once to the mesh's type as the reference's
 * seam does, results converted back); every output may be NULL:
 *   sd_out        signed distance, negative inside (_compute_signed_distance_octree, :583-607)
 *   tri_out       nearest triangle, 0-based; of equidistant triangles the smallest index
 *                 (the reference keeps whichever its octree traversal meets first: unpinned)
 *   closest_out   n x 3 closest point on that triangle (closest_point_on_triangle, geometric_utils.jl:68-136)
 *   inside_out    isinside(p, octree) (:97-99): inside the vertex bbox and sd < 0
 *   projected_out n x 3 _project_to_boundary(p, octree, offset) (src/repel.jl:522-537):
 *                 closest point - offset * face normal of the landing triangle  */
int wtp_mesh_query(wtp_ctx* ctx, const void* xyz, int64_t n, int dtype, double offset, void* sd_out,
                   int32_t* tri_out, void* closest_out, uint8_t* inside_out, void* projected_out);
/* Installs the wall rule _constrain_octree (src/repel.jl:448-469) on the current relax session:
 * after every sweep a boundary point is re-projected onto the mesh (offset_dist inward) and a
 * volume point that left the domain returns to its previous position and is flagged escaped
 * (stats.n_escaped counts them).  The first n_boundary movable points start as boundary points
 * (src/repel.jl:148).  The mesh must stay set until wtp_relax_end.  */
int wtp_relax_set_wall(wtp_ctx* ctx, int64_t n_boundary, double offset_dist);
/* Per movable point: landing triangle of its last projection (0-based, -1 never projected),
 * boundary membership, escaped flag (cleared by this read when clear_escaped != 0: the
 * deposition pass consumes it, src/repel.jl:486-487).  Outputs may be NULL.  */
int wtp_relax_get_wall(wtp_ctx* ctx, int32_t* tri_out, uint8_t* is_bnd_out, uint8_t* escaped_out,
                       int clear_escaped);
/* The k nearest snapshot points of arbitrary positions (nq x dim host array of the session's
 * dtype), searched in the structure of the last rebuild — the `tree` the sweep used: replaces
 * knn(tree, site, kq, true) of _deposit_escaped! (src/repel.jl:502).  idx_out: nq x k snapshot
 * indices (0-based), ascending (d2, index); dist_out: nq x k distances or NULL.  */
int wtp_relax_query_knn(wtp_ctx* ctx, const void* xyz, int64_t nq, int k, int32_t* idx_out, void* dist_out);
/* Writes membership and landing triangles back after the host-side deposition pass
 * (_deposit_escaped!, src/repel.jl:471-520, serial by design).  */
int wtp_relax_set_wall_flags(wtp_ctx* ctx, const uint8_t* is_bnd, const int32_t* tri);

/* ---- consumers of the k-NN rows ------------------------------------
 * Replaces compute_normals(points; k) / update_normals! (src/normals.jl:15-69): per point the
 * eigenvector of the smallest eigenvalue of the covariance of its k nearest points (self
 * included, KNearestSearch(points, k)), unit length.  eigen() leaves the sign open (the
 * reference fixes it afterwards with orient_normals!); here the component of largest
 * magnitude is positive.  normals_out: n x dim of dtype.  2 <= k <= n.  */
int wtp_pca_normals(wtp_ctx* ctx, const void* xyz, int64_t n, int dim, int dtype, int k, void* normals_out);
/* Replaces _gradient_limit_field (src/discretization/algorithms/octree.jl:677-717) on the
 * leaf centres: k-NN graph (self included, distances) + min-plus Jacobi sweeps
 * h[i] <- min(h[i], min_j h[j] + g d_ij) until the largest relative change of a sweep is
 * below tol or max_sweeps ran.  h0 / h_out: n values of dtype; sweeps_out: sweeps applied.  */
int wtp_gradient_limit(wtp_ctx* ctx, const void* centers, int64_t n, int dim, int dtype, int k, const void* h0,
                       double g, double tol, int max_sweeps, void* h_out, int* sweeps_out);

/* ---- sharded sessions (no counterpart in the reference) --------------
 * One rank sweeps one spatial slab.  Its session's fixed head is the ghost layer received
 * from the neighbouring ranks, its movable tail the points it owns.  The two calls below
 * keep such a session resident across iterations: only the layers cross the boundary.  */

/* Run the library on a caller-owned HIP stream (e.g. the stream RCCL's results are ordered
 * on) instead of the context's own: external = 1 lends hip_stream (NULL is the device's
 * default stream), external = 0 restores the context's stream.  Device pointers handed to
 * the *_dev entry points must be ready in the order of the stream in use.  */
int wtp_set_stream(wtp_ctx* ctx, void* hip_stream, int external);

/* Boundary layers of the movable points at their current positions, as packed 4-vectors
 * {x, y, z, bits(movable index)} (dtype of the session) in device memory:
 *   d_lo4  <- points with coord[axis] <  lo_in      counts[0]
 *   d_hi4  <- points with coord[axis] >= hi_in      counts[1]
 * and, counted only, the strays: coord < lo_out -> counts[2], coord >= hi_out -> counts[3].
 * Order within a layer is deterministic (slot order of the last rebuild).  At most `cap`
 * points are written per layer; counts report the true sizes.  */
int wtp_relax_layers_dev(wtp_ctx* ctx, int axis, double lo_in, double hi_in, double lo_out, double hi_out,
                         void* d_lo4, void* d_hi4, int64_t cap, int64_t counts[4]);

/* wtp_relax_step followed by wtp_relax_layers_dev on the positions it produced, with one read-back
 * and one synchronisation for both: the layers of iteration i+1 come home with the statistics of
 * iteration i.  */
int wtp_relax_step_layers(wtp_ctx* ctx, int rebuild, wtp_step_stats* stats, int axis, double lo_in, double hi_in,
                          double lo_out, double hi_out, void* d_lo4, void* d_hi4, int64_t cap, int64_t counts[4]);

/* The same for a block decomposition: layers along up to three axes (bit a of axes_mask set), all with the one
 * read-back.  counts[4*a .. 4*a+3] as above; d_lo4[a] / d_hi4[a] hold `cap` rows each.  */
int wtp_relax_step_layers3(wtp_ctx* ctx, int rebuild, wtp_step_stats* stats, int axes_mask, const double lo_in[3],
                           const double hi_in[3], const double lo_out[3], const double hi_out[3], void* const d_lo4[3],
                           void* const d_hi4[3], int64_t cap, int64_t counts[12]);

/* ---- the context owns an RCCL communicator ---------------------------------------------------
 * One process per GPU, one context per process.  With these four calls and the wtp_relax_*_dev / *_layers3 /
 * set_fixed_dev / set_coverage_box entry points a caller without torch.distributed (the Julia side) runs the block
 * decomposition of DESIGN.md §7b through the C ABI alone; INTEGRATION.md lists the loop.  librccl is opened with dlopen
 * at the first call, so single-GPU users never load it.  No counterpart in the reference (it has no multi-GPU path). */
#define WTP_COMM_ID_BYTES 128
/* Rank 0 creates the id (ncclGetUniqueId); the caller carries the 128 bytes to the other ranks (MPI, a file, ...). */
int wtp_comm_unique_id(wtp_ctx* ctx, void* id_out);
/* Collective over all ranks: ncclCommInitRank on the context's device. */
int wtp_comm_init(wtp_ctx* ctx, const void* id, int rank, int nranks);
int wtp_comm_finalize(wtp_ctx* ctx);
/* One point-to-point round with the low and the high neighbour along an axis (peer = rank, or -1 for none).  Rows are
 * 16 bytes — the packed {x, y, z, w} fp32 rows wtp_relax_step_layers3 fills and wtp_relax_set_fixed_dev takes.  The
 * counts travel first, then the rows, all on the context's stream: on return the received counts are known on the
 * host and the rows are stream-ordered (the next launch on this context sees them).  d_recv_* hold `cap` rows; if a
 * peer sends more, the rows are dropped, the counts are still returned and the call fails with WTP_ERR_ARG (nobody
 * is left hanging).  Every rank that names a peer must be named by that peer in the same call. */
int wtp_comm_exchange_rows(wtp_ctx* ctx, int peer_lo, int peer_hi, const void* d_send_lo, int64_t n_send_lo,
                           const void* d_send_hi, int64_t n_send_hi, void* d_recv_lo, void* d_recv_hi, int64_t cap,
                           int64_t* n_recv_lo, int64_t* n_recv_hi);
/* The global view of a sweep, in place: maximum of max_force; sums of sum_u, sum_u2, n_move, n_fallback, n_uncovered,
 * n_escaped — what the stop rules of `_relax!` (src/repel.jl:293,305-334) and the ghost-width check read.  argmin_r
 * becomes the global minimum; argmin_i / argmin_j stay local indices on the rank that holds it and become -1 elsewhere. */
int wtp_comm_allreduce_stats(wtp_ctx* ctx, wtp_step_stats* stats);

/* ---- the whole sharded iteration behind the boundary ------------------------------------------------
 * wtp_block_* run one rank's share of a multi-GPU repel: what `_relax!`'s loop body (src/repel.jl:243-334) becomes when
 * the cloud is cut into the cells of an orthtree partition, one cell (an axis-aligned box) per rank.  The caller hands
 * over its owned points and every rank's box once, then calls wtp_block_step once per iteration — ghost exchange,
 * migration, hash, sweep, the global statistics and the ghost-width check all happen inside:
 *
 *   exchange   ONE grouped point-to-point round with every spatially adjacent rank (ncclSend/ncclRecv inside one
 *              group: faces, edges and corners directly, 7 peers for 2 x 2 x 2 octants, one xGMI link each).  Row counts
 *              are known beforehand: they rode with the previous iteration's statistics, so no host round trip sits
 *              between the counts and the rows.
 *   ghosts     rank q receives every foreign point within ghost_width + margin of its box; they become the fixed
 *              head of the local snapshot (searched, never moved, never counted).
 *   migration  a point that strayed more than `margin` past its owner's box travels in the same round, straight to
 *              the rank whose box holds it (64-bit global ids travel along); the sender keeps it as a ghost for
 *              this iteration.
 *   sweep      hash + sweep of [ghosts ; owned] (wtp_relax_step's kernels), then, from the positions it produced, the
 *              rows and counts of the NEXT exchange.
 *   statistics one all-gather per iteration carries {the step's statistics, the next row counts}; every rank reduces
 *              the gathered statistics in rank order (a fixed order: the global sums are reproducible).  This is the
 *              iteration's only host synchronisation.
 *   coverage   the sweep counts the queries whose support reaches past the box the snapshot is complete for; if any rank
 *              reports one, all ranks undo the step, widen the ghost layer by 1.5x and repeat it.
 * fp32, 3-D; constant spacing or a device-evaluated law (LOGLIKE / BOUNDARY_LAYER: ghosts need no spacing, only queries
 * do).  Needs wtp_comm_init (the context's RCCL communicator) or a caller-supplied transport.  */
typedef struct wtp_block_desc {
    int32_t rank, nranks;
    const double* boxes;  /* nranks x 6: {lo x, lo y, lo z, hi x, hi y, hi z} of every rank's box, half-open [lo, hi);
                             the boxes tile space (outer ends +-inf); identical on every rank                       */
    double ghost_width;   /* w: at least the largest support u0 * s of a query near a face (checked, widened on demand) */
    double margin;        /* lazy migration: a point changes owner once it is this far outside its box; < 0: w / 4  */
} wtp_block_desc;

typedef struct wtp_block_info {
    int64_t n_owned, n_ghost;   /* this rank, in the iteration just done                                         */
    int64_t n_sent_rows, n_recv_rows; /* ghost rows of the iteration's exchange (16 bytes each)                  */
    int64_t n_emigrated, n_immigrated;
    int32_t n_peers;            /* ranks this rank exchanges with                                                 */
    int32_t widened;            /* times the ghost layer was widened so far                                       */
    int32_t host_syncs;         /* host synchronisations this call made (1 in steady state)                       */
    int32_t redone;             /* 1 if the step was undone and repeated with a wider layer                       */
    double ghost_width;         /* current w                                                                      */
    int32_t overlapped;         /* 1 if the owned points were ranked into the cells while the ghost rows travelled */
    int32_t reserved;
} wtp_block_info;

/* Optional transport in place of the context's RCCL communicator (MPI without GPU awareness, tests with several ranks
 * on one GPU): host buffers only; the library stages through the host around the callbacks.  All ranks call in step. */
typedef struct wtp_transport {
    void* user;
    /* every rank contributes `bytes` bytes; recv holds nranks * bytes, rank-major */
    int (*allgather)(void* user, const void* send, void* recv, int64_t bytes);
    /* message j goes to and comes from rank peers[j] (send_bytes[j] / recv_bytes[j] may be 0) */
    int (*exchange)(void* user, int n_msgs, const int* peers, const void* const* send, const int64_t* send_bytes,
                    void* const* recv, const int64_t* recv_bytes);
} wtp_transport;
int wtp_block_set_transport(wtp_ctx* ctx, const wtp_transport* t); /* NULL: back to RCCL */

/* d_owned_xyz: n_owned x 3 fp32 on the context's GPU; d_gid: n_owned int64 global ids (device).  spacing / force / k /
 * alpha as wtp_relax_init.  Collective: every rank calls it.  */
int wtp_block_open(wtp_ctx* ctx, const wtp_block_desc* desc, const void* d_owned_xyz, const int64_t* d_gid,
                   int64_t n_owned, const wtp_spacing_desc* spacing, const wtp_force_desc* force, int k,
                   double alpha_lo, double alpha_max);
/* One iteration; stats = the GLOBAL view (max / sums over all ranks; argmin_i / argmin_j are global ids, on every rank);
 * info may be NULL.  */
int wtp_block_step(wtp_ctx* ctx, wtp_step_stats* stats, wtp_block_info* info);
/* n_iters iterations; conv_out[n_iters] = global max |F| s per iteration (may be NULL).  */
int wtp_block_run(wtp_ctx* ctx, int n_iters, double* conv_out, wtp_step_stats* last, wtp_block_info* info);
/* The reference's stop rules on the global statistics (src/repel.jl:305-334, same order as wtp_relax_run_until): every
 * rank sees the same numbers, so all stop at the same iteration.  */
int wtp_block_run_until(wtp_ctx* ctx, int max_iters, double tol, int stall_after, double cv_target, double* conv_out,
                        int* n_done, int* reason, wtp_step_stats* last);
/* Owned points now: *n_owned of them; d_xyz_out (n x 3 fp32) and d_gid_out (int64) are device buffers of at least
 * `cap` entries, either may be NULL (then only the count is returned).  */
int wtp_block_get(wtp_ctx* ctx, void* d_xyz_out, int64_t* d_gid_out, int64_t cap, int64_t* n_owned);
int wtp_block_close(wtp_ctx* ctx);
/* Host-only helpers (no GPU): the block grid px x py x pz for nranks (as cubic as possible, larger factors on later
 * axes) and the Morton rank of block (ix, iy, iz) — the orthtree's leaf order along its Z-curve, so neighbouring ranks
 * are spatial neighbours (src/octree/spatial_octree.jl:283 `find_leaf` is the reference's only use of that order). */
int wtp_block_grid(int nranks, int p_out[3]);
int wtp_block_morton_rank(int ix, int iy, int iz, const int p[3]);
/* Grouped point-to-point primitive of the exchange, exposed for callers that drive their own iteration: message j is
 * sent to and received from rank peers[j], rows of 16 bytes, device buffers, stream-ordered on the context's stream.
 * Counts must agree on both sides (ncclSend/ncclRecv semantics).  peers[j] may equal the caller's own rank. */
int wtp_comm_exchange_peers(wtp_ctx* ctx, int n_msgs, const int* peers, const void* const* d_send, const int64_t* n_send,
                            void* const* d_recv, const int64_t* n_recv);
/* All-gather of `bytes` bytes per rank between device buffers on the context's stream (bytes a multiple of 8). */
int wtp_comm_allgather_dev(wtp_ctx* ctx, const void* d_send, void* d_recv, int64_t bytes);

/* Coverage of a sharded session: the caller guarantees that the snapshot holds every point of
 * the global cloud with lo <= coord[axis] <= hi (its slab plus the ghost layers; an end may be
 * +-inf).  A sweep then counts in stats.n_uncovered the movable points whose answer needs more:
 * a k-th neighbour (compact-support sweep: the law's support u0*s, or the nearest neighbour)
 * farther away than the nearer end of that range — so the caller can widen the layer and redo
 * the step (wtp_relax_revert).  axis < 0: unlimited (default).  */
int wtp_relax_set_coverage(wtp_ctx* ctx, int axis, double lo, double hi);
/* The same for a block decomposition (orthtree cells, 2 x 2 x 2 octants on 8 GPUs): the snapshot
 * holds every point of the global cloud inside the box lo[a] <= coord[a] <= hi[a] (the rank's block plus the
 * ghost layers received from its face, edge and corner neighbours; ends may be +-inf).  */
int wtp_relax_set_coverage_box(wtp_ctx* ctx, const double lo[3], const double hi[3]);

/* Replace the fixed head of the snapshot by n_fixed_new points (packed 4-vectors in device
 * memory, 4th component ignored).  Movable indices are unchanged; the next wtp_relax_step
 * rebuilds.  Constant spacing or a device-evaluated law (LOGLIKE / BOUNDARY_LAYER: the law is evaluated at the
 * movable points only, so the new head needs no spacing values); not with a PER_POINT array.  */
int wtp_relax_set_fixed_dev(wtp_ctx* ctx, const void* d_fixed4, int64_t n_fixed_new);

/* ---- measurement hooks (bench.py, profiles/) -------------------------------------- */

/* Device time (ms, HIP events on the context's stream) spent in each phase since the
 * last reset: [0] hash build, [1] neighbour sweep (dominant kernel), [2] fallback +
 * reductions, [3] sweep-kernel launches counted.  */
int wtp_timers_get(wtp_ctx* ctx, double out[4]);
int wtp_timers_reset(wtp_ctx* ctx);
/* Diagnostic builds only (-DWTP_DIAG=1): per-phase wave-cycle sums of the sweep kernel since the last
 * call ([0..6] phases, [7] waves); release builds return zeros.  tools/exp_diag.py. */
int wtp_debug_diag(wtp_ctx* ctx, unsigned long long out[16]);

/* Synthetic workload generator of SURVEY.md §8d, written straight into device memory:
 * value(i, axis) = (splitmix64(seed*2^40 + 3*(first+i) + axis) >> 40) * 2^-24.
 * d_out: n x dim of dtype on the context's GPU.  (bench.py / sharded driver only.) */
int wtp_gen_uniform_dev(wtp_ctx* ctx, uint64_t seed, int64_t first, int64_t n, int dim, int dtype,
                        void* d_out);

#ifdef __cplusplus
}
#endif
#endif /* WTP_H */
