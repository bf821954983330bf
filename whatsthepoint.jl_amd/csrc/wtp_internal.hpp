// wtp_internal.hpp — shared types of libwtp (gfx950 only; no portability layer).
//
// Data layout in HBM:
//   Pt<T>      one point = {x, y, z, bits(id)} : float4 (16 B) / double4 (32 B).  One 16-B
//              (or 2x16-B) coalesced access moves a whole point; the id rides along so the
//              counting sort permutes nothing else.
//   cell_start int32[ncells+1]  exclusive scan of per-cell counts (row-major cz,cy,cx).
//   Points of one cell are contiguous, cells of one x-row are contiguous.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/wtp.h"

namespace wtp {

// ---- point record ---------------------------------------------------------------------------
template <typename T> struct PtOf;
template <> struct PtOf<float> { using type = float4; };
template <> struct PtOf<double> { using type = double4; };
template <typename T> using Pt = typename PtOf<T>::type;

__host__ __device__ inline float id_to_w(float, int32_t id) { return __builtin_bit_cast(float, id); }
__host__ __device__ inline double id_to_w(double, int32_t id) {
    return __builtin_bit_cast(double, (int64_t)id);
}
__host__ __device__ inline int32_t w_to_id(float w) { return __builtin_bit_cast(int32_t, w); }
__host__ __device__ inline int32_t w_to_id(double w) {
    return (int32_t)__builtin_bit_cast(int64_t, w);
}

// ---- uniform grid (device resident; written by grid_setup_kernel) ------------------------------
// Cell of coordinate v on axis a: clamp((int)floor((v - org[a]) * inv_c), 0, n[a]-1).
// Points outside the box (repel has no wall inside the sweep) pile into the edge cells; the
// exactness radius treats those cells as unbounded outward.
template <typename T> struct Grid {
    T org[3];
    T c;      // cell edge
    T inv_c;  // 1/c
    T margin; // c * 2^-8: covers the rounding of the cell map (n[a] <= 4096 enforced)
    int32_t n[3];
    int32_t ncells;
    int32_t nb[3];   // bricks per axis
    int32_t nbricks;
    int32_t dim;
    int32_t npts;
    int32_t rad_wave_only; // RadiusTopology builds: rows are expected to outgrow the brick kernel's 32 entries — the wave kernel serves every query
    int32_t pad_;
};

constexpr int kMaxAxisCells = 4096;

// Brick geometry of the fast path: one workgroup sweeps BX x BY x BZ cells, staging the
// (BX+2)(BY+2)(BZ+2) halo in LDS.
constexpr int BX = 4, BY = 4, BZ = 4;
constexpr int HX = BX + 2, HY = BY + 2, HZ = BZ + 2;
constexpr int HCELLS = HX * HY * HZ;
// RadiusTopology: a brick whose halo holds more points than this (~7.4 per cell) has rows beyond the lane-per-query
// kernel's 32 entries (a row is ~4.06 cells' worth of points): it belongs to the dense kernel, or to the wave kernel
constexpr int kRadDenseMin = 1600;
constexpr int kBrickThreads = 256;
// partial-reduction slots: [0, brick_partials()) brick blocks, then kWavePartials, then kGenericPartials
constexpr int kWavePartials = 4096;    // wave-per-query kernel blocks
constexpr int kGenericPartials = 1024; // serial last-resort kernel blocks
int brick_partials();
constexpr int kGenericKMax = 128; // largest k the library accepts (generic kernel's list)

// ---- per-block partial reductions of one sweep -------------------------------------------------
struct Partial {
    double max_force;
    double sum_u;
    double sum_u2;
    double argmin_r;   // +inf when empty
    int64_t argmin_i;  // snapshot-global id of the movable point (ties: lowest id)
    int64_t argmin_j;
    int64_t n_move;
};

struct ForceParams {
    int32_t kind;
    double beta, u0, gamma;
};

// ---- kernel parameter blocks -------------------------------------------------------------------
template <typename T> struct SearchArgs {
    const Grid<T>* grid;
    const Pt<T>* snap;         // sorted snapshot (search structure)
    const int32_t* cell_start; // ncells+1
    const Pt<T>* query;        // query positions, slot-aligned with snap (== snap when fresh)
    int32_t n;
    int32_t k;                 // neighbours wanted (relax: kk incl. self slot)
    int32_t include_self;      // topology: 1 = raw search result, 0 = self removed by index
    // topology outputs (row = original id)
    int32_t* idx_out;
    T* dist_out;
    // relax
    Pt<T>* out;                // new positions, slot order
    T* forces;                 // slot order
    T* nn_dist;
    int32_t* nn_id;
    const T* spacing_pp;       // per-point spacing by original id, or nullptr
    T spacing_const;
    T alpha_lo, alpha_max;
    T beta, u0, gamma;
    int32_t force_kind;
    int32_t n_fixed;
    Partial* partials;         // [n_partials]
    int32_t n_partials;
    // blocks each sweep kernel was launched with = partial slots it wrote (set by the launchers):
    // brick [0, used_brick), wave [brick_partials(), +used_wave), serial [n_partials - kGenericPartials, +used_generic)
    int32_t used_brick, used_wave, used_generic;
    // RadiusTopology through the brick kernel: r^2, row lengths out (count phase) or row starts in (fill phase)
    T radius2;
    int32_t* rad_counts;
    const int64_t* rad_offsets;
    int32_t rad_fill;
    int32_t* rad_tmp;          // count phase, fp32 brick kernel: the sorted row of every query it serves (32 ids each) is parked here,
    uint8_t* rad_done;         // and the query marked (1), so that the fill phase copies rows instead of searching again (or nullptr)
    int32_t* rad_arena;        // count phase, wave kernel: ranked rows of any length, bump-allocated (mark 2, start in rad_arena_off)
    int64_t* rad_arena_off;
    unsigned long long* rad_arena_pos;
    int64_t rad_arena_cap;
    int32_t* rad_bricks;       // bricks listed for the dense kernel (at most one per point)
    int32_t rad_dense;         // > 0: the brick-staged wave-per-query kernel (wtp_radb.hip) takes the bricks whose halo holds more than kRadDenseMin and at most this many points
    // fallback work list
    int32_t* fb_list;
    int32_t* fb_count;
    int32_t* fb2_list;         // second level: wave kernel -> serial kernel
    int32_t* fb2_count;
    int32_t* nn_list;          // round-2 sweep: queries whose nearest neighbour the follow-up kernel still has to find
    int32_t* nn_count;
    int32_t* ball_list;        // what the ball kernel (variable-spacing hand-backs) leaves for the exact path, and its count
    int32_t* ball_count;
    const int32_t* stop;       // wtp_relax_run_until: non-zero once a stop rule has fired; later sweeps of the batch do nothing
    // sharded sessions: the snapshot is complete only for cover_lo <= coord[cover_axis] <= cover_hi;
    // queries whose neighbourhood reaches past that range are counted (wtp_relax_set_coverage)
    int32_t cover_axis;        // -1: unlimited; 0..2: a slab along that axis; 3: the box cover_lo3 .. cover_hi3
    T cover_lo, cover_hi;
    T cover_lo3[3], cover_hi3[3];
    int32_t* uncovered;
    // tunables
    T gamma_cap;               // initial filter radius cap, in cell edges
    float cap_count;           // k-selection kernels: points the first filter ball is expected to hold (0: fixed gamma_cap * c)
    T tnn_frac;                // CS sweeps: nearest-neighbour margin of the ring, in cell edges (WTP_TNN, default 0.8)
    int32_t brick_hcap;        // LDS point capacity for the brick kernel (0 = default)
    int32_t cs2_bx;            // > 0: brick length (own cells along x) of the round-2 compact-support sweep (wtp_cs2.hip)
    const uint8_t* brick_dead; // wtp_cs2.hip, variable spacing: bricks whose points all went to the ball kernel's list already (cs2_dead_kernel), or nullptr
    int32_t brick_dead_cap;
    int32_t cs2_chunked;       // wtp_cs2.hip: runs longer than the hit masks are taken in chunks (variable spacing, several points per cell)
    int32_t counters_cleared;  // topology calls: the caller cleared fb_count / fb2_count (one 64-byte block) itself
    int32_t fb_r0;             // first block radius (cells) of the exact path for hand-backs; 0: the default (2: the 27 cells failed already)
    int32_t ksel_bx;           // > 0: the grid was built for the k-selection kernels of wtp_ksel.hip; largest brick length along x
    unsigned long long* diag;  // -DWTP_DIAG builds: per-phase wave-cycle sums (8 slots), else unused
};

// ---- device buffer with capacity -----------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

// Input view of a hash build whose input array still holds stale fixed points and has the new
// ones appended: entry i is dropped when i < n_old and its id < fixed_old; kept entries below
// n_old get id + id_shift (fixed head resized), entries from n_old on keep their id.
struct HashView {
    bool active = false;
    int64_t n_in = 0;    // entries in the input array
    int64_t n_old = 0;   // entries of the previous snapshot (stale fixed points among them)
    int32_t fixed_old = 0;
    int32_t id_shift = 0;
};

// Part of a hash build issued ahead of time (block driver: while the ghost rows of the iteration travel): the old
// snapshot's entries [0, n_old) of `in` are already ranked into the cell counts.  One-shot; build_hash takes it over only
// when every pointer and number still matches, otherwise it zeroes the counts and ranks everything itself.
struct Prerank {
    bool valid = false;
    const void* in = nullptr;
    int64_t n_old = 0;
    int32_t fixed_old = 0;
    const void *cnt = nullptr, *cr = nullptr, *dirty = nullptr;
};

struct RelaxState {
    bool active = false;
    int64_t n = 0, n_fixed = 0;
    int dim = 3, dtype = 0, k = 0;
    int k_req = 0;           // k as requested (k = min(k_req, n) follows n when the fixed head is swapped)
    int spacing_kind = 0;
    double spacing_const = 0, alpha_lo = 0, alpha_max = 0;
    ForceParams force{};
    int bufS = -1, bufP = -1, bufOld = -1; // indices into pts[3]
    bool have_tree = false;
    bool can_revert = false;
    bool have_point_data = false;
    double spacing_max = 0;  // largest spacing value (host-side max of the per-point array)
    int brick_hcap = 0;      // LDS point capacity of the sweep's brick kernel (0 = not chosen yet)
    bool grid_tuned = false; // cell_scale / spacing_typ measured on the first rebuild
    int grid_age = 0;             // rebuilds since the grid (bounding box, cell edge) was last computed
    int sweeps_since_rebuild = 0; // every sweep moves a point by at most its spacing (src/repel.jl:286-289)
    bool moved_by_hand = false;   // wtp_relax_set since the last rebuild: that bound is gone
    double cell_scale = 1.0; // < 1: cells shrunk because the occupied ones hold more than the box average
    double spacing_typ = 0;  // mean spacing over the snapshot (floor of the compact-support cell edge)
    bool cs_sweep = false;   // compact-support sweep in use (ClippedSpacingForce)
    bool cs_disabled = false; // measured on the first rebuild: support cells would be over-full, use the k-selection sweep
    bool ksel_sweep = false; // k-selection sweep on the x-slowest layout (wtp_ksel.hip)
    int ksel_bx = 0, ksel_hcap = 0; // its brick length along x and LDS point area, measured with the grid
    double ksel_rho = 0;     // the occupancy picked for this cloud (ksel_pick_rho)
    double cs2_rho = 0;      // points per cell the sweep's bricks were sized for (cs2_tune)
    int cs2_bx = 0;          // > 0: the round-2 compact-support sweep (wtp_cs2.hip) with bricks of this many cells along x
    int64_t tuned_fixed = 0; // fixed points the grid / brick geometry was measured with (a swapped head re-measures when it differs by > 5 % of n)
    double sp_p0 = 0, sp_p1 = 0, sp_p2 = 0; // LOGLIKE / BOUNDARY_LAYER parameters
    HashView pending;        // wtp_relax_set_fixed_dev left its work to the next rebuild (see there)
    int64_t shard_extra = 0; // extra capacity of the point buffers once the fixed head gets replaced
    int64_t aux_off = 0;     // device-evaluated spacing laws: sp_hint / sp_cert are stored at [id - aux_off] (a swapped fixed head shifts the movable ids, not the entries)
    int swap_target = -1;    // relax_swap_begin .. relax_swap_commit (wtp_block.hip: migration)
    bool shard_grid_reuse = false; // block sessions: the grid is kept across a swapped ghost head (points outside it pile into edge cells, which every search treats as unbounded outward)
    int64_t grid_fixed = -1;       // fixed points the current grid's bounding box was computed with
    // fp64 sweeps through fp32 candidates: what the float copy's grid was measured with (cloud size, cell scale, occupancy, brick geometry)
    int64_t f64k_n = 0;
    double f64k_scale = 1.0, f64k_rho = 0.0;
    int f64k_bx = 0, f64k_hcap = 0;
    double last_rho_cs = 0.0;      // occupancy argument of the session's last hash build (relax_prerank sizes its scratch alike)
    bool wall_active = false; // octree method: _constrain_octree runs after every sweep (wtp_relax_set_wall)
    double wall_offset = 0;   // inward nudge of a projected boundary point (src/repel.jl:143)
    int64_t wall_nm = 0;      // movable points the wall arrays are sized for
    int cover_axis = -1;     // sharded session: snapshot complete for cover_lo <= coord[axis] <= cover_hi
    double cover_lo = 0, cover_hi = 0;
    double cover_lo3[3] = {0, 0, 0}, cover_hi3[3] = {0, 0, 0}; // cover_axis == 3: a box (ends may be +-inf)
};

} // namespace wtp

struct wtp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;     // the stream every launch goes to
    hipStream_t own_stream = nullptr; // created with the context; `stream` unless wtp_set_stream lent another
    std::string err;
    int sm_count = 256;
    // tunables (env WTP_RHO / WTP_GAMMA_CAP / WTP_FORCE_GENERIC)
    double rho = 9.0;            // WTP_RHO: points per cell of the k = 21 selection grids (round 2: 8 -> 9, fewer hand-backs; measured)
    double gamma_cap = 1.0;      // WTP_GAMMA_CAP: first filter radius of the topology kernels, in cell edges (round 2: 1.08 -> 1.0, fewer ring prunes)
    double gamma_cap_sweep = 0.96; // WTP_GAMMA_CAP_SWEEP: the same for the sweep with explicit k-selection (4.28 -> 3.55 ms at 10 M)
    double tnn_frac = 0.8;     // WTP_TNN: measured optimum between candidate volume and isolated-query hand-backs (0.9: 1.55 ms, 0.8: 1.44, 0.7: 1.69 per 10 M step)
    int force_generic = 0;
    int full_select = 0;       // WTP_FULL_SELECT=1: never use the compact-support sweep
    int cs2 = 1;               // WTP_CS2=0: the round-1 compact-support sweep (brick_kernel<1,0,1>) instead of wtp_cs2.hip
    double rho_cs2 = 1.0;      // WTP_RHO_CS: target points per cell of the round-2 sweep (the support floor usually binds)
    int ksel = 1;              // WTP_KSEL=0: the round-1 k-selection kernels (4 x 4 x 4 bricks, wtp_brick.hip) instead of wtp_ksel.hip
    double rho_ksel = 1.2;     // WTP_RHO_KSEL: points per cell of the wtp_ksel.hip grids at k + self = 22 (scales with k)
    double cap_ksel = 40.0;    // WTP_CAP_KSEL: points the first filter ball of wtp_ksel.hip is expected to hold at k + self = 22
    size_t cs2_smem = 0;       // launch attributes of cs2_kernel cached per context
    const void* cs2_fn = nullptr; // (and the variant they belong to)
    int cs2_occ = 0;
    // (kernel, dynamic LDS bytes) -> blocks per CU, per CONTEXT: the dynamic-LDS attribute and the occupancy are
    // properties of a kernel on one device, and several contexts (devices) may live in one process
    std::map<std::pair<const void*, size_t>, int> launch_cache;
    double styp_sigma = 0.0;   // WTP_STYP_SIGMA: typical spacing = mean + this many standard deviations (measured: > 0 only hurts)
    // the same cache for the fp32 candidate search of fp64 topology calls (knn_dev_f64)
    int64_t knn64_tune_n = -1;
    int knn64_tune_dim = 0, knn64_tune_k = 0, knn64_tune_ksel = -1, knn64_tune_bx = 0, knn64_tune_hcap = 0;
    double knn64_tune_scale = 1.0, knn64_tune_rho = 0;
    double knn_tune_rho = 0;   // occupancy the wtp_ksel.hip grid of that cloud was built with
    int knn_tune_ksel = -1, knn_tune_bx = 0, knn_tune_hcap = 0; // wtp_ksel.hip layout in use for that cloud, its brick geometry
    int64_t knn_tune_n = -1;   // topology calls: cloud size / dim / k the cached cell scale was measured for
    int knn_tune_dim = 0, knn_tune_k = 0;
    double knn_tune_scale = 1.0;
    bool knn_tune_boxed = false;
    // pooled device buffers
    wtp::DevBuf pts[3];        // Pt arrays
    wtp::DevBuf raw_in;        // AoS staging of host input
    wtp::DevBuf cell_of, rank_of, cell_cnt, cell_start, scan_tmp;
    wtp::DevBuf grid, bbox_part, occ;
    wtp::HashView hash_view;   // consumed by the next build_hash call (set and cleared by the caller)
    wtp::DevBuf box_dev;       // robust box {lo xyz, hi xyz} (doubles) + histogram scratch behind it
    bool topology_build = false; // set around the hash builds of KNN / radius topology calls: their rows are ordered by (d2, id) explicitly, so the
                                 // within-cell order by id (canon_kernel: 0.3 of a 1.1 ms KNN call on unsorted input) buys nothing there
    bool reuse_grid = false;   // one-shot: the next build_hash keeps the previous Grid (no bounding-box pass)
    int grid_reuse_max = 7;    // WTP_GRID_REUSE: rebuilds of a relax session that may reuse a grid (0 = never)
    bool box_active = false;   // grid_setup clips the bounding box to box_dev (outliers piled into edge cells)
    const void* ncells_dev = nullptr; // device address of Grid::ncells of the last build_hash
    wtp::DevBuf idx_out, dist_out, counts_out;
    wtp::DevBuf cand_idx, cand_dist, f32_pts; // fp64 topology: fp32 candidate lists and the float copy of the cloud
    // fp64 sweeps through fp32 candidates (wtp_sweep64.hip): the session's grid and cell table parked while the float copy's
    // are built and searched; the fp64 points and their session slots in the float copy's order; the search's own lists
    wtp::DevBuf grid_b, cell_start_b, f64k_s64, f64k_slot, f64k_lists, f64k_cnt;
    int ball64 = 1;                           // WTP_BALL64=0: Float64 variable-spacing hand-backs straight to the wave kernel
    int f64_ksel = 1;                         // WTP_F64_KSEL=0: the exact wave-per-query path for those sweeps
    wtp::DevBuf forces, nn_dist, nn_id, spacing_pp;
    wtp::DevBuf partials, stats, fb_list, fb_count, fb2_list, fb2_count, nn_list;
    wtp::DevBuf rad_pos;           // counter block: [0, 8) next free id of the arena (wtp_radb.hip takes pieces of it), [8, 12) bricks listed
    wtp::DevBuf rad_bricks;        // the dense kernel's brick list
    bool rad_dense_attr[2] = {false, false}; // wtp_radb.hip: the kernel's LDS size has been declared (fp32, fp64)
    bool rad_dense_used = false;   // the count phase ran the dense kernel: the fill phase's wave kernel works from the hand-back list
    wtp::DevBuf rad_tmp, rad_done; // RadiusTopology: rows parked by the count phase (32 ids per query), one byte per query
    wtp::DevBuf rad_arena, rad_arena_off; // ... and the wave kernel's rows (any length), their starts; the bump counter sits behind the starts
    wtp::DevBuf brick_dead;    // wtp_cs2.hip, variable spacing: one byte per brick (cs2_dead_kernel)
    wtp::Prerank prerank;             // wtp_hash.hip: prerank_old_snapshot
    int64_t preranked_builds = 0;     // hash builds that took a first half over
    hipStream_t comm_stream = nullptr; // block driver: the grouped exchange runs here while the owned points are ranked
    hipEvent_t ev_comm_a = nullptr, ev_comm_b = nullptr;
    bool hash_scratch_clean = false;  // cell counts and dirty map are all-zero (every completed build leaves them so)
    bool counters_clean = false;      // the 64-byte counter block is all-zero (the step's final reduction leaves it so)
    wtp::DevBuf stop_state;           // wtp_relax_run_until: {stopped, reason, n_done, last_impr, best_cv} on the device
    const int32_t* stop_dev = nullptr; // its first word while such a run is enqueued, else NULL (kernels then never look)
    wtp::DevBuf scratch;       // misc (relax_get staging, radius rows)
    wtp::DevBuf diag;          // diagnostic builds only
    wtp::DevBuf ins_in, ins_elems, ins_partial, ins_out; // isinside filter
    // triangle mesh of the octree method (wtp_mesh.hip): bounding-volume tree nodes, pseudonormals
    wtp::DevBuf mesh_nodes, mesh_pn, mesh_io;
    int64_t mesh_nt = 0;
    int mesh_dtype = -1;
    double mesh_bbox[6] = {0, 0, 0, 0, 0, 0};
    double mesh_scale = 0;
    std::vector<double> mesh_face_host; // unit face normals (the returned boundary's normals, src/repel.jl:614)
    wtp::DevBuf wall_flags, wall_tri;   // per movable point: is_bnd | escaped (+ counter), landing triangle
    wtp::DevBuf wall_hint;              // per movable point: tree node of its nearest triangle at the last sweep
    wtp::DevBuf mesh_cls;               // inside/outside class per cell of a uniform grid over the mesh bbox
    bool mesh_cls_ready = false;
    int mesh_cls_dim[3] = {0, 0, 0};
    double mesh_cls_cell = 0;
    int mesh_packet = 0;                // WTP_MESH_PACKET=1: standalone queries walk the tree as wave packets
    wtp::DevBuf sp_hint;       // variable spacings: nearest tree node of each snapshot point at the last sweep
    wtp::DevBuf kd_nodes;      // variable spacings: kd-tree over the boundary points (heap order)
    int64_t kd_m = 0;          // nodes in it; the key below identifies the boundary it was built from
    uint64_t kd_key = 0;
    int kd_dim = 0, kd_dtype = -1;
    int64_t n_syncs = 0;       // host synchronisations of the context's stream so far (wtp_block_info.host_syncs counts with it)
    void* block = nullptr;     // wtp::BlockState (wtp_block.hip): this rank's share of a block-decomposed repel
    void* comm = nullptr;      // ncclComm_t (wtp_comm.hip); rank and size of the communicator
    int comm_rank = 0, comm_size = 0;
    wtp::DevBuf comm_scratch;
    wtp::DevBuf sp_cert;       // device-evaluated spacing laws: per point, where it stood at its last tree walk and the bound that walk left (wtp_spacing.hip)
    void* host_pinned = nullptr;
    size_t host_pinned_cap = 0;
    // radius two-phase state
    int64_t rad_n = 0;
    int rad_dim = 0, rad_dtype = 0;
    double rad_r = 0;
    int64_t rad_nnz = 0;
    bool rad_valid = false;
    bool rad_rows_cached = false; // the count phase parked the brick kernel's rows (rad_tmp / rad_done): the fill phase copies them
    bool rad_offsets_dev = false; // wtp_radius_offsets left the CSR offsets in dist_out (device): fill may take them from there
    wtp::RelaxState relax;
    // timers
    bool timing = false;      // per-phase event pairs around every step: off until wtp_timers_reset asks for them (6 event records per step are a third of a small cloud's step)
    bool timing_forced = false; // WTP_TIMING in the environment decides, wtp_timers_reset does not
    double t_hash = 0, t_sweep = 0, t_other = 0;
    int64_t n_sweep_launches = 0;
    std::vector<hipEvent_t> ev_pool;
    struct Span { int a, b, kind; };
    std::vector<Span> spans;
    int ev_used = 0;
    int ev_last_end = -1;     // the event that closed the latest span: the next span starts from it (no second record)
};

namespace wtp {

// error plumbing
int fail(wtp_ctx* ctx, int code, const std::string& msg);
int ensure_pinned(wtp_ctx* ctx, size_t bytes); // the context's page-locked staging block, at least this large
#define WTP_HIP(ctx, call)                                                                    \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return ::wtp::fail(ctx, e_ == hipErrorOutOfMemory ? WTP_ERR_OOM : WTP_ERR_HIP,    \
                               std::string(#call) + ": " + hipGetErrorString(e_));            \
    } while (0)

int ensure(wtp_ctx* ctx, DevBuf& b, size_t bytes);
// sets the kernel's dynamic-LDS limit once per (context, kernel, size) and returns the blocks per CU it can hold
int launch_occupancy_of(wtp_ctx* ctx, const void* fn, int threads, size_t smem);

// timing spans: kind 0 hash, 1 sweep, 2 other
int span_begin(wtp_ctx* ctx, int kind);
void span_end(wtp_ctx* ctx, int span);
void spans_collect(wtp_ctx* ctx);

// ---- launch wrappers (implemented per translation unit) ----------------------------------------
// hash build: from Pt array `in` (n points) produce sorted `out`, cell_start and the grid.
// radius > 0 forces cell edge >= radius (RadiusTopology); k scales the target occupancy.
template <typename T>
int build_hash(wtp_ctx* ctx, const Pt<T>* in, Pt<T>* out, int64_t n, int dim, int k, double radius,
               double rho_direct = 0.0, double min_cell = 0.0, double cell_scale = 1.0);
template <typename T>
int prerank_old_snapshot(wtp_ctx* ctx, const Pt<T>* in, int64_t n_old, int32_t fixed_old, int64_t n_next, int64_t n_in_next,
                         int k, double rho_direct, double cell_scale);
// occupancy of the grid the last build_hash made: d_out3 = [sum cnt^2, sum cnt, max cnt]
int launch_occupancy(wtp_ctx* ctx, unsigned long long* d_out3);
template <typename T> int launch_sum(wtp_ctx* ctx, const T* d_v, int64_t n, double* d_out);
int launch_offsets_scan(wtp_ctx* ctx, const int32_t* d_cnt, int64_t n, int64_t* d_tmp, int64_t* d_off);
size_t offsets_scan_tmp_bytes(int64_t n);
// per-axis coordinate histograms over d_range = {lo xyz, hi xyz}: 3 x 1024 bins
template <typename T>
int launch_axis_hist(wtp_ctx* ctx, const Pt<T>* pts, int64_t n, int dim, const double* d_range, unsigned int* d_hist);

template <typename T>
int load_points(wtp_ctx* ctx, const T* d_xyz, Pt<T>* out, int64_t n, int dim);

template <typename T> int launch_topology(wtp_ctx* ctx, SearchArgs<T>& a);
template <typename T> int launch_sweep(wtp_ctx* ctx, SearchArgs<T>& a, bool fresh);
// fp64 compact-support brick sweep (wtp_brick64.hip); hand-backs land in a.fb_list
template <typename T> int launch_brick_cs(wtp_ctx* ctx, SearchArgs<T>& a);
// exact paths: wave-per-query (list = fb_list or all points), then the serial kernel on fb2_list
template <typename T> int launch_wave_topology(wtp_ctx* ctx, SearchArgs<T>& a, bool all);
template <typename T> int launch_wave_sweep(wtp_ctx* ctx, SearchArgs<T>& a, bool all);
template <typename T>
int launch_wave_radius_count(wtp_ctx* ctx, SearchArgs<T>& a, T r, int32_t* d_counts, const int32_t* list,
                             const int32_t* list_count);
template <typename T>
int launch_wave_radius_fill(wtp_ctx* ctx, SearchArgs<T>& a, T r, const int64_t* d_offsets, int32_t* d_idx,
                            const int32_t* list, const int32_t* list_count);
template <typename T> int launch_generic_topology(wtp_ctx* ctx, SearchArgs<T>& a, bool all);
template <typename T> int launch_query_knn(wtp_ctx* ctx, SearchArgs<T>& a, const T* d_xyz, int dim, Pt<T>* d_packed);
template <typename T> int launch_generic_sweep(wtp_ctx* ctx, SearchArgs<T>& a, bool all);
inline int total_partials() { return brick_partials() + kWavePartials + kGenericPartials; }
int launch_brick_radius(wtp_ctx* ctx, SearchArgs<float>& a);
// round-2 compact-support sweep (wtp_cs2.hip)
int launch_cs2(wtp_ctx* ctx, SearchArgs<float>& a);
int launch_cs2_followup(wtp_ctx* ctx, SearchArgs<float>& a);
int debug_kd_steps(unsigned long long out[2]); // -DWTP_DIAG builds: node visits / wave-walks of the spacing law's tree walk since the last call
int launch_cs_ball(wtp_ctx* ctx, SearchArgs<float>& a, int32_t* rest_list, int32_t* rest_count);
int launch_cs2_census(wtp_ctx* ctx, int BX, unsigned int* d_out513);
int launch_cs2_dead(wtp_ctx* ctx, SearchArgs<float>& a, uint8_t* d_dead, int dead_cap);
int cs2_max_bx();
// wtp_ksel.hip: k-selection on the x-slowest layout (fp32, 3-D, k + self <= ksel_kmax())
int launch_ksel_topology(wtp_ctx* ctx, SearchArgs<float>& a);
int launch_ksel_sweep(wtp_ctx* ctx, SearchArgs<float>& a);
int ksel_max_bx();
int ksel_kmax();
template <typename T>
int launch_radius_count(wtp_ctx* ctx, SearchArgs<T>& a, T r, int32_t* d_counts);
template <typename T>
int launch_radius_fill(wtp_ctx* ctx, SearchArgs<T>& a, T r, const int64_t* d_offsets,
                       int32_t* d_idx);
int launch_reduce_partials(wtp_ctx* ctx, const Partial* parts, int n_parts, int used_brick, int used_wave,
                           int used_generic, const int32_t* fb_count, const int32_t* uncovered,
                           const int32_t* escaped, wtp_step_stats* d_stats_slot);
// consumers of the rows (wtp_consumers.hip)
template <typename T>
int launch_pca_normals(wtp_ctx* ctx, const T* d_xyz, int64_t n, int dim, const int32_t* d_rows, int k, T* d_out);
template <typename T>
int launch_minplus_batch(wtp_ctx* ctx, const int32_t* d_rows, const T* d_dist, int64_t n, int k, double g, double tol,
                         T* d_h0, T* d_h1, int first, int sweeps, unsigned long long* d_state);
// wall rule of the octree method (wtp_mesh.hip)
template <typename TP>
int launch_mesh_constrain(wtp_ctx* ctx, const Pt<TP>* old, Pt<TP>* cur, int64_t n, int64_t n_fixed, double offset,
                          const uint8_t* is_bnd, uint8_t* escaped, int32_t* tri_idx, int32_t* hint, int32_t* n_escaped);
template <typename T>
int launch_unpermute(wtp_ctx* ctx, const Pt<T>* pts, int64_t n, int64_t n_fixed, int dim, T* d_xyz_out);
template <typename T>
int launch_unpermute_point_data(wtp_ctx* ctx, const Pt<T>* pts, int64_t n, int64_t n_fixed,
                                const T* forces, const T* nn_dist, const int32_t* nn_id,
                                T* forces_o, T* nn_dist_o, int32_t* nn_id_o);
template <typename T>
int launch_set_points(wtp_ctx* ctx, Pt<T>* pts, int64_t n, const int32_t* d_ids, int64_t m, int dim, const T* d_v);
template <typename T>
int launch_set_point(wtp_ctx* ctx, Pt<T>* pts, int64_t n, int32_t id, int dim, const T* d_xyz3);
template <typename T>
int launch_gen_uniform(wtp_ctx* ctx, uint64_t seed, int64_t first, int64_t n, int dim, T* d_out);
// variable spacing laws on the device (wtp_spacing.hip)
template <typename T> size_t kd_bytes(int64_t m);
template <typename T> void kd_build_host(const T* xyz, int64_t m, int dim, void* out);
template <typename T>
int launch_spacing_eval(wtp_ctx* ctx, const T* d_xyz, int64_t n, int dim, const void* d_nodes, int64_t m, int kind,
                        double p0, double p1, double p2, T* d_out);
template <typename T>
int launch_spacing_session(wtp_ctx* ctx, const Pt<T>* pts, int64_t n, int64_t first_id, const void* d_nodes, int64_t m,
                           int kind, double p0, double p1, double p2, T* d_spacing_pp, int32_t* d_hint,
                           const int32_t* d_cell_start = nullptr, const void* d_grid = nullptr, void* d_cert = nullptr);
// fp64 topology through an fp32 candidate search (wtp_hash.hip)
int launch_origin(wtp_ctx* ctx, const double4* pts, int64_t n, double* d_org4);
// wtp_sweep64.hip: fp64 sweeps of the k-nearest laws through fp32 candidates
int launch_f64k_local(wtp_ctx* ctx, const double4* snap, int64_t n, const double* d_org4, float4* out);
int launch_f64k_relabel(wtp_ctx* ctx, const double4* snap, float4* sorted32, int32_t* sslot, double4* s64, int64_t n);
int launch_refine_sweep_f64(wtp_ctx* ctx, SearchArgs<double>& a, const double4* s64, const int32_t* sslot, const int32_t* cand,
                            const float* cdist, const double* d_org4);
int launch_to_local_f32(wtp_ctx* ctx, const double4* in, int64_t n, const double* d_org4, float4* out);
int launch_refine_f64(wtp_ctx* ctx, const double4* raw, const int32_t* cand, const float* cdist, int64_t n, int kc, int k,
                      int include_self, const double* d_org4, int32_t* idx_out, double* dist_out, int32_t* fail_list,
                      int32_t* fail_count);
int launch_relabel_slots(wtp_ctx* ctx, const double4* raw, float4* sorted32, double4* sorted64, int64_t n);
int launch_refine_f64_slots(wtp_ctx* ctx, const double4* sorted, const int32_t* cand, const float* cdist, int64_t n, int kc, int k,
                            int include_self, const double* d_org4, int32_t* idx_out, double* dist_out, int32_t* fail_list,
                            int32_t* fail_count);
// isinside post-filter (wtp_inside.hip)
int isinside_chunks(wtp_ctx* ctx, int64_t n, int64_t m, int points_per_block);
int isinside_greens_ppb();
int isinside_winding_ppb();
size_t isinside_elem_bytes(int dtype);
template <typename T>
int launch_isinside_greens(wtp_ctx* ctx, const T* d_test, int64_t n, const T* d_p, const T* d_nrm, const T* d_area,
                           int64_t m, void* d_elems, int chunks, T* d_partial, T* d_g, uint8_t* d_inside);
template <typename T>
int launch_isinside_winding(wtp_ctx* ctx, const T* d_test, int64_t n, const T* d_poly, int64_t m, int chunks,
                            T* d_partial, int32_t* d_coincident, T* d_sum, uint8_t* d_inside);
// sharded sessions: boundary layers of the movable points / replacement of the fixed head
int layer_blocks(int64_t n);
template <typename T>
int launch_layers(wtp_ctx* ctx, const Pt<T>* pts, int64_t n, int64_t n_fixed, int axis, double lo_in, double hi_in,
                  double lo_out, double hi_out, Pt<T>* d_lo, Pt<T>* d_hi, int64_t cap, int2* d_blk, int32_t* d_totals,
                  bool slot_ordered, double reach);
template <typename T> int launch_append_fixed(wtp_ctx* ctx, const Pt<T>* d_src, int64_t n, Pt<T>* d_dst);
// block decomposition (wtp_block.hip) <-> session internals (wtp_api.hip)
int relax_step_enqueue(wtp_ctx* ctx, int rebuild, wtp_step_stats* d_slot); // one sweep, statistics into a device slot, no synchronisation
int relax_swap_begin(wtp_ctx* ctx, int64_t n_move_new, void** d_buf_out);  // a free point buffer for a replaced movable set ...
int relax_swap_commit(wtp_ctx* ctx, int64_t n_move_new);                   // ... which becomes the session's P (no fixed head, tuning kept)
int relax_set_fixed_dev_impl(wtp_ctx* ctx, const void* d_fixed4, int64_t n_fixed_new, bool keep_alive);
int comm_exchange_peers_on(wtp_ctx* ctx, hipStream_t stream, int n_msgs, const int* peers, const void* const* d_send,
                           const int64_t* n_send, void* const* d_recv, const int64_t* n_recv); // wtp_comm.hip
template <typename T> int launch_radius_dense(wtp_ctx* ctx, SearchArgs<T>& a, T r, int32_t* d_counts); // wtp_radb.hip
template <typename T> int radius_dense_hcap();
int launch_cs_all_slots(wtp_ctx* ctx, int32_t* list, int32_t n, int32_t* count); // wtp_cs2.hip: list = 0 .. n-1, *count = n
int launch_cs_ball64(wtp_ctx* ctx, SearchArgs<double>& a, int32_t* rest_list, int32_t* rest_count); // wtp_ball64.hip
int relax_prerank(wtp_ctx* ctx, int64_t n_fixed_new); // first half of the next rebuild's hash, ahead of wtp_relax_set_fixed_dev (see wtp_api.hip)
void block_destroy(wtp_ctx* ctx);                                          // frees ctx->block (wtp_destroy)
template <typename T>
int launch_refix(wtp_ctx* ctx, const Pt<T>* in, int64_t n_old, int64_t n_fixed_old, int64_t n_fixed_new,
                 const Pt<T>* d_fixed_new, Pt<T>* out, int32_t* d_counter);

} // namespace wtp
