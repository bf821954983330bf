// wtp_api.hip — the C ABI of include/wtp.h: context, buffer pool, host<->device staging and
// the call sequences that replace _build_knn_neighbors / _build_radius_neighbors
// (src/topology.jl:79-97) and the body of _relax!'s loop (src/repel.jl:243-334).
// No CPU fallback exists: without a usable gfx950 device wtp_create fails.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "wtp_internal.hpp"

namespace wtp {

static thread_local std::string g_create_err;

int fail(wtp_ctx* ctx, int code, const std::string& msg) {
    if (ctx)
        ctx->err = msg;
    else
        g_create_err = msg;
    return code;
}

int ensure(wtp_ctx* ctx, DevBuf& b, size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (b.cap >= bytes) return WTP_OK;
    if (b.p) {
        hipFree(b.p);
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = bytes + bytes / 16 + 256;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        (void)hipGetLastError(); // clear sticky OOM
        e = hipMalloc(&b.p, bytes);
        want = bytes;
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        b.p = nullptr;
        return fail(ctx, WTP_ERR_OOM, "hipMalloc of " + std::to_string(bytes) + " bytes failed");
    }
    b.cap = want;
    return WTP_OK;
}

int launch_occupancy_of(wtp_ctx* ctx, const void* fn, int threads, size_t smem) {
    const auto key = std::make_pair(fn, smem);
    auto it = ctx->launch_cache.find(key);
    if (it != ctx->launch_cache.end()) return it->second;
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    int occ = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, threads, smem);
    if (e != hipSuccess || occ < 1) occ = 1;
    ctx->launch_cache[key] = occ;
    return occ;
}

int ensure_pinned(wtp_ctx* ctx, size_t bytes) {
    if (ctx->host_pinned_cap >= bytes) return WTP_OK;
    if (ctx->host_pinned) hipHostFree(ctx->host_pinned);
    ctx->host_pinned = nullptr;
    ctx->host_pinned_cap = 0;
    WTP_HIP(ctx, hipHostMalloc(&ctx->host_pinned, bytes, hipHostMallocDefault));
    ctx->host_pinned_cap = bytes;
    return WTP_OK;
}

// ---- timing spans -----------------------------------------------------------------------------
static int take_event(wtp_ctx* ctx) {
    if (ctx->ev_used == (int)ctx->ev_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return -1;
        ctx->ev_pool.push_back(e);
    }
    return ctx->ev_used++;
}

// Spans of a step follow each other without a gap (hash | sweep | follow-ups | reduction | next hash ...), so the
// event that closes one opens the next: one record per span instead of two (an event record costs ~4 us of stream
// time; eight per step were a third of a 50 k-point step).  Work enqueued between two spans counts for the later one.
int span_begin(wtp_ctx* ctx, int kind) {
    if (!ctx->timing) return -1;
    if (ctx->spans.size() > 8192) spans_collect(ctx); // bounded pool; costs one sync
    int a = ctx->ev_last_end;
    if (a < 0) {
        a = take_event(ctx);
        if (a < 0) return -1;
        hipEventRecord(ctx->ev_pool[a], ctx->stream);
    }
    const int b = take_event(ctx);
    if (b < 0) return -1;
    ctx->spans.push_back({a, b, kind});
    return (int)ctx->spans.size() - 1;
}

void span_end(wtp_ctx* ctx, int span) {
    if (span < 0) return;
    hipEventRecord(ctx->ev_pool[ctx->spans[span].b], ctx->stream);
    ctx->ev_last_end = ctx->spans[span].b;
}

void spans_collect(wtp_ctx* ctx) {
    ctx->ev_last_end = -1; // the pool is recycled below (and a caller that reads the timers has synchronised: a gap)
    if (ctx->spans.empty()) return;
    hipStreamSynchronize(ctx->stream);
    for (auto& s : ctx->spans) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ctx->ev_pool[s.a], ctx->ev_pool[s.b]) == hipSuccess) {
            if (s.kind == 0) ctx->t_hash += ms;
            else if (s.kind == 1) ctx->t_sweep += ms;
            else ctx->t_other += ms;
        }
    }
    ctx->spans.clear();
    ctx->ev_used = 0;
}

static int sync(wtp_ctx* ctx) {
    ctx->ev_last_end = -1; // the host waits here: whatever it does next is not part of a span
    ctx->n_syncs += 1;
    WTP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return WTP_OK;
}

static size_t tsize(int dtype) { return dtype == WTP_F64 ? 8 : 4; }

static double host_max(const void* v, int64_t n, int dtype) {
    double m = 0;
    if (dtype == WTP_F64) {
        const double* p = (const double*)v;
        for (int64_t i = 0; i < n; ++i) m = p[i] > m ? p[i] : m;
    } else {
        const float* p = (const float*)v;
        for (int64_t i = 0; i < n; ++i) m = p[i] > m ? p[i] : m;
    }
    return m;
}

static int check_cloud(wtp_ctx* ctx, const void* xyz, int64_t n, int dim, int dtype) {
    if (!ctx) return WTP_ERR_ARG;
    if (!xyz) return fail(ctx, WTP_ERR_ARG, "xyz is NULL");
    if (n < 1) return fail(ctx, WTP_ERR_ARG, "n must be >= 1");
    if (n > 2000000000LL) return fail(ctx, WTP_ERR_ARG, "n exceeds the int32 index space");
    if (dim != 2 && dim != 3) return fail(ctx, WTP_ERR_ARG, "dim must be 2 or 3");
    if (dtype != WTP_F32 && dtype != WTP_F64) return fail(ctx, WTP_ERR_ARG, "dtype must be WTP_F32 or WTP_F64");
    return WTP_OK;
}

// ---- hash with a measured cell edge ------------------------------------------------------------------
// build_hash sizes its cells from the box average n / volume.  That is right for a cloud that fills
// its box evenly and wrong for graded clouds (a 64x density contrast puts ~85 points into every wall
// cell), for surface clouds and for boxes stretched by a few outliers: the occupied cells then hold
// far more points than intended, halos overflow the LDS and whole bricks drop to the slow exact
// path (measured: 157 ms instead of ~2 ms per iteration on a 1 M-point graded cloud).  So the
// first build of a session / call measures the occupancy the POINTS see (sum cnt^2 / sum cnt, = rho + 1
// for a Poisson cloud) and shrinks the cell edge until that matches the target; at most 3 builds,
// one small read-back each.  Floors (radius, the force law's support) stay in force.
static double hash_target_rho(const wtp_ctx* ctx, int dim, int k, double radius, double rho_direct) {
    if (radius > 0) return 2.0;
    if (rho_direct > 0) return rho_direct < 1.0 ? 1.0 : rho_direct;
    const double r = (dim == 3 ? 0.381 : 0.436) * (double)(k > 0 ? k : 21) * (ctx->rho / 8.0);
    return r < 1.0 ? 1.0 : r;
}

// Quantile box: when a few far outliers stretch the bounding box so much that even the finest grid
// the caps allow (4096 cells per axis, 8 n cells) leaves the bulk of the cloud in a handful of cells,
// the grid is laid over the bulk only.  Per-axis histograms are zoomed (<= 4 rounds of 1024 bins)
// onto the range that holds all but 0.05 % of the points on either side; points outside are clamped
// into the edge cells, which every search treats as unbounded outward — results stay exact.
template <typename T>
static int find_robust_box(wtp_ctx* ctx, const Pt<T>* in, int64_t n, int dim, const Grid<T>& hg) {
    int rc;
    const size_t hist_bytes = sizeof(unsigned int) * 3 * 1024;
    if ((rc = ensure(ctx, ctx->box_dev, 64 + hist_bytes))) return rc;
    if ((rc = ensure_pinned(ctx, 64 + hist_bytes))) return rc;
    double box[6];
    for (int a = 0; a < 3; ++a) {
        box[a] = (double)hg.org[a];
        box[3 + a] = (double)hg.org[a] + (double)hg.n[a] * (double)hg.c;
    }
    const uint64_t tail = (uint64_t)(n / 2000) + 1; // 0.05 % per side
    for (int round = 0; round < 4; ++round) {
        memcpy(ctx->host_pinned, box, sizeof(box));
        WTP_HIP(ctx, hipMemcpyAsync(ctx->box_dev.p, ctx->host_pinned, sizeof(box), hipMemcpyHostToDevice, ctx->stream));
        unsigned int* d_hist = (unsigned int*)((char*)ctx->box_dev.p + 64);
        if ((rc = launch_axis_hist<T>(ctx, in, n, dim, (const double*)ctx->box_dev.p, d_hist))) return rc;
        WTP_HIP(ctx, hipMemcpyAsync((char*)ctx->host_pinned + 64, d_hist, hist_bytes, hipMemcpyDeviceToHost, ctx->stream));
        if ((rc = sync(ctx))) return rc;
        const unsigned int* h = (const unsigned int*)((const char*)ctx->host_pinned + 64);
        bool shrunk = false;
        for (int a = 0; a < dim; ++a) {
            const double w = (box[3 + a] - box[a]) / 1024.0;
            if (!(w > 0)) continue;
            uint64_t run = 0;
            int b_lo = 0, b_hi = 1023;
            for (int b = 0; b < 1024; ++b) {
                run += h[a * 1024 + b];
                if (run >= tail) { b_lo = b; break; }
            }
            run = 0;
            for (int b = 1023; b >= 0; --b) {
                run += h[a * 1024 + b];
                if (run >= tail) { b_hi = b; break; }
            }
            if (b_hi < b_lo) b_hi = b_lo;
            const double nlo = box[a] + w * b_lo, nhi = box[a] + w * (b_hi + 1);
            if ((nhi - nlo) < 0.5 * (box[3 + a] - box[a])) shrunk = true;
            box[a] = nlo;
            box[3 + a] = nhi;
        }
        if (!shrunk) break;
    }
    memcpy(ctx->host_pinned, box, sizeof(box));
    WTP_HIP(ctx, hipMemcpyAsync(ctx->box_dev.p, ctx->host_pinned, sizeof(box), hipMemcpyHostToDevice, ctx->stream));
    if ((rc = sync(ctx))) return rc;
    ctx->box_active = true;
    return WTP_OK;
}

template <typename T>
static int build_hash_tuned(wtp_ctx* ctx, const Pt<T>* in, Pt<T>* out, int64_t n, int dim, int k, double radius,
                            double rho_direct, double min_cell, double* scale_io, double* rho_eff_out, Grid<T>* hg_out) {
    const double target = hash_target_rho(ctx, dim, k, radius, rho_direct);
    double scale = *scale_io > 0 ? *scale_io : 1.0;
    double prev_c = -1;
    int rc;
    if ((rc = ensure(ctx, ctx->occ, 64))) return rc;
    if ((rc = ensure_pinned(ctx, 16384))) return rc;
    ctx->box_active = false; // every tuned build starts from the true bounding box
    bool boxed = false;
    for (int round = 0; round < 3; ++round) {
        if ((rc = build_hash<T>(ctx, in, out, n, dim, k, radius, rho_direct, min_cell, scale))) return rc;
        if ((rc = launch_occupancy(ctx, (unsigned long long*)ctx->occ.p))) return rc;
        char* hp = (char*)ctx->host_pinned;
        WTP_HIP(ctx, hipMemcpyAsync(hp, ctx->occ.p, 24, hipMemcpyDeviceToHost, ctx->stream));
        WTP_HIP(ctx, hipMemcpyAsync(hp + 64, ctx->grid.p, sizeof(Grid<T>), hipMemcpyDeviceToHost, ctx->stream));
        if ((rc = sync(ctx))) return rc;
        const unsigned long long* o = (const unsigned long long*)hp;
        const double rho_eff = o[1] ? (double)o[0] / (double)o[1] : 1.0;
        Grid<T> hg;
        memcpy(&hg, hp + 64, sizeof(hg));
        *rho_eff_out = rho_eff;
        *hg_out = hg;
        const double excess = (rho_eff - 1.0) / target;
        // the caps on the cell count bind (4096 per axis / 8 n) and the cells are still far over-full:
        // the box is stretched by outliers — lay the grid over the bulk and start over, once
        const bool capped = hg.n[0] >= kMaxAxisCells || hg.n[1] >= kMaxAxisCells || hg.n[2] >= kMaxAxisCells ||
                            (double)hg.ncells > 6.0 * (double)n;
        if (excess > 4.0 && capped && !boxed && radius <= 0) {
            if ((rc = find_robust_box<T>(ctx, in, n, dim, hg))) return rc;
            boxed = true;
            scale = 1.0;
            prev_c = -1;
            round = -1;
            continue;
        }
        if (!(excess > 1.6) || round == 2) break;
        if (prev_c > 0 && !((double)hg.c < prev_c * 0.999)) break; // a floor binds: shrinking changes nothing
        prev_c = (double)hg.c;
        double f = std::cbrt(1.15 / excess); // occupancy ~ c^3 (c^2 on surfaces: the next round catches up)
        if (f < 0.3) f = 0.3;
        scale *= f;
        if (scale < 0.02) scale = 0.02;
    }
    *scale_io = scale;
    return WTP_OK;
}

// ---- topology -----------------------------------------------------------------------------------
// Brick geometry of wtp_ksel.hip from the measured grid: the brick length along x that puts ~224 queries on the 256
// lanes (four own cells per column; the denser of the box average and the occupancy the points see), and the LDS
// point area for its halo of 36 cells per column plus five standard deviations.
static void ksel_geometry(wtp_ctx* ctx, double n, double ncells, int n0, double rho_eff, int* bx_out, int* hcap_out) {
    double rho_cell = ncells > 0 ? n / ncells : 1.0;
    if (rho_eff - 1.0 > rho_cell) rho_cell = rho_eff - 1.0;
    if (rho_cell < 0.05) rho_cell = 0.05;
    // Bricks of equal length along x: among the splits of the n0 columns, the one with the lowest expected cost per
    // query — a brick costs one round of the 256 lanes, two when its Q own points (Poisson) exceed them
    int bx = ksel_max_bx();
    double best = 1e300;
    for (int nbx = 1; nbx <= n0; ++nbx) {
        const int b = (n0 + nbx - 1) / nbx;
        if (b > ksel_max_bx()) continue;
        const double q = 4.0 * rho_cell * b;
        const double p2 = 0.5 * std::erfc((256.0 - q) / std::sqrt(2.0 * (q > 1 ? q : 1)));
        const double cost = (1.0 + p2 + (q > 512.0 ? 100.0 : 0.0)) / q;
        if (cost < best) {
            best = cost;
            bx = b;
        }
        if (b < 8) break;
    }
    bx = bx < 8 ? (n0 < 8 ? (n0 > 0 ? n0 : 1) : 8) : bx;
    const double halo = 36.0 * (bx + 4) * rho_cell;
    int hc = (int)(halo + 5.0 * std::sqrt(halo)) + 32;
    hc = (hc + 63) / 64 * 64;
    *bx_out = bx;
    *hcap_out = hc < 512 ? 512 : (hc > 3072 ? 3072 : hc);
    if (getenv("WTP_DEBUG"))
        fprintf(stderr, "[wtp] ksel geometry: rho_cell %.3f (box average %.3f), %d columns -> bricks of %d, LDS point area %d\n",
                rho_cell, ncells > 0 ? n / ncells : 0.0, n0, *bx_out, *hcap_out);
}
// The occupancy (points per cell) that serves THIS cloud best, between 1.08 and 1.26 times k/22: the x axis holds a whole
// number of equal bricks, so the lane fill of a brick steps with the number of columns (at 10 M points 1.2 leaves bricks of
// 41 columns, 77 % of the lanes; 1.1 fills 90 %).  Model per query: (one brick round, two with probability p2) / Q own
// points, times the round's cost (a scan in proportion to the occupancy on top of a fixed part), plus the hand-backs that
// grow as the provable radius 2c shrinks.  `n0`, `c`: the grid just built with occupancy rho_cur.
static double ksel_pick_rho(const wtp_ctx* ctx, double n, double ncells, int n0, double rho_cur, double rho_eff) {
    if (getenv("WTP_RHO_KSEL")) return rho_cur; // the caller fixed it
    double fill_cur = ncells > 0 ? n / ncells : rho_cur; // points per cell, box average
    if (rho_eff - 1.0 > fill_cur) fill_cur = rho_eff - 1.0;
    double best = 1e300, best_rho = rho_cur;
    for (double f = 0.90; f <= 1.051; f += 0.0125) { // candidate occupancy = f * rho_cur
        if (rho_cur * f > 1.27) continue;                 // (a run of 173 cells must fit the 256 slots of the hit masks)
        const double edge = std::cbrt(f);                 // cell edge relative to the current one
        const int cols = (int)(((double)n0 - 0.5) / edge) + 1;
        const double rho_cell = fill_cur * f;
        double cost_b = 1e300;
        for (int nbx = 1; nbx <= cols; ++nbx) {
            const int b = (cols + nbx - 1) / nbx;
            if (b > ksel_max_bx()) continue;
            const double q = 4.0 * rho_cell * b;
            const double p2 = 0.5 * std::erfc((256.0 - q) / std::sqrt(2.0 * (q > 1 ? q : 1)));
            const double cst = (1.0 + p2 + (q > 512.0 ? 100.0 : 0.0)) / q;
            cost_b = cst < cost_b ? cst : cost_b;
            if (b < 8) break;
        }
        const double round = 0.63 + 0.37 * f;                       // per-round work: fixed part + scan
        const double handback = 1.0 + 0.20 * (1.0 - f) / 0.1 * 0.1;   // ~2 % more total time per 10 % less occupancy
        const double cost = cost_b * round * handback;
        if (cost < best) {
            best = cost;
            best_rho = rho_cur * f;
        }
    }
    return best_rho;
}

// occupancy of the wtp_ksel.hip grid for kq = k + self: in proportion to kq (the cell edge follows r_k), capped where a run of
// 173 cells still fits the 256 slots of the hit masks
static double ksel_rho_for(const wtp_ctx* ctx, int kq) {
    const double rho = ctx->rho_ksel * (double)kq / 22.0;
    return rho > 1.26 ? 1.26 : rho;
}
// points the first filter ball is expected to hold: k + self plus the same number of standard deviations as
// cap_ksel leaves at 22
static double ksel_cap_count(const wtp_ctx* ctx, int kq) {
    return (double)kq + (ctx->cap_ksel - 22.0) / std::sqrt(22.0) * std::sqrt((double)kq);
}

template <typename T>
static int knn_dev_t(wtp_ctx* ctx, const T* d_xyz, int64_t n, int dim, int k, int include_self,
                     int32_t* d_idx, T* d_dist) {
    int rc;
    if ((rc = ensure(ctx, ctx->pts[0], sizeof(Pt<T>) * (size_t)n))) return rc;
    if ((rc = ensure(ctx, ctx->pts[1], sizeof(Pt<T>) * (size_t)n))) return rc;
    if ((rc = ensure(ctx, ctx->fb_list, sizeof(int32_t) * (size_t)n))) return rc;
    if ((rc = ensure(ctx, ctx->fb_count, 64))) return rc;
    ctx->counters_clean = false; // (this call counts in the block; the next sweep clears it itself)
    if ((rc = ensure(ctx, ctx->fb2_list, sizeof(int32_t) * (size_t)n))) return rc;
    if ((rc = ensure(ctx, ctx->fb2_count, 64))) return rc;
    Pt<T>* raw = (Pt<T>*)ctx->pts[0].p;
    Pt<T>* sorted = (Pt<T>*)ctx->pts[1].p;
    int sp = span_begin(ctx, 0);
    if ((rc = load_points<T>(ctx, d_xyz, raw, n, dim))) return rc;
    // neighbours sought per query inside the structure: k others + self
    // The measured cell scale of the last topology call is reused for a cloud of the same size (the
    // usual case: rebuild_topology! on the same points); it only affects speed, never the result.
    const int kq = include_self ? k : k + 1;
    // fp32 3-D clouds with k + self <= 24 (the reference's k = 21 among them): the x-slowest layout of wtp_ksel.hip —
    // cells of ~1.2 points, the k nearest inside the 5 x 5 x 5 block around the query's cell
    const bool ksel = sizeof(T) == 4 && dim == 3 && ctx->ksel && !ctx->force_generic && kq <= ksel_kmax() && n >= 4096;
    const double rho_direct = ksel ? ksel_rho_for(ctx, kq) : 0.0;
    ctx->topology_build = true;
    if (ctx->knn_tune_n == n && ctx->knn_tune_dim == dim && ctx->knn_tune_k == kq && !ctx->knn_tune_boxed &&
        ctx->knn_tune_ksel == (int)ksel) {
        ctx->box_active = false;
        if ((rc = build_hash<T>(ctx, raw, sorted, n, dim, kq, 0.0, ksel ? ctx->knn_tune_rho : 0.0, 0.0, ctx->knn_tune_scale))) {
            ctx->topology_build = false;
            return rc;
        }
    } else {
        double scale = 1.0, rho_eff = 0;
        Grid<T> hg;
        if ((rc = build_hash_tuned<T>(ctx, raw, sorted, n, dim, kq, 0.0, rho_direct, 0.0, &scale, &rho_eff, &hg))) {
            ctx->topology_build = false;
            return rc;
        }
        ctx->knn_tune_rho = rho_direct;
        if (ksel) { // the occupancy whose grid fills the bricks' lanes best: one more measured build, this call only
            const double pick = ksel_pick_rho(ctx, (double)n, (double)hg.ncells, hg.n[0], rho_direct, rho_eff);
            if (std::fabs(pick - rho_direct) > 0.01 * rho_direct) {
                scale = 1.0;
                if ((rc = build_hash_tuned<T>(ctx, raw, sorted, n, dim, kq, 0.0, pick, 0.0, &scale, &rho_eff, &hg))) {
                    ctx->topology_build = false;
                    return rc;
                }
                ctx->knn_tune_rho = pick;
            }
        }
        ctx->knn_tune_n = n;
        ctx->knn_tune_dim = dim;
        ctx->knn_tune_k = kq;
        ctx->knn_tune_scale = scale;
        ctx->knn_tune_boxed = ctx->box_active; // a clipped box belongs to this very cloud: never reuse it
        ctx->knn_tune_ksel = (int)ksel;
        if (ksel) ksel_geometry(ctx, (double)n, (double)hg.ncells, hg.n[0], rho_eff, &ctx->knn_tune_bx, &ctx->knn_tune_hcap);
    }
    ctx->topology_build = false;
    span_end(ctx, sp);
    SearchArgs<T> a{};
    a.grid = (const Grid<T>*)ctx->grid.p;
    a.snap = sorted;
    a.query = sorted;
    a.cell_start = (const int32_t*)ctx->cell_start.p;
    a.n = (int32_t)n;
    a.k = k;
    a.include_self = include_self;
    a.idx_out = d_idx;
    a.dist_out = d_dist;
    a.fb_list = (int32_t*)ctx->fb_list.p;
    a.fb_count = (int32_t*)ctx->fb_count.p;
    a.fb2_list = (int32_t*)ctx->fb2_list.p;
    // one 64-byte counter block, cleared once: [0] hand-backs of the brick kernel, [4] of the wave kernel
    a.fb2_count = (int32_t*)ctx->fb_count.p + 4;
    WTP_HIP(ctx, hipMemsetAsync(ctx->fb_count.p, 0, 64, ctx->stream));
    a.counters_cleared = 1;
    if ((rc = ensure(ctx, ctx->diag, 128))) return rc;
    a.diag = (unsigned long long*)ctx->diag.p; // (written by -DWTP_DIAG builds only)
    if (ksel) {
        a.ksel_bx = ctx->knn_tune_bx;
        a.brick_hcap = ctx->knn_tune_hcap;
        a.cap_count = (float)ksel_cap_count(ctx, kq);
    }
    sp = span_begin(ctx, 1);
    rc = launch_topology<T>(ctx, a);
    span_end(ctx, sp);
    if (!rc && getenv("WTP_DEBUG")) { // hand-backs of the brick kernel to the exact path
        int32_t h[2] = {0, 0};
        hipMemcpyAsync(&h[0], a.fb_count, 4, hipMemcpyDeviceToHost, ctx->stream);
        hipMemcpyAsync(&h[1], a.fb2_count, 4, hipMemcpyDeviceToHost, ctx->stream);
        hipStreamSynchronize(ctx->stream);
        fprintf(stderr, "[wtp] knn n=%lld k=%d: %d queries to the wave kernel, %d to the serial one\n", (long long)n, k, h[0], h[1]);
    }
    ctx->n_sweep_launches += 1;
    ctx->relax.have_tree = false; // pts[] reused
    ctx->rad_valid = false;
    return rc;
}

// fp64 clouds, KNNTopology: Float64 is the reference's default type, and the exact wave-per-query
// path costs 16 ns per query.  Faster and still exact: search CANDIDATES in fp32 (the cloud moved to
// its own origin and rounded to float; the k+2 nearest per query through the fp32 brick kernel),
// re-rank them in exact fp64, and certify per query that nothing outside the candidate list can
// belong to the answer (refine_f64_kernel).  Queries that fail the certificate — coincident
// clusters larger than the list, clouds whose extent/spacing ratio exhausts float — take the exact
// path.  Returns 1 in *done when it handled the call.
static int knn_dev_f64(wtp_ctx* ctx, const double* d_xyz, int64_t n, int dim, int k, int include_self, int32_t* d_idx,
                       double* d_dist, bool* done) {
    *done = false;
    const int kq = include_self ? k : k + 1;
    // two candidates beyond the kq wanted: the certificate needs ONE whose fp32 distance clears the exact
    // kq-th by more than the rounding bound (gaps between consecutive neighbour distances are ~1e-2 of
    // the distance, the bound ~1e-6), and longer lists overflow the brick kernel's 64-entry ring
    int kc = kq + 2;
    if ((int64_t)kc > n) kc = (int)n;
    if (ctx->force_generic || kc > 31) return WTP_OK; // beyond the fp32 brick kernel's list length: exact path
    int rc;
    if ((rc = ensure(ctx, ctx->pts[0], sizeof(double4) * (size_t)n))) return rc;
    if ((rc = ensure(ctx, ctx->pts[1], sizeof(double4) * (size_t)n))) return rc;
    if ((rc = ensure(ctx, ctx->f32_pts, 2 * sizeof(float4) * (size_t)n))) return rc;
    if ((rc = ensure(ctx, ctx->cand_idx, sizeof(int32_t) * (size_t)n * kc))) return rc;
    if ((rc = ensure(ctx, ctx->cand_dist, sizeof(float) * (size_t)n * kc))) return rc;
    if ((rc = ensure(ctx, ctx->fb_list, sizeof(int32_t) * (size_t)n))) return rc;
    if ((rc = ensure(ctx, ctx->fb_count, 64))) return rc;
    ctx->counters_clean = false; // (this call counts in the block; the next sweep clears it itself)
    if ((rc = ensure(ctx, ctx->fb2_list, sizeof(int32_t) * (size_t)n))) return rc;
    if ((rc = ensure(ctx, ctx->fb2_count, 64))) return rc;
    if ((rc = ensure(ctx, ctx->occ, 64))) return rc;
    double4* raw64 = (double4*)ctx->pts[0].p;
    float4* raw32 = (float4*)ctx->f32_pts.p;
    float4* sorted32 = raw32 + n;
    double* org4 = (double*)ctx->occ.p + 4; // behind the occupancy counters
    int sp = span_begin(ctx, 0);
    if ((rc = load_points<double>(ctx, d_xyz, raw64, n, dim))) return rc;
    if ((rc = launch_origin(ctx, raw64, n, org4))) return rc;
    if ((rc = launch_to_local_f32(ctx, raw64, n, org4, raw32))) return rc;
    double scale = 1.0, rho_eff = 0;
    Grid<float> hg;
    // the fp32 candidate search takes the x-slowest layout of wtp_ksel.hip where that applies (3-D, k + self + 2 <= 24)
    const bool ksel = dim == 3 && ctx->ksel && kc <= ksel_kmax() && n >= 4096;
    double rho_direct = ksel ? ksel_rho_for(ctx, kc) : 0.0;
    ctx->topology_build = true;
    // the measured cell scale (and the brick geometry that goes with it) of the last fp64 call is reused for a cloud of the
    // same size, as in the fp32 calls: it only affects speed, and saves two occupancy passes and a host synchronisation
    const bool cached = ctx->knn64_tune_n == n && ctx->knn64_tune_dim == dim && ctx->knn64_tune_k == kc &&
                        ctx->knn64_tune_ksel == (int)ksel;
    if (cached) {
        ctx->box_active = false;
        rc = build_hash<float>(ctx, raw32, sorted32, n, dim, kc, 0.0, ksel ? ctx->knn64_tune_rho : 0.0, 0.0, ctx->knn64_tune_scale);
    } else {
        rc = build_hash_tuned<float>(ctx, raw32, sorted32, n, dim, kc, 0.0, rho_direct, 0.0, &scale, &rho_eff, &hg);
        if (!rc && ksel) {
            const double pick = ksel_pick_rho(ctx, (double)n, (double)hg.ncells, hg.n[0], rho_direct, rho_eff);
            if (std::fabs(pick - rho_direct) > 0.01 * rho_direct) {
                rho_direct = pick;
                scale = 1.0;
                rc = build_hash_tuned<float>(ctx, raw32, sorted32, n, dim, kc, 0.0, rho_direct, 0.0, &scale, &rho_eff, &hg);
            }
        }
        if (!rc && !ctx->box_active) { // (a clipped box belongs to this very cloud: never reused)
            ctx->knn64_tune_n = n;
            ctx->knn64_tune_dim = dim;
            ctx->knn64_tune_k = kc;
            ctx->knn64_tune_ksel = (int)ksel;
            ctx->knn64_tune_scale = scale;
            ctx->knn64_tune_rho = rho_direct;
            if (ksel) ksel_geometry(ctx, (double)n, (double)hg.ncells, hg.n[0], rho_eff, &ctx->knn64_tune_bx, &ctx->knn64_tune_hcap);
        } else {
            ctx->knn64_tune_n = -1;
        }
    }
    ctx->topology_build = false;
    if (rc) return rc;
    span_end(ctx, sp);
    ctx->knn_tune_n = -1; // the cached scale belongs to fp32 calls
    SearchArgs<float> a{};
    a.grid = (const Grid<float>*)ctx->grid.p;
    a.snap = sorted32;
    a.query = sorted32;
    a.cell_start = (const int32_t*)ctx->cell_start.p;
    a.n = (int32_t)n;
    a.k = kc;
    a.include_self = 1;
    a.idx_out = (int32_t*)ctx->cand_idx.p;
    a.dist_out = (float*)ctx->cand_dist.p;
    a.fb_list = (int32_t*)ctx->fb_list.p;
    a.fb_count = (int32_t*)ctx->fb_count.p;
    a.fb2_list = (int32_t*)ctx->fb2_list.p;
    a.fb2_count = (int32_t*)ctx->fb2_count.p;
    if (ksel) {
        if (ctx->knn64_tune_n != n) // (a clipped box: geometry of this very build)
            ksel_geometry(ctx, (double)n, (double)hg.ncells, hg.n[0], rho_eff, &ctx->knn64_tune_bx, &ctx->knn64_tune_hcap);
        a.ksel_bx = ctx->knn64_tune_bx;
        a.brick_hcap = ctx->knn64_tune_hcap;
        a.cap_count = (float)ksel_cap_count(ctx, kc);
    }
    sp = span_begin(ctx, 1);
    // k = 21 without self (kc = 24): search and re-ranking in slot order (see refine_f64_slots_kernel)
    const bool slots = kc == 24 && !ctx->force_generic && getenv("WTP_F64_SLOTS_OFF") == nullptr;
    double4* slot64 = (double4*)ctx->pts[1].p;
    if (slots && (rc = launch_relabel_slots(ctx, raw64, sorted32, slot64, n))) return rc;
    rc = launch_topology<float>(ctx, a);
    if (rc) return rc;
    if (slots)
        rc = launch_refine_f64_slots(ctx, slot64, a.idx_out, a.dist_out, n, kc, k, include_self, org4, d_idx, d_dist,
                                     (int32_t*)ctx->fb_list.p, (int32_t*)ctx->fb_count.p);
    else
        rc = launch_refine_f64(ctx, raw64, a.idx_out, a.dist_out, n, kc, k, include_self, org4, d_idx, d_dist,
                               (int32_t*)ctx->fb_list.p, (int32_t*)ctx->fb_count.p);
    span_end(ctx, sp);
    if (rc) return rc;
    ctx->n_sweep_launches += 1;
    ctx->relax.have_tree = false;
    ctx->rad_valid = false;
    if ((rc = ensure_pinned(ctx, 64))) return rc;
    WTP_HIP(ctx, hipMemcpyAsync(ctx->host_pinned, ctx->fb_count.p, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = sync(ctx))) return rc;
    const int32_t n_fail = *(const int32_t*)ctx->host_pinned;
    if (n_fail > 0) { // exact fp64 path for the uncertified queries: wave kernel over their ids, fp64 grid
        double4* sorted64 = (double4*)ctx->pts[1].p;
        ctx->box_active = false;
        ctx->topology_build = true; // rows are ordered by (d2, id) explicitly: no canonical-order pass (0.5 ms on unsorted input)
        rc = build_hash<double>(ctx, raw64, sorted64, n, dim, kq, 0.0);
        ctx->topology_build = false;
        if (rc) return rc;
        SearchArgs<double> b{};
        b.grid = (const Grid<double>*)ctx->grid.p;
        b.snap = sorted64;
        b.query = raw64; // list entries are ids: raw64[id] is the query, its w the id
        b.cell_start = (const int32_t*)ctx->cell_start.p;
        b.n = (int32_t)n;
        b.k = k;
        b.include_self = include_self;
        b.idx_out = d_idx;
        b.dist_out = d_dist;
        b.fb_list = (int32_t*)ctx->fb_list.p;
        b.fb_count = (int32_t*)ctx->fb_count.p;
        b.fb2_list = (int32_t*)ctx->fb2_list.p;
        b.fb2_count = (int32_t*)ctx->fb2_count.p;
        if ((rc = launch_generic_topology<double>(ctx, b, false))) return rc;
    }
    *done = true;
    return WTP_OK;
}

static int check_idle(wtp_ctx* ctx) {
    if (ctx->relax.active)
        return fail(ctx, WTP_ERR_STATE,
                    "context holds a relax session (its buffers are live): call wtp_relax_end or use another context");
    return WTP_OK;
}

static int check_k(wtp_ctx* ctx, int64_t n, int k, int include_self) {
    if (k < 1) return fail(ctx, WTP_ERR_ARG, "k must be >= 1");
    if ((int64_t)k > n - (include_self ? 0 : 1))
        return fail(ctx, WTP_ERR_ARG, "k exceeds the number of available neighbours (k+1 > n)");
    if (k > kGenericKMax) return fail(ctx, WTP_ERR_ARG, "k > 128 is not supported");
    return WTP_OK;
}

} // namespace wtp

using namespace wtp;

#define WTP_API extern "C"

WTP_API const char* wtp_version(void) { return "wtp-mi355x 0.1.0 (gfx950)"; }

WTP_API const char* wtp_last_error(const wtp_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

WTP_API int wtp_create(const int* device_ordinals, int n_dev, wtp_ctx** out) {
    if (!out) return fail(nullptr, WTP_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (n_dev != 1)
        return fail(nullptr, WTP_ERR_ARG, "one context drives one GPU: create one context per process/GPU");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1) {
        (void)hipGetLastError();
        return fail(nullptr, WTP_ERR_NO_DEVICE, "no HIP device visible (libwtp has no CPU path)");
    }
    int dev = device_ordinals ? device_ordinals[0] : 0;
    if (dev < 0 || dev >= count) return fail(nullptr, WTP_ERR_ARG, "device ordinal out of range");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess)
        return fail(nullptr, WTP_ERR_HIP, "hipGetDeviceProperties failed");
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(nullptr, WTP_ERR_NO_DEVICE,
                    std::string("device is ") + prop.gcnArchName + ", libwtp is built for gfx950 only");
    if (hipSetDevice(dev) != hipSuccess) return fail(nullptr, WTP_ERR_HIP, "hipSetDevice failed");
    wtp_ctx* ctx = new wtp_ctx();
    ctx->device = dev;
    ctx->sm_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return fail(nullptr, WTP_ERR_HIP, "hipStreamCreate failed");
    }
    ctx->stream = ctx->own_stream;
    if (const char* e = getenv("WTP_RHO")) ctx->rho = atof(e) > 0 ? atof(e) : ctx->rho;
    if (const char* e = getenv("WTP_GAMMA_CAP")) ctx->gamma_cap = atof(e) > 0 ? atof(e) : ctx->gamma_cap;
    if (const char* e = getenv("WTP_GAMMA_CAP_SWEEP")) ctx->gamma_cap_sweep = atof(e) > 0 ? atof(e) : ctx->gamma_cap_sweep;
    if (const char* e = getenv("WTP_TNN")) ctx->tnn_frac = atof(e) > 0 && atof(e) < 0.99 ? atof(e) : ctx->tnn_frac;
    if (const char* e = getenv("WTP_FORCE_GENERIC")) ctx->force_generic = atoi(e);
    if (const char* e = getenv("WTP_FULL_SELECT")) ctx->full_select = atoi(e);
    if (const char* e = getenv("WTP_CS2")) ctx->cs2 = atoi(e);
    if (const char* e = getenv("WTP_KSEL")) ctx->ksel = atoi(e);
    if (const char* e = getenv("WTP_F64_KSEL")) ctx->f64_ksel = atoi(e);
    if (const char* e = getenv("WTP_BALL64")) ctx->ball64 = atoi(e);
    if (const char* e = getenv("WTP_RHO_KSEL")) ctx->rho_ksel = atof(e) > 0 ? atof(e) : ctx->rho_ksel;
    if (const char* e = getenv("WTP_CAP_KSEL")) ctx->cap_ksel = atof(e) > 0 ? atof(e) : ctx->cap_ksel;
    if (const char* e = getenv("WTP_RHO_CS")) ctx->rho_cs2 = atof(e) >= 1.0 ? atof(e) : ctx->rho_cs2;
    if (const char* e = getenv("WTP_STYP_SIGMA")) ctx->styp_sigma = atof(e);
    if (const char* e = getenv("WTP_TIMING")) {
        ctx->timing = atoi(e) != 0;
        ctx->timing_forced = true;
    }
    if (const char* e = getenv("WTP_MESH_PACKET")) ctx->mesh_packet = atoi(e);
    if (const char* e = getenv("WTP_GRID_REUSE")) ctx->grid_reuse_max = atoi(e) >= 0 ? atoi(e) : ctx->grid_reuse_max;
    *out = ctx;
    return WTP_OK;
}

WTP_API int wtp_destroy(wtp_ctx* ctx) {
    if (ctx) block_destroy(ctx);
    if (!ctx) return WTP_OK;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    wtp_comm_finalize(ctx);
    DevBuf* bufs[] = {&ctx->pts[0], &ctx->pts[1], &ctx->pts[2], &ctx->raw_in, &ctx->cell_of, &ctx->rank_of,
                      &ctx->cell_cnt, &ctx->cell_start, &ctx->scan_tmp, &ctx->grid, &ctx->bbox_part,
                      &ctx->idx_out, &ctx->dist_out, &ctx->counts_out, &ctx->forces, &ctx->nn_dist,
                      &ctx->nn_id, &ctx->spacing_pp, &ctx->partials, &ctx->stats, &ctx->fb_list,
                      &ctx->fb_count, &ctx->fb2_list, &ctx->fb2_count, &ctx->nn_list, &ctx->brick_dead, &ctx->rad_tmp, &ctx->rad_done, &ctx->rad_arena, &ctx->rad_arena_off, &ctx->rad_pos, &ctx->rad_bricks, &ctx->grid_b, &ctx->cell_start_b, &ctx->f64k_s64, &ctx->f64k_slot, &ctx->f64k_lists, &ctx->f64k_cnt, &ctx->scratch, &ctx->diag,
                      &ctx->ins_in, &ctx->ins_elems, &ctx->ins_partial, &ctx->ins_out, &ctx->mesh_nodes, &ctx->mesh_pn, &ctx->mesh_io,
                      &ctx->wall_flags, &ctx->wall_tri, &ctx->wall_hint, &ctx->mesh_cls, &ctx->kd_nodes, &ctx->sp_hint, &ctx->occ, &ctx->box_dev,
                      &ctx->cand_idx, &ctx->cand_dist, &ctx->f32_pts, &ctx->comm_scratch, &ctx->sp_cert};
    for (DevBuf* b : bufs)
        if (b->p) hipFree(b->p);
    if (ctx->host_pinned) hipHostFree(ctx->host_pinned);
    for (auto e : ctx->ev_pool) hipEventDestroy(e);
    if (ctx->ev_comm_a) hipEventDestroy(ctx->ev_comm_a);
    if (ctx->ev_comm_b) hipEventDestroy(ctx->ev_comm_b);
    if (ctx->comm_stream) hipStreamDestroy(ctx->comm_stream);
    hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return WTP_OK;
}

WTP_API int wtp_knn_dev(wtp_ctx* ctx, const void* d_xyz, int64_t n, int dim, int dtype, int k, int include_self,
                int32_t* d_idx_out, void* d_dist_out) {
    int rc = check_cloud(ctx, d_xyz, n, dim, dtype);
    if (rc) return rc;
    if ((rc = check_k(ctx, n, k, include_self))) return rc;
    if ((rc = check_idle(ctx))) return rc;
    if (!d_idx_out) return fail(ctx, WTP_ERR_ARG, "idx_out is NULL");
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    if (dtype == WTP_F32) {
        rc = knn_dev_t<float>(ctx, (const float*)d_xyz, n, dim, k, include_self, d_idx_out, (float*)d_dist_out);
    } else {
        bool done = false;
        rc = knn_dev_f64(ctx, (const double*)d_xyz, n, dim, k, include_self, d_idx_out, (double*)d_dist_out, &done);
        if (!rc && !done)
            rc = knn_dev_t<double>(ctx, (const double*)d_xyz, n, dim, k, include_self, d_idx_out, (double*)d_dist_out);
    }
    if (rc) return rc;
    return sync(ctx);
}

WTP_API int wtp_knn(wtp_ctx* ctx, const void* xyz, int64_t n, int dim, int dtype, int k, int include_self,
            int32_t* idx_out, void* dist_out) {
    int rc = check_cloud(ctx, xyz, n, dim, dtype);
    if (rc) return rc;
    if ((rc = check_k(ctx, n, k, include_self))) return rc;
    if ((rc = check_idle(ctx))) return rc;
    if (!idx_out) return fail(ctx, WTP_ERR_ARG, "idx_out is NULL");
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t ts = tsize(dtype);
    if ((rc = ensure(ctx, ctx->raw_in, ts * (size_t)n * dim))) return rc;
    if ((rc = ensure(ctx, ctx->idx_out, sizeof(int32_t) * (size_t)n * k))) return rc;
    if (dist_out && (rc = ensure(ctx, ctx->dist_out, ts * (size_t)n * k))) return rc;
    WTP_HIP(ctx, hipMemcpyAsync(ctx->raw_in.p, xyz, ts * (size_t)n * dim, hipMemcpyHostToDevice, ctx->stream));
    void* ddist = dist_out ? ctx->dist_out.p : nullptr;
    if (dtype == WTP_F32)
        rc = knn_dev_t<float>(ctx, (const float*)ctx->raw_in.p, n, dim, k, include_self, (int32_t*)ctx->idx_out.p,
                              (float*)ddist);
    else {
        bool done = false;
        rc = knn_dev_f64(ctx, (const double*)ctx->raw_in.p, n, dim, k, include_self, (int32_t*)ctx->idx_out.p,
                         (double*)ddist, &done);
        if (!rc && !done)
            rc = knn_dev_t<double>(ctx, (const double*)ctx->raw_in.p, n, dim, k, include_self,
                                   (int32_t*)ctx->idx_out.p, (double*)ddist);
    }
    if (rc) return rc;
    WTP_HIP(ctx, hipMemcpyAsync(idx_out, ctx->idx_out.p, sizeof(int32_t) * (size_t)n * k, hipMemcpyDeviceToHost,
                                ctx->stream));
    if (dist_out)
        WTP_HIP(ctx, hipMemcpyAsync(dist_out, ctx->dist_out.p, ts * (size_t)n * k, hipMemcpyDeviceToHost, ctx->stream));
    return sync(ctx);
}

// ---- consumers of the rows ------------------------------------------------------
// rows (self included) and, if wanted, distances of a host cloud, left in ctx->idx_out / dist_out
static int knn_rows_on_device(wtp_ctx* ctx, const void* xyz, int64_t n, int dim, int dtype, int k, bool want_dist) {
    const size_t ts = tsize(dtype);
    int rc;
    if ((rc = ensure(ctx, ctx->raw_in, ts * (size_t)n * dim))) return rc;
    if ((rc = ensure(ctx, ctx->idx_out, sizeof(int32_t) * (size_t)n * k))) return rc;
    if (want_dist && (rc = ensure(ctx, ctx->dist_out, ts * (size_t)n * k))) return rc;
    WTP_HIP(ctx, hipMemcpyAsync(ctx->raw_in.p, xyz, ts * (size_t)n * dim, hipMemcpyHostToDevice, ctx->stream));
    void* ddist = want_dist ? ctx->dist_out.p : nullptr;
    if (dtype == WTP_F32)
        return knn_dev_t<float>(ctx, (const float*)ctx->raw_in.p, n, dim, k, 1, (int32_t*)ctx->idx_out.p, (float*)ddist);
    bool done = false;
    rc = knn_dev_f64(ctx, (const double*)ctx->raw_in.p, n, dim, k, 1, (int32_t*)ctx->idx_out.p, (double*)ddist, &done);
    if (!rc && !done)
        rc = knn_dev_t<double>(ctx, (const double*)ctx->raw_in.p, n, dim, k, 1, (int32_t*)ctx->idx_out.p, (double*)ddist);
    return rc;
}

WTP_API int wtp_pca_normals(wtp_ctx* ctx, const void* xyz, int64_t n, int dim, int dtype, int k, void* normals_out) {
    int rc = check_cloud(ctx, xyz, n, dim, dtype);
    if (rc) return rc;
    if (k < 2) return fail(ctx, WTP_ERR_ARG, "k must be >= 2 (a covariance needs two points)");
    if ((rc = check_k(ctx, n, k, 1))) return rc;
    if ((rc = check_idle(ctx))) return rc;
    if (!normals_out) return fail(ctx, WTP_ERR_ARG, "normals_out is NULL");
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t ts = tsize(dtype);
    if ((rc = knn_rows_on_device(ctx, xyz, n, dim, dtype, k, false))) return rc;
    if ((rc = ensure(ctx, ctx->scratch, ts * (size_t)n * dim))) return rc;
    int sp = span_begin(ctx, 2);
    rc = dtype == WTP_F32 ? launch_pca_normals<float>(ctx, (const float*)ctx->raw_in.p, n, dim, (const int32_t*)ctx->idx_out.p,
                                                      k, (float*)ctx->scratch.p)
                          : launch_pca_normals<double>(ctx, (const double*)ctx->raw_in.p, n, dim,
                                                       (const int32_t*)ctx->idx_out.p, k, (double*)ctx->scratch.p);
    span_end(ctx, sp);
    if (rc) return rc;
    WTP_HIP(ctx, hipMemcpyAsync(normals_out, ctx->scratch.p, ts * (size_t)n * dim, hipMemcpyDeviceToHost, ctx->stream));
    return sync(ctx);
}

WTP_API int wtp_gradient_limit(wtp_ctx* ctx, const void* centers, int64_t n, int dim, int dtype, int k, const void* h0,
                               double g, double tol, int max_sweeps, void* h_out, int* sweeps_out) {
    int rc = check_cloud(ctx, centers, n, dim, dtype);
    if (rc) return rc;
    if ((rc = check_k(ctx, n, k, 1))) return rc;
    if ((rc = check_idle(ctx))) return rc;
    if (!h0 || !h_out) return fail(ctx, WTP_ERR_ARG, "NULL array");
    if (max_sweeps < 0) return fail(ctx, WTP_ERR_ARG, "max_sweeps must be >= 0");
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t ts = tsize(dtype);
    if ((rc = knn_rows_on_device(ctx, centers, n, dim, dtype, k, true))) return rc;
    const size_t o1 = (ts * (size_t)n + 255) / 256 * 256;
    if ((rc = ensure(ctx, ctx->scratch, 2 * o1 + 64))) return rc;
    char* b = (char*)ctx->scratch.p;
    unsigned long long* st = (unsigned long long*)(b + 2 * o1);
    WTP_HIP(ctx, hipMemcpyAsync(b, h0, ts * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    WTP_HIP(ctx, hipMemsetAsync(st, 0, 64, ctx->stream));
    if ((rc = ensure_pinned(ctx, 64))) return rc;
    unsigned long long* hst = (unsigned long long*)ctx->host_pinned;
    hst[0] = hst[1] = 0;
    int sp = span_begin(ctx, 2);
    for (int first = 0; first < max_sweeps && !hst[0];) { // batches: one read-back per 16 sweeps
        const int batch = max_sweeps - first < 16 ? max_sweeps - first : 16;
        rc = dtype == WTP_F32
                 ? launch_minplus_batch<float>(ctx, (const int32_t*)ctx->idx_out.p, (const float*)ctx->dist_out.p, n, k, g,
                                               tol, (float*)b, (float*)(b + o1), first, batch, st)
                 : launch_minplus_batch<double>(ctx, (const int32_t*)ctx->idx_out.p, (const double*)ctx->dist_out.p, n, k,
                                                g, tol, (double*)b, (double*)(b + o1), first, batch, st);
        if (rc) return rc;
        WTP_HIP(ctx, hipMemcpyAsync(hst, st, 16, hipMemcpyDeviceToHost, ctx->stream));
        if ((rc = sync(ctx))) return rc;
        first += batch;
    }
    span_end(ctx, sp);
    const int applied = (int)hst[1];
    if (sweeps_out) *sweeps_out = applied;
    WTP_HIP(ctx, hipMemcpyAsync(h_out, b + ((applied & 1) ? o1 : 0), ts * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    return sync(ctx);
}

// ---- RadiusTopology ------------------------------------------------------------------------------
template <typename T> static int radius_count_t(wtp_ctx* ctx, int64_t n, int dim, double r, int32_t* d_counts) {
    int rc;
    if ((rc = ensure(ctx, ctx->pts[0], sizeof(Pt<T>) * (size_t)n))) return rc;
    if ((rc = ensure(ctx, ctx->pts[1], sizeof(Pt<T>) * (size_t)n))) return rc;
    Pt<T>* raw = (Pt<T>*)ctx->pts[0].p;
    Pt<T>* sorted = (Pt<T>*)ctx->pts[1].p;
    int sp = span_begin(ctx, 0);
    if ((rc = load_points<T>(ctx, (const T*)ctx->raw_in.p, raw, n, dim))) return rc;
    ctx->box_active = false;
    ctx->topology_build = true;
    rc = build_hash<T>(ctx, raw, sorted, n, dim, 0, r > 0 ? r : 1e-300);
    ctx->topology_build = false;
    if (rc) return rc;
    span_end(ctx, sp);
    SearchArgs<T> a{};
    a.grid = (const Grid<T>*)ctx->grid.p;
    a.snap = sorted;
    a.query = sorted;
    a.cell_start = (const int32_t*)ctx->cell_start.p;
    a.n = (int32_t)n;
    if ((rc = ensure(ctx, ctx->fb_list, sizeof(int32_t) * (size_t)n))) return rc;
    if ((rc = ensure(ctx, ctx->fb_count, 64))) return rc;
    ctx->counters_clean = false; // (this call counts in the block; the next sweep clears it itself)
    a.fb_list = (int32_t*)ctx->fb_list.p; // queries the brick kernel hands back to the wave kernel
    a.fb_count = (int32_t*)ctx->fb_count.p;
    // fp32: the brick kernel parks the rows it finds (32 ids per query) and marks the query, so that wtp_radius_fill copies
    // them instead of running the whole search a second time (128 B per point of scratch; WTP_RADIUS_CACHE=0 switches it off)
    ctx->rad_rows_cached = false;
    if (ctx->force_generic != 2 && !(getenv("WTP_RADIUS_CACHE") && atoi(getenv("WTP_RADIUS_CACHE")) == 0)) {
        if ((rc = ensure(ctx, ctx->rad_done, (size_t)n + 64))) return rc;
        WTP_HIP(ctx, hipMemsetAsync(ctx->rad_done.p, 0, (size_t)n, ctx->stream));
        a.rad_done = (uint8_t*)ctx->rad_done.p;
        if (sizeof(T) == 4 && !ctx->force_generic) { // the brick kernel's rows: 32 ids per query
            if ((rc = ensure(ctx, ctx->rad_tmp, sizeof(int32_t) * 32 * (size_t)n))) return rc;
            a.rad_tmp = (int32_t*)ctx->rad_tmp.p;
        }
        // the wave kernel's rows (any length up to its list), where it serves every query (fp64; fp32 grids whose rows are
        // expected to outgrow the brick kernel, Grid::rad_wave_only; WTP_FORCE_GENERIC=1): an arena of 48 ids per point,
        // shared out evenly among the waves; a row that does not fit any more is simply searched again by the fill phase.
        // (For the hand-backs of the fp32 brick kernel it buys nothing: measured 2.07 -> 2.14 ms per graded 1 M cloud —
        // ranking in the count phase costs what it saves in the fill phase; the kernel decides by the grid's flag.)
        {
            const int64_t arena_cap = 48 * n;
            if ((rc = ensure(ctx, ctx->rad_arena, sizeof(int32_t) * (size_t)arena_cap))) return rc;
            if ((rc = ensure(ctx, ctx->rad_arena_off, sizeof(int64_t) * (size_t)(n + 2)))) return rc;
            a.rad_arena = (int32_t*)ctx->rad_arena.p;
            a.rad_arena_off = (int64_t*)ctx->rad_arena_off.p;
            if ((rc = ensure(ctx, ctx->rad_pos, 64))) return rc;
            WTP_HIP(ctx, hipMemsetAsync(ctx->rad_pos.p, 0, 16, ctx->stream));
            if ((rc = ensure(ctx, ctx->rad_bricks, sizeof(int32_t) * (size_t)(n + 64)))) return rc;
            a.rad_bricks = (int32_t*)ctx->rad_bricks.p;
            a.rad_arena_pos = (unsigned long long*)ctx->rad_pos.p; // the dense kernel takes pieces of the arena (wtp_radb.hip)
            a.rad_arena_cap = arena_cap;
        }
        ctx->rad_rows_cached = true;
    }
    sp = span_begin(ctx, 1);
    rc = launch_radius_count<T>(ctx, a, (T)r, d_counts);
    ctx->rad_dense_used = a.rad_dense > 0;
    span_end(ctx, sp);
    if (!rc && a.rad_dense > 0 && getenv("WTP_DEBUG")) {
        int32_t h[4] = {0, 0, 0, 0};
        WTP_HIP(ctx, hipMemcpyAsync(h, ctx->rad_pos.p, 16, hipMemcpyDeviceToHost, ctx->stream));
        WTP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        fprintf(stderr, "[wtp] radius, dense kernel: %d bricks, %d of %lld queries, %lld ids parked\n", h[2], h[3], (long long)n,
                (long long)(((unsigned long long)(uint32_t)h[1] << 32) | (uint32_t)h[0]));
    }
    return rc;
}

template <typename T> static int radius_fill_t(wtp_ctx* ctx, const int64_t* d_off, int32_t* d_idx) {
    SearchArgs<T> a{};
    a.grid = (const Grid<T>*)ctx->grid.p;
    a.snap = (const Pt<T>*)ctx->pts[1].p;
    a.query = a.snap;
    a.cell_start = (const int32_t*)ctx->cell_start.p;
    a.n = (int32_t)ctx->rad_n;
    a.fb2_list = (int32_t*)ctx->fb2_list.p;
    a.fb2_count = (int32_t*)ctx->fb2_count.p;
    a.fb_list = (int32_t*)ctx->fb_list.p; // ensured by the count phase
    a.fb_count = (int32_t*)ctx->fb_count.p;
    if (ctx->rad_rows_cached) { // (set by the count phase of this very cloud: rad_valid guards the pair of calls)
        a.rad_tmp = sizeof(T) == 4 && !ctx->force_generic ? (int32_t*)ctx->rad_tmp.p : nullptr;
        a.rad_done = (uint8_t*)ctx->rad_done.p;
        {
            a.rad_arena = (int32_t*)ctx->rad_arena.p;
            a.rad_arena_off = (int64_t*)ctx->rad_arena_off.p;
            a.rad_arena_cap = 48 * ctx->rad_n;
        }
        a.rad_dense = ctx->rad_dense_used ? 1 : 0; // (the hand-back list of the count phase is what is left to search)
    }
    int sp = span_begin(ctx, 1);
    int rc = launch_radius_fill<T>(ctx, a, (T)ctx->rad_r, d_off, d_idx);
    span_end(ctx, sp);
    return rc;
}

WTP_API int wtp_radius_count(wtp_ctx* ctx, const void* xyz, int64_t n, int dim, int dtype, double r, int32_t* counts_out) {
    int rc = check_cloud(ctx, xyz, n, dim, dtype);
    if (rc) return rc;
    if (!(r >= 0) || !std::isfinite(r)) return fail(ctx, WTP_ERR_ARG, "radius must be finite and >= 0");
    if (!counts_out) return fail(ctx, WTP_ERR_ARG, "counts_out is NULL");
    if ((rc = check_idle(ctx))) return rc;
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t ts = tsize(dtype);
    ctx->rad_valid = false;
    ctx->rad_offsets_dev = false;
    ctx->relax.have_tree = false;
    if ((rc = ensure(ctx, ctx->raw_in, ts * (size_t)n * dim))) return rc;
    if ((rc = ensure(ctx, ctx->counts_out, sizeof(int32_t) * (size_t)n))) return rc;
    WTP_HIP(ctx, hipMemcpyAsync(ctx->raw_in.p, xyz, ts * (size_t)n * dim, hipMemcpyHostToDevice, ctx->stream));
    rc = dtype == WTP_F32 ? radius_count_t<float>(ctx, n, dim, r, (int32_t*)ctx->counts_out.p)
                          : radius_count_t<double>(ctx, n, dim, r, (int32_t*)ctx->counts_out.p);
    if (rc) return rc;
    WTP_HIP(ctx, hipMemcpyAsync(counts_out, ctx->counts_out.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost,
                                ctx->stream));
    if ((rc = sync(ctx))) return rc;
    ctx->rad_n = n;
    ctx->rad_dim = dim;
    ctx->rad_dtype = dtype;
    ctx->rad_r = r;
    ctx->rad_valid = true;
    return WTP_OK;
}

// count + exclusive scan on the device: the caller gets the offsets it needs to allocate, the counts never
// cross the bus and the offsets stay resident for the fill
WTP_API int wtp_radius_offsets(wtp_ctx* ctx, const void* xyz, int64_t n, int dim, int dtype, double r,
                               int64_t* offsets_out) {
    int rc = check_cloud(ctx, xyz, n, dim, dtype);
    if (rc) return rc;
    if (!(r >= 0) || !std::isfinite(r)) return fail(ctx, WTP_ERR_ARG, "radius must be finite and >= 0");
    if (!offsets_out) return fail(ctx, WTP_ERR_ARG, "offsets_out is NULL");
    if ((rc = check_idle(ctx))) return rc;
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t ts = tsize(dtype);
    ctx->rad_valid = false;
    ctx->rad_offsets_dev = false;
    ctx->relax.have_tree = false;
    if ((rc = ensure(ctx, ctx->raw_in, ts * (size_t)n * dim))) return rc;
    if ((rc = ensure(ctx, ctx->counts_out, sizeof(int32_t) * (size_t)n))) return rc;
    if ((rc = ensure(ctx, ctx->dist_out, sizeof(int64_t) * (size_t)(n + 1)))) return rc; // offsets live here until the fill
    WTP_HIP(ctx, hipMemcpyAsync(ctx->raw_in.p, xyz, ts * (size_t)n * dim, hipMemcpyHostToDevice, ctx->stream));
    rc = dtype == WTP_F32 ? radius_count_t<float>(ctx, n, dim, r, (int32_t*)ctx->counts_out.p)
                          : radius_count_t<double>(ctx, n, dim, r, (int32_t*)ctx->counts_out.p);
    if (rc) return rc;
    // (radius_count_t uses scratch for nothing; the scan's tile sums go there)
    if ((rc = ensure(ctx, ctx->scratch, offsets_scan_tmp_bytes(n)))) return rc;
    int sp = span_begin(ctx, 2);
    rc = launch_offsets_scan(ctx, (const int32_t*)ctx->counts_out.p, n, (int64_t*)ctx->scratch.p, (int64_t*)ctx->dist_out.p);
    span_end(ctx, sp);
    if (rc) return rc;
    WTP_HIP(ctx, hipMemcpyAsync(offsets_out, ctx->dist_out.p, sizeof(int64_t) * (size_t)(n + 1), hipMemcpyDeviceToHost,
                                ctx->stream));
    if ((rc = sync(ctx))) return rc;
    ctx->rad_n = n;
    ctx->rad_dim = dim;
    ctx->rad_dtype = dtype;
    ctx->rad_r = r;
    ctx->rad_valid = true;
    ctx->rad_offsets_dev = true;
    ctx->rad_nnz = offsets_out[n];
    return WTP_OK;
}

WTP_API int wtp_radius_fill(wtp_ctx* ctx, const int64_t* offsets, int32_t* idx_out) {
    if (!ctx) return WTP_ERR_ARG;
    if (!ctx->rad_valid) return fail(ctx, WTP_ERR_STATE, "wtp_radius_fill needs a preceding wtp_radius_count");
    if (!offsets && !ctx->rad_offsets_dev)
        return fail(ctx, WTP_ERR_ARG, "offsets is NULL (only wtp_radius_offsets leaves them on the device)");
    const int64_t n = ctx->rad_n;
    if (offsets) {
        if (offsets[0] != 0) return fail(ctx, WTP_ERR_ARG, "offsets[0] must be 0");
        for (int64_t i = 0; i < n; ++i)
            if (offsets[i + 1] < offsets[i]) return fail(ctx, WTP_ERR_ARG, "offsets must be non-decreasing");
    }
    const int64_t nnz = offsets ? offsets[n] : ctx->rad_nnz;
    if (nnz > 0 && !idx_out) return fail(ctx, WTP_ERR_ARG, "idx_out is NULL");
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t ts = tsize(ctx->rad_dtype);
    int rc;
    if ((rc = ensure(ctx, ctx->idx_out, sizeof(int32_t) * (size_t)(nnz + 1)))) return rc;
    if ((rc = ensure(ctx, ctx->scratch, ts * (size_t)(nnz + 1)))) return rc;
    if ((rc = ensure(ctx, ctx->dist_out, sizeof(int64_t) * (size_t)(n + 1)))) return rc; // offsets staging
    if ((rc = ensure(ctx, ctx->fb2_list, sizeof(int32_t) * (size_t)n))) return rc;
    if ((rc = ensure(ctx, ctx->fb2_count, 64))) return rc;
    if (offsets)
        WTP_HIP(ctx, hipMemcpyAsync(ctx->dist_out.p, offsets, sizeof(int64_t) * (size_t)(n + 1), hipMemcpyHostToDevice,
                                    ctx->stream));
    rc = ctx->rad_dtype == WTP_F32 ? radius_fill_t<float>(ctx, (const int64_t*)ctx->dist_out.p, (int32_t*)ctx->idx_out.p)
                                   : radius_fill_t<double>(ctx, (const int64_t*)ctx->dist_out.p, (int32_t*)ctx->idx_out.p);
    if (rc) return rc;
    if (nnz > 0)
        WTP_HIP(ctx, hipMemcpyAsync(idx_out, ctx->idx_out.p, sizeof(int32_t) * (size_t)nnz, hipMemcpyDeviceToHost,
                                    ctx->stream));
    return sync(ctx);
}

// ---- variable spacings: kd-tree over the law's boundary points, cached per context -----------------
static bool spacing_on_device(int kind) { return kind == WTP_SPACING_LOGLIKE || kind == WTP_SPACING_BOUNDARY_LAYER; }

static int check_spacing_law(wtp_ctx* ctx, const wtp_spacing_desc* s) {
    if (!s->boundary_xyz || s->n_boundary < 1)
        return fail(ctx, WTP_ERR_ARG, "boundary_points must be non-empty"); // spacings.jl:61-62,106-107
    if (s->n_boundary > 2000000000LL) return fail(ctx, WTP_ERR_ARG, "n_boundary exceeds the int32 index space");
    if (s->kind == WTP_SPACING_BOUNDARY_LAYER && !(s->p2 > 0))
        return fail(ctx, WTP_ERR_ARG, "layer_thickness must be positive"); // spacings.jl:108-109
    return WTP_OK;
}

static int ensure_kd(wtp_ctx* ctx, const wtp_spacing_desc* s, int dim, int dtype) {
    const size_t ts = tsize(dtype);
    const size_t bytes = ts * (size_t)s->n_boundary * dim;
    uint64_t h = 1469598103934665603ull; // FNV-1a over the coordinates: same boundary -> same tree
    const unsigned char* b = (const unsigned char*)s->boundary_xyz;
    for (size_t i = 0; i < bytes; ++i) h = (h ^ b[i]) * 1099511628211ull;
    if (ctx->kd_m == s->n_boundary && ctx->kd_key == h && ctx->kd_dim == dim && ctx->kd_dtype == dtype) return WTP_OK;
    const size_t kdsz = dtype == WTP_F32 ? kd_bytes<float>(s->n_boundary) : kd_bytes<double>(s->n_boundary);
    int rc;
    if ((rc = ensure(ctx, ctx->kd_nodes, kdsz))) return rc;
    std::vector<char> host(kdsz);
    if (dtype == WTP_F32)
        kd_build_host<float>((const float*)s->boundary_xyz, s->n_boundary, dim, host.data());
    else
        kd_build_host<double>((const double*)s->boundary_xyz, s->n_boundary, dim, host.data());
    WTP_HIP(ctx, hipMemcpyAsync(ctx->kd_nodes.p, host.data(), host.size(), hipMemcpyHostToDevice, ctx->stream));
    if ((rc = sync(ctx))) return rc; // `host` dies with this scope
    ctx->kd_m = s->n_boundary;
    ctx->kd_key = h;
    ctx->kd_dim = dim;
    ctx->kd_dtype = dtype;
    return WTP_OK;
}

// upper bound of a law's values (sizes the compact-support grid; need not be attained)
static double spacing_law_max(const wtp_spacing_desc* s) {
    if (s->kind == WTP_SPACING_LOGLIKE) return s->p0;
    return s->p0 > s->p1 ? s->p0 : s->p1;
}

// ---- repel ------------------------------------------------------------------------------------------
static int flush_pending(wtp_ctx* ctx); // defined with wtp_relax_set_fixed_dev

static int pick_free(const RelaxState& r, int avoid_a, int avoid_b) {
    for (int i = 0; i < 3; ++i)
        if (i != avoid_a && i != avoid_b) return i;
    (void)r;
    return 0;
}

static int relax_init_impl(wtp_ctx* ctx, const void* snap_xyz, bool on_device, int64_t n, int64_t n_fixed, int dim,
                           int dtype, const wtp_spacing_desc* spacing, const wtp_force_desc* force, int k,
                           double alpha_lo, double alpha_max) {
    int rc = check_cloud(ctx, snap_xyz, n, dim, dtype);
    if (rc) return rc;
    if (n_fixed < 0 || n_fixed > n) return fail(ctx, WTP_ERR_ARG, "n_fixed must be in [0, n]");
    if (!spacing || !force) return fail(ctx, WTP_ERR_ARG, "spacing/force descriptor is NULL");
    if (k < 1) return fail(ctx, WTP_ERR_ARG, "k must be >= 1");
    if (force->kind < 0 || force->kind > 3) return fail(ctx, WTP_ERR_ARG, "unknown force kind");
    if (!(force->beta > 0)) return fail(ctx, WTP_ERR_ARG, "force beta must be > 0");
    if (spacing->kind == WTP_SPACING_CONSTANT) {
        if (!(spacing->constant > 0)) return fail(ctx, WTP_ERR_ARG, "constant spacing must be > 0");
    } else if (spacing->kind == WTP_SPACING_PER_POINT) {
        if (!spacing->per_point) return fail(ctx, WTP_ERR_ARG, "per_point spacing array is NULL");
    } else if (spacing_on_device(spacing->kind)) {
        if ((rc = check_spacing_law(ctx, spacing))) return rc;
    } else {
        return fail(ctx, WTP_ERR_ARG, "unknown spacing kind");
    }
    if (!(alpha_lo >= 0) || !(alpha_max >= alpha_lo)) return fail(ctx, WTP_ERR_ARG, "need 0 <= alpha_lo <= alpha_max");
    const int kk = (int64_t)k < n ? k : (int)n; // kk = min(k, length(snap)), src/repel.jl:208
    if (kk > kGenericKMax) return fail(ctx, WTP_ERR_ARG, "k > 128 is not supported");
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t ts = tsize(dtype);
    const size_t ptsz = dtype == WTP_F32 ? sizeof(float4) : sizeof(double4);
    RelaxState& r = ctx->relax;
    r = RelaxState{};
    ctx->rad_valid = false;
    for (int i = 0; i < 2; ++i)
        if ((rc = ensure(ctx, ctx->pts[i], ptsz * (size_t)n))) return rc;
    if ((rc = ensure(ctx, ctx->raw_in, ts * (size_t)n * dim))) return rc;
    if ((rc = ensure(ctx, ctx->forces, ts * (size_t)n))) return rc;
    if ((rc = ensure(ctx, ctx->nn_dist, ts * (size_t)n))) return rc;
    if ((rc = ensure(ctx, ctx->nn_id, sizeof(int32_t) * (size_t)n))) return rc;
    if ((rc = ensure(ctx, ctx->fb_list, sizeof(int32_t) * (size_t)n))) return rc;
    if ((rc = ensure(ctx, ctx->fb_count, 64))) return rc;
    ctx->counters_clean = false; // (this call counts in the block; the next sweep clears it itself)
    if ((rc = ensure(ctx, ctx->fb2_list, sizeof(int32_t) * (size_t)n))) return rc;
    if ((rc = ensure(ctx, ctx->fb2_count, 64))) return rc;
    const int n_partials = total_partials();
    if ((rc = ensure(ctx, ctx->partials, sizeof(Partial) * (size_t)n_partials))) return rc;
    WTP_HIP(ctx, hipMemcpyAsync(ctx->raw_in.p, snap_xyz, ts * (size_t)n * dim,
                                on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
    if (dtype == WTP_F32)
        rc = load_points<float>(ctx, (const float*)ctx->raw_in.p, (float4*)ctx->pts[0].p, n, dim);
    else
        rc = load_points<double>(ctx, (const double*)ctx->raw_in.p, (double4*)ctx->pts[0].p, n, dim);
    if (rc) return rc;
    if (spacing->kind == WTP_SPACING_PER_POINT) {
        if ((rc = ensure(ctx, ctx->spacing_pp, ts * (size_t)n))) return rc;
        WTP_HIP(ctx, hipMemcpyAsync(ctx->spacing_pp.p, spacing->per_point, ts * (size_t)n, hipMemcpyHostToDevice,
                                    ctx->stream));
    }
    if (spacing_on_device(spacing->kind)) {
        // spacings = spacing.(snap) (src/repel.jl:209): every snapshot point once, the wall included
        if ((rc = ensure_kd(ctx, spacing, dim, dtype))) return rc;
        if ((rc = ensure(ctx, ctx->spacing_pp, ts * (size_t)n))) return rc;
        if ((rc = ensure(ctx, ctx->sp_hint, sizeof(int32_t) * (size_t)n))) return rc;
        WTP_HIP(ctx, hipMemsetAsync(ctx->sp_hint.p, 0xFF, sizeof(int32_t) * (size_t)n, ctx->stream)); // -1: no hint
        if ((rc = ensure(ctx, ctx->sp_cert, 4 * ts * (size_t)n))) return rc;
        WTP_HIP(ctx, hipMemsetAsync(ctx->sp_cert.p, 0xFF, 4 * ts * (size_t)n, ctx->stream)); // no certificate yet
        if (dtype == WTP_F32)
            rc = launch_spacing_session<float>(ctx, (const float4*)ctx->pts[0].p, n, 0, ctx->kd_nodes.p, ctx->kd_m,
                                               spacing->kind, spacing->p0, spacing->p1, spacing->p2,
                                               (float*)ctx->spacing_pp.p, (int32_t*)ctx->sp_hint.p, nullptr, nullptr,
                                               ctx->sp_cert.p);
        else
            rc = launch_spacing_session<double>(ctx, (const double4*)ctx->pts[0].p, n, 0, ctx->kd_nodes.p, ctx->kd_m,
                                                spacing->kind, spacing->p0, spacing->p1, spacing->p2,
                                                (double*)ctx->spacing_pp.p, (int32_t*)ctx->sp_hint.p, nullptr, nullptr,
                                                ctx->sp_cert.p);
        if (rc) return rc;
    }
    if ((rc = sync(ctx))) return rc;
    r.active = true;
    r.n = n;
    r.n_fixed = n_fixed;
    r.dim = dim;
    r.dtype = dtype;
    r.k = kk;
    r.k_req = k;
    r.spacing_kind = spacing->kind;
    r.spacing_const = spacing->constant;
    r.spacing_max = spacing->constant;
    if (spacing->kind == WTP_SPACING_PER_POINT) r.spacing_max = host_max(spacing->per_point, n, dtype);
    if (spacing_on_device(spacing->kind)) {
        r.spacing_max = spacing_law_max(spacing);
        r.sp_p0 = spacing->p0;
        r.sp_p1 = spacing->p1;
        r.sp_p2 = spacing->p2;
    }
    r.alpha_lo = alpha_lo;
    r.alpha_max = alpha_max;
    r.force.kind = force->kind;
    r.force.beta = force->beta;
    r.force.u0 = force->u0;
    r.force.gamma = force->gamma;
    r.bufP = 0;
    r.bufS = -1;
    r.bufOld = -1;
    return WTP_OK;
}

WTP_API int wtp_relax_init(wtp_ctx* ctx, const void* snap_xyz, int64_t n, int64_t n_fixed, int dim, int dtype,
                           const wtp_spacing_desc* spacing, const wtp_force_desc* force, int k, double alpha_lo,
                           double alpha_max) {
    return relax_init_impl(ctx, snap_xyz, false, n, n_fixed, dim, dtype, spacing, force, k, alpha_lo, alpha_max);
}

WTP_API int wtp_relax_init_dev(wtp_ctx* ctx, const void* d_snap_xyz, int64_t n, int64_t n_fixed, int dim, int dtype,
                               const wtp_spacing_desc* spacing, const wtp_force_desc* force, int k,
                               double alpha_lo, double alpha_max) {
    return relax_init_impl(ctx, d_snap_xyz, true, n, n_fixed, dim, dtype, spacing, force, k, alpha_lo, alpha_max);
}

// Brick geometry of the round-2 compact-support sweep (wtp_cs2.hip), from a census of the grid just built:
// the brick length BX is set so that 97 % of the non-empty bricks hold at most ~244 queries (one round of
// the 256-thread workgroup) and the LDS point area so that 99.9 % of the halos fit; the rest takes a second
// round / the exact path.  Two or three tiny launches and read-backs, once per session.
static int cs2_tune(wtp_ctx* ctx, RelaxState& r, const Grid<float>& hg, double rho_eff) {
    int rc;
    if ((rc = ensure(ctx, ctx->occ, 513 * sizeof(unsigned int)))) return rc;
    if ((rc = ensure_pinned(ctx, 16384))) return rc;
    const double cells = (double)hg.n[0] * hg.n[1] * hg.n[2];
    double rho_est = cells > 0 ? (double)r.n / cells : 1.0;
    if (rho_eff - 1.0 > rho_est) rho_est = rho_eff - 1.0;
    if (rho_est < 0.25) rho_est = 0.25;
    const int bx_max = hg.n[0] < cs2_max_bx() ? (hg.n[0] < 1 ? 1 : hg.n[0]) : cs2_max_bx();
    auto clampbx = [&](double v) {
        int b = (int)(v + 0.5);
        return b < 2 ? (bx_max < 2 ? bx_max : 2) : (b > bx_max ? bx_max : b);
    };
    r.cs2_rho = rho_est;
    int bx = clampbx(220.0 / (4.0 * rho_est));
    int q97 = 0, h999 = 0;
    for (int it = 0; it < 4; ++it) {
        if ((rc = launch_cs2_census(ctx, bx, (unsigned int*)ctx->occ.p))) return rc;
        WTP_HIP(ctx, hipMemcpyAsync(ctx->host_pinned, ctx->occ.p, 513 * sizeof(unsigned int), hipMemcpyDeviceToHost,
                                    ctx->stream));
        if ((rc = sync(ctx))) return rc;
        const unsigned int* h = (const unsigned int*)ctx->host_pinned;
        const double nb = (double)h[512];
        auto quant = [&](int base, double frac, int width) {
            double run = 0;
            for (int b = 0; b < 256; ++b) {
                run += h[base + b];
                if (run >= frac * nb) return (b + 1) * width;
            }
            return 256 * width;
        };
        q97 = nb > 0 ? quant(0, 0.97, 2) : 0;
        h999 = nb > 0 ? quant(256, 0.999, 8) : 0;
        if (nb <= 0) break;
        // queries: aim at 236 for the 97th percentile; halo: at most ~1060 points (four workgroups per CU)
        double f = 1.0;
        if (q97 > 248 || q97 < 216) f = 236.0 / (double)q97;
        if (h999 * f > 1060.0) f = 1060.0 / (double)h999;
        const int nbx = clampbx(bx * f);
        if (nbx == bx || it == 3) break;
        bx = nbx;
    }
    // equal bricks along x: n[0] = 214 cells cut into bricks of 51 leaves a fifth brick of 10 cells that pays the
    // whole per-brick setup for a fifth of the work.  Cut the row into equal parts instead — as few as the two
    // limits (one round of the workgroup for 97 % of the bricks, the LDS point area) allow.
    if (hg.n[0] > bx && q97 > 0) {
        int best = 0;
        for (int parts = hg.n[0] / bx > 1 ? hg.n[0] / bx : 1; parts <= (hg.n[0] + bx - 1) / bx; ++parts) {
            const int bxc = (hg.n[0] + parts - 1) / parts;
            if (bxc > bx_max) continue;
            const double grow = (double)bxc / (double)bx;
            if (q97 * grow <= 254.0 && h999 * grow <= 1100.0) {
                best = bxc;
                h999 = (int)(h999 * grow) + 1;
                break; // the fewest parts that fit
            }
        }
        if (!best) {
            const int parts = (hg.n[0] + bx - 1) / bx;
            best = (hg.n[0] + parts - 1) / parts;
        }
        bx = best;
    }
    int hc = (int)(h999 * 1.05) + 48;
    hc = (hc + 63) / 64 * 64;
    r.brick_hcap = hc < 256 ? 256 : (hc > 1920 ? 1920 : hc);
    r.cs2_bx = bx;
    if (getenv("WTP_DEBUG"))
        fprintf(stderr, "[wtp] cs2 geometry: BX=%d hcap=%d (q97=%d h999=%d rho_est=%.3f grid %dx%dx%d c=%g)\n", bx,
                r.brick_hcap, q97, h999, rho_est, hg.n[0], hg.n[1], hg.n[2], (double)hg.c);
    return WTP_OK;
}

// Float64 sweep of a k-nearest law on a fresh snapshot (wtp_sweep64.hip): candidates from the fp32 k-selection kernels on a
// float copy of the snapshot with its own grid — the session's grid and cell table are parked meanwhile and come back
// untouched for the exact path —, exact re-ranking + force sum + step per query, the wave kernel for what is not certified.
static int relax_f64_ksel_sweep(wtp_ctx* ctx, SearchArgs<double>& a) {
    RelaxState& r = ctx->relax;
    const int64_t n = r.n;
    const int kc = 24;
    int rc;
    if ((rc = ensure(ctx, ctx->f32_pts, 2 * sizeof(float4) * (size_t)n))) return rc;
    if ((rc = ensure(ctx, ctx->cand_idx, sizeof(int32_t) * (size_t)n * kc))) return rc;
    if ((rc = ensure(ctx, ctx->cand_dist, sizeof(float) * (size_t)n * kc))) return rc;
    if ((rc = ensure(ctx, ctx->f64k_s64, sizeof(double4) * (size_t)n))) return rc;
    if ((rc = ensure(ctx, ctx->f64k_slot, sizeof(int32_t) * (size_t)n))) return rc;
    if ((rc = ensure(ctx, ctx->f64k_lists, sizeof(int32_t) * 2 * (size_t)n))) return rc;
    if ((rc = ensure(ctx, ctx->f64k_cnt, 64))) return rc;
    if ((rc = ensure(ctx, ctx->occ, 64))) return rc;
    const double4* snap = (const double4*)a.snap; // the sorted snapshot (fresh: the queries are its points)
    float4* raw32 = (float4*)ctx->f32_pts.p;
    float4* sorted32 = raw32 + n;
    double* org4 = (double*)ctx->occ.p + 4; // behind the occupancy counters
    int sp = span_begin(ctx, 0);
    if ((rc = launch_origin(ctx, snap, n, org4))) return rc;
    if ((rc = launch_f64k_local(ctx, snap, n, org4, raw32))) return rc;
    std::swap(ctx->grid, ctx->grid_b);
    std::swap(ctx->cell_start, ctx->cell_start_b);
    const void* ncells_session = ctx->ncells_dev;
    const bool box_session = ctx->box_active;
    ctx->box_active = false; // (a clipped box is in the session's coordinates)
    ctx->topology_build = true; // rows are ordered explicitly: no canonical-order pass
    const bool tuned = r.f64k_n > 0 && std::llabs((long long)(n - r.f64k_n)) * 20 <= (long long)n;
    if (tuned) {
        rc = build_hash<float>(ctx, raw32, sorted32, n, 3, kc, 0.0, r.f64k_rho, 0.0, r.f64k_scale);
    } else { // once per session: cell scale and occupancy measured on the float copy (host reads)
        double scale = 1.0, rho_eff = 0, rho_direct = ksel_rho_for(ctx, kc);
        Grid<float> hg;
        rc = build_hash_tuned<float>(ctx, raw32, sorted32, n, 3, kc, 0.0, rho_direct, 0.0, &scale, &rho_eff, &hg);
        if (!rc) {
            const double pick = ksel_pick_rho(ctx, (double)n, (double)hg.ncells, hg.n[0], rho_direct, rho_eff);
            if (std::fabs(pick - rho_direct) > 0.01 * rho_direct) {
                rho_direct = pick;
                scale = 1.0;
                rc = build_hash_tuned<float>(ctx, raw32, sorted32, n, 3, kc, 0.0, rho_direct, 0.0, &scale, &rho_eff, &hg);
            }
        }
        if (!rc) {
            r.f64k_n = n;
            r.f64k_scale = scale;
            r.f64k_rho = rho_direct;
            ksel_geometry(ctx, (double)n, (double)hg.ncells, hg.n[0], rho_eff, &r.f64k_bx, &r.f64k_hcap);
        }
    }
    ctx->topology_build = false;
    if (!rc) rc = launch_f64k_relabel(ctx, snap, sorted32, (int32_t*)ctx->f64k_slot.p, (double4*)ctx->f64k_s64.p, n);
    span_end(ctx, sp);
    sp = span_begin(ctx, 1);
    if (!rc) {
        SearchArgs<float> b{};
        b.grid = (const Grid<float>*)ctx->grid.p;
        b.snap = sorted32;
        b.query = sorted32;
        b.cell_start = (const int32_t*)ctx->cell_start.p;
        b.n = (int32_t)n;
        b.k = kc;
        b.include_self = 1;
        b.idx_out = (int32_t*)ctx->cand_idx.p;
        b.dist_out = (float*)ctx->cand_dist.p;
        b.fb_list = (int32_t*)ctx->f64k_lists.p;
        b.fb_count = (int32_t*)ctx->f64k_cnt.p;
        b.fb2_list = (int32_t*)ctx->f64k_lists.p + n;
        b.fb2_count = (int32_t*)ctx->f64k_cnt.p + 4;
        b.stop = ctx->stop_dev;
        b.diag = a.diag;
        b.ksel_bx = r.f64k_bx;
        b.brick_hcap = r.f64k_hcap;
        b.cap_count = (float)ksel_cap_count(ctx, kc);
        WTP_HIP(ctx, hipMemsetAsync(ctx->f64k_cnt.p, 0, 64, ctx->stream));
        b.counters_cleared = 1;
        rc = launch_topology<float>(ctx, b);
    }
    // the session's structures again (the float copy's stay where they are until the next sweep overwrites them)
    std::swap(ctx->grid, ctx->grid_b);
    std::swap(ctx->cell_start, ctx->cell_start_b);
    ctx->ncells_dev = ncells_session;
    ctx->box_active = box_session;
    if (rc) return rc;
    ctx->n_sweep_launches += 1;
    rc = launch_refine_sweep_f64(ctx, a, (const double4*)ctx->f64k_s64.p, (const int32_t*)ctx->f64k_slot.p,
                                 (const int32_t*)ctx->cand_idx.p, (const float*)ctx->cand_dist.p, org4);
    span_end(ctx, sp);
    if (rc) return rc;
    sp = span_begin(ctx, 2);
    rc = launch_generic_sweep<double>(ctx, a, false); // the exact path for what the certificate turned down
    span_end(ctx, sp);
    return rc;
}

template <typename T> static int relax_step_t(wtp_ctx* ctx, int rebuild, wtp_step_stats* d_slot) {
    RelaxState& r = ctx->relax;
    int rc;
    if (!r.have_tree || r.pending.active) rebuild = 1; // the reference builds its first tree in the setup (src/repel.jl:218)
    // Float64, a k-nearest law, 3-D: on a fresh snapshot the sweep takes its candidates from the fp32 k-selection kernels
    // (wtp_sweep64.hip).  Every sum on that route and on the exact path behind it is ordered by (d2, index) explicitly, so
    // the snapshot's cells need no canonical order.
    const bool f64k_ok = sizeof(T) == 8 && r.dim == 3 && ctx->ksel && ctx->f64_ksel && !ctx->force_generic &&
                         r.force.kind != WTP_FORCE_CLIPPED_SPACING && r.k >= 2 && r.k <= 22 && r.n >= 4096;
    // (ClippedSpacingForce keeps its compact-support kernels: measured through this route on the graded 10 M-point cloud,
    // 39 ms per iteration against 20 — the k-selection grid hands a quarter of a graded cloud's queries back)
    if (rebuild) {
        // snapshot tail <- p, tree rebuilt (src/repel.jl:245-253): scatter P into a free buffer
        const int t = pick_free(r, r.bufP, -1);
        if ((rc = ensure(ctx, ctx->pts[t], sizeof(Pt<T>) * (size_t)(r.n + r.shard_extra)))) return rc;
        ctx->hash_view = r.pending; // a replaced fixed head waiting in P (wtp_relax_set_fixed_dev)
        int sp = span_begin(ctx, 0);
        // Compact-support sweep (fp32, ClippedSpacingForce, k >= 2): cells only have to cover the
        // law's support u0*s and the nearest-neighbour radius, so they can be smaller than the k-NN
        // cells (rho ~ 3.5 instead of ~8): 2.3x fewer candidates per query.
        r.cs_sweep = r.force.kind == WTP_FORCE_CLIPPED_SPACING && r.k >= 2 && r.k < 32 && !ctx->full_select &&
                     !ctx->force_generic && !r.cs_disabled;
        // round-2 sweep (wtp_cs2.hip, fp32 3-D): the nearest neighbour comes from the support or from a
        // per-wave follow-up, so the cells only cover the support: rho ~ 1
        const bool cs2 = r.cs_sweep && ctx->cs2 && sizeof(T) == 4 && r.dim == 3;
        double rho_cs = r.cs_sweep ? (cs2 ? ctx->rho_cs2 : 3.5 * (ctx->rho / 9.0)) : 0.0;
        // every other law (and ClippedSpacingForce with WTP_FULL_SELECT=1 or over-full support cells): the sweep with the
        // explicit k-selection — on the x-slowest layout of wtp_ksel.hip where that applies (fp32, 3-D, k <= 22)
        const bool ksel_ok = sizeof(T) == 4 && r.dim == 3 && ctx->ksel && !ctx->force_generic && r.k >= 2 &&
                             r.k <= ksel_kmax() && r.n >= 4096;
        r.ksel_sweep = !r.cs_sweep && ksel_ok;
        if (r.ksel_sweep) rho_cs = r.grid_tuned && r.ksel_rho > 0 ? r.ksel_rho : ksel_rho_for(ctx, r.k);
        if (r.spacing_typ <= 0) { // once per session: the spacing a typical point asks for
            r.spacing_typ = r.spacing_const;
            if (r.spacing_kind != WTP_SPACING_CONSTANT) {
                if ((rc = ensure(ctx, ctx->occ, 64))) return rc;
                if ((rc = ensure_pinned(ctx, 1024))) return rc;
                if ((rc = launch_sum<T>(ctx, (const T*)ctx->spacing_pp.p, r.n, (double*)ctx->occ.p))) return rc;
                WTP_HIP(ctx, hipMemcpyAsync(ctx->host_pinned, ctx->occ.p, 16, hipMemcpyDeviceToHost, ctx->stream));
                if ((rc = sync(ctx))) return rc;
                const double mean = ((const double*)ctx->host_pinned)[0] / (double)r.n;
                const double var = ((const double*)ctx->host_pinned)[1] / (double)r.n - mean * mean;
                r.spacing_typ = mean + ctx->styp_sigma * std::sqrt(var > 0 ? var : 0.0);
                if (!(r.spacing_typ > 0)) r.spacing_typ = r.spacing_max;
            }
        }
        // The compact-support sweep needs cells that cover the law's support u0*s.  With a variable
        // spacing the cell edge follows the spacing a typical point asks for (the mean over points, which
        // the dense regions dominate), not the largest one: the few points whose support is wider than
        // that are handed to the exact path, instead of everybody's cells being 64x over-full.
        // (constant spacing: c - margin = c (1 - 1/256) must reach u0 s, 1.01 does; variable: 10 % headroom over the mean)
        const double cell_f = (cs2 && r.spacing_kind == WTP_SPACING_CONSTANT) ? 1.01 : 1.1;
        double min_cell = r.cs_sweep ? cell_f * r.force.u0 * (r.spacing_typ < r.spacing_max ? r.spacing_typ : r.spacing_max)
                                     : 0.0;
        if (!r.grid_tuned) { // once per session: measured cell edge, LDS point area sized from the real grid
            double rho_eff = 0;
            Grid<T> hg;
            rc = build_hash_tuned<T>(ctx, (const Pt<T>*)ctx->pts[r.bufP].p, (Pt<T>*)ctx->pts[t].p, r.n, r.dim, r.k, 0.0,
                                     rho_cs, min_cell, &r.cell_scale, &rho_eff, &hg);
            if (rc) return rc;
            // A spacing far coarser than the cloud (the reference's own tests repel 46 786 face centres 0.22 apart
            // with a spacing of 3): cells that cover the law's support then hold hundreds of points, every support
            // ball holds more than k of them and each query would go to the exact path one by one.  Such a session
            // takes the k-selection sweep on cells sized for the k-th neighbour instead (same results).  Only when
            // the crowded points are themselves queries: a dense FIXED wall around a few movable points is served
            // well by the support cells (their balls hold few points), and badly by small cells (the movable
            // points' k-th neighbour is many cells away).
            if (r.cs_sweep && rho_eff > (cs2 ? 5.0 : 4.0 * rho_cs) && 2 * r.n_fixed < r.n) {
                r.cs_disabled = true;
                r.cs_sweep = false;
                r.ksel_sweep = ksel_ok;
                rho_cs = r.ksel_sweep ? ksel_rho_for(ctx, r.k) : 0.0;
                min_cell = 0.0;
                r.cell_scale = 1.0;
                rc = build_hash_tuned<T>(ctx, (const Pt<T>*)ctx->pts[r.bufP].p, (Pt<T>*)ctx->pts[t].p, r.n, r.dim, r.k, 0.0,
                                         rho_cs, min_cell, &r.cell_scale, &rho_eff, &hg);
                if (rc) return rc;
            }
            r.cs2_bx = 0;
            if (r.cs_sweep && cs2) {
                Grid<float> hgf;
                memcpy(&hgf, &hg, sizeof(hgf)); // T == float here
                if ((rc = cs2_tune(ctx, r, hgf, rho_eff))) return rc;
            } else if (r.cs_sweep) {
                int hc = (int)(HCELLS * rho_eff * 1.15) + 128;
                hc = (hc + 63) / 64 * 64;
                r.brick_hcap = hc < 640 ? 640 : (hc > 2560 ? 2560 : hc);
            }
            if (r.ksel_sweep) {
                const double pick = ksel_pick_rho(ctx, (double)r.n, (double)hg.ncells, hg.n[0], rho_cs, rho_eff);
                if (std::fabs(pick - rho_cs) > 0.01 * rho_cs) {
                    rho_cs = pick;
                    r.cell_scale = 1.0;
                    rc = build_hash_tuned<T>(ctx, (const Pt<T>*)ctx->pts[r.bufP].p, (Pt<T>*)ctx->pts[t].p, r.n, r.dim, r.k, 0.0,
                                             rho_cs, min_cell, &r.cell_scale, &rho_eff, &hg);
                    if (rc) return rc;
                }
                r.ksel_rho = rho_cs;
                ksel_geometry(ctx, (double)r.n, (double)hg.ncells, hg.n[0], rho_eff, &r.ksel_bx, &r.ksel_hcap);
            }
            r.grid_tuned = true;
            r.tuned_fixed = r.n_fixed;
            r.grid_fixed = r.n_fixed;
            r.grid_age = 0;
        } else {
            // The bounding box moves by at most a spacing per sweep: it is recomputed every few rebuilds only
            // (and always after a point was placed by hand, a fixed head was swapped, or with a clipped box).
            // A block session swaps its ghost head every iteration; the layer keeps its place and, nearly, its size, so the
            // box of the last full pass still fits (what sticks out is clamped into edge cells: exact, as for a moved point).
            const bool head_ok = !ctx->hash_view.active ||
                                 (r.shard_grid_reuse && r.grid_fixed > 0 &&
                                  std::llabs((long long)(r.n_fixed - r.grid_fixed)) * 10 <= (long long)r.grid_fixed + 640);
            const bool reuse = r.grid_age < ctx->grid_reuse_max && !r.moved_by_hand && head_ok && !ctx->box_active;
            ctx->reuse_grid = reuse;
            r.grid_age = reuse ? r.grid_age + 1 : 0;
            if (!reuse) r.grid_fixed = r.n_fixed;
            ctx->topology_build = f64k_ok; // (no canonical-order pass: 0.37 ms per 10 M Float64 points)
            rc = build_hash<T>(ctx, (const Pt<T>*)ctx->pts[r.bufP].p, (Pt<T>*)ctx->pts[t].p, r.n, r.dim, r.k, 0.0, rho_cs,
                               min_cell, r.cell_scale);
            ctx->topology_build = false;
            ctx->reuse_grid = false;
        }
        r.last_rho_cs = rho_cs;
        span_end(ctx, sp);
        ctx->hash_view.active = false;
        if (rc) return rc;
        r.pending.active = false;
        r.bufS = t;
        r.bufP = t;
        r.have_tree = true;
        r.sweeps_since_rebuild = 0;
        r.moved_by_hand = false;
    }
    if (spacing_on_device(r.spacing_kind)) {
        // s = spacing(x_i) at the point's current position (src/repel.jl:251 on rebuilds, :260 in every
        // sweep): the movable tail is re-evaluated before each sweep, the wall keeps its setup values
        int sps = span_begin(ctx, 2);
        // (the slot order of P is the sorted order of the last rebuild: the grid groups the points of a wave compactly)
        rc = launch_spacing_session<T>(ctx, (const Pt<T>*)ctx->pts[r.bufP].p, r.n, r.n_fixed, ctx->kd_nodes.p, ctx->kd_m,
                                       r.spacing_kind, r.sp_p0, r.sp_p1, r.sp_p2, (T*)ctx->spacing_pp.p,
                                       (int32_t*)ctx->sp_hint.p - r.aux_off, r.have_tree ? (const int32_t*)ctx->cell_start.p : nullptr,
                                       ctx->grid.p, getenv("WTP_SP_CERT_OFF") ? nullptr : (void*)((Pt<T>*)ctx->sp_cert.p - r.aux_off)); // (the switch: A/B of the certificates)
        span_end(ctx, sps);
        if (rc) return rc;
    }
    const bool fresh = (r.bufS == r.bufP);
    const int o = pick_free(r, r.bufS, r.bufP);
    if ((rc = ensure(ctx, ctx->pts[o], sizeof(Pt<T>) * (size_t)(r.n + r.shard_extra)))) return rc;
    SearchArgs<T> a{};
    a.grid = (const Grid<T>*)ctx->grid.p;
    a.snap = (const Pt<T>*)ctx->pts[r.bufS].p;
    a.query = (const Pt<T>*)ctx->pts[r.bufP].p;
    a.cell_start = (const int32_t*)ctx->cell_start.p;
    a.n = (int32_t)r.n;
    a.k = r.k;
    a.include_self = 1;
    a.out = (Pt<T>*)ctx->pts[o].p;
    a.forces = (T*)ctx->forces.p;
    a.nn_dist = (T*)ctx->nn_dist.p;
    a.nn_id = (int32_t*)ctx->nn_id.p;
    a.spacing_pp = r.spacing_kind != WTP_SPACING_CONSTANT ? (const T*)ctx->spacing_pp.p : nullptr;
    a.spacing_const = (T)r.spacing_const;
    a.alpha_lo = (T)r.alpha_lo;
    a.alpha_max = (T)r.alpha_max;
    a.beta = (T)r.force.beta;
    a.u0 = (T)r.force.u0;
    a.gamma = (T)r.force.gamma;
    a.force_kind = r.force.kind;
    a.n_fixed = (int32_t)r.n_fixed;
    a.partials = (Partial*)ctx->partials.p;
    a.n_partials = total_partials();
    a.fb_list = (int32_t*)ctx->fb_list.p;
    a.fb_count = (int32_t*)ctx->fb_count.p;
    a.fb2_list = (int32_t*)ctx->fb2_list.p;
    // one 64-byte counter block, cleared once per sweep: [0] hand-backs of the brick kernel,
    // [4] of the wave kernel, [8] uncovered queries
    a.fb2_count = (int32_t*)ctx->fb_count.p + 4;
    a.stop = ctx->stop_dev;
    a.nn_list = nullptr;
    a.nn_count = (int32_t*)ctx->fb_count.p + 2;
    // wtp_cs2.hip: cs_ball_kernel — variable spacing (supports wider than a cell), and every query of a sweep against a stale snapshot
    const bool ball = r.cs_sweep && sizeof(T) == 4 && (r.spacing_kind != WTP_SPACING_CONSTANT || r.bufS != r.bufP);
    if (r.cs_sweep && (r.cs2_bx > 0 || ball)) {
        if ((rc = ensure(ctx, ctx->nn_list, sizeof(int32_t) * (size_t)r.n))) return rc;
        a.nn_list = (int32_t*)ctx->nn_list.p;
    }
    if (ball) { // the follow-up kernel has consumed the list by the time the ball kernel refills it
        a.ball_list = (int32_t*)ctx->nn_list.p;
        a.ball_count = (int32_t*)ctx->fb_count.p + 6;
    }
    if (r.cs_sweep && sizeof(T) == 8 && (r.spacing_kind != WTP_SPACING_CONSTANT || r.bufS != r.bufP) && ctx->ball64) { // wtp_ball64.hip: its Float64 twin
        if ((rc = ensure(ctx, ctx->nn_list, sizeof(int32_t) * (size_t)r.n))) return rc;
        a.ball_list = (int32_t*)ctx->nn_list.p;
        a.ball_count = (int32_t*)ctx->fb_count.p + 6;
    }
    if ((rc = ensure(ctx, ctx->diag, 128))) return rc;
    a.diag = (unsigned long long*)ctx->diag.p;
    a.brick_hcap = r.cs_sweep ? r.brick_hcap : (r.ksel_sweep ? r.ksel_hcap : 0);
    a.cs2_bx = r.cs_sweep ? r.cs2_bx : 0;
    a.cs2_chunked = (r.spacing_kind != WTP_SPACING_CONSTANT || r.cs2_rho > 1.6) ? 1 : 0;
    a.ksel_bx = r.ksel_sweep ? r.ksel_bx : 0;
    if (r.ksel_sweep) a.cap_count = (float)ksel_cap_count(ctx, r.k);
    a.tnn_frac = (T)ctx->tnn_frac;
    a.cover_axis = r.cover_axis;
    a.cover_lo = (T)r.cover_lo;
    a.cover_hi = (T)r.cover_hi;
    for (int ax = 0; ax < 3; ++ax) {
        a.cover_lo3[ax] = (T)r.cover_lo3[ax];
        a.cover_hi3[ax] = (T)r.cover_hi3[ax];
    }
    a.uncovered = (int32_t*)ctx->fb_count.p + 8;
    if (!ctx->counters_clean) WTP_HIP(ctx, hipMemsetAsync(ctx->fb_count.p, 0, 64, ctx->stream));
    ctx->counters_clean = false; // (set again by the step's final reduction, which zeroes the block after reading it)
    a.used_brick = a.used_wave = a.used_generic = 0;
    bool by_candidates = false;
    if constexpr (sizeof(T) == 8) {
        // the k-nearest laws in Float64 on a fresh snapshot: fp32 candidates, exact re-ranking (wtp_sweep64.hip)
        by_candidates = fresh && f64k_ok;
        if (by_candidates && (rc = relax_f64_ksel_sweep(ctx, a))) return rc;
    }
    if (!by_candidates && (rc = launch_sweep<T>(ctx, a, fresh))) return rc;
    int sp = span_begin(ctx, 2);
    if (r.wall_active) { // p[id] = constrain(id, x_i, x_i + disp) (src/repel.jl:290): the octree wall rule
        char* wf = (char*)ctx->wall_flags.p;
        rc = launch_mesh_constrain<T>(ctx, a.query, a.out, r.n, r.n_fixed, r.wall_offset, (const uint8_t*)wf,
                                      (uint8_t*)wf + r.wall_nm, (int32_t*)ctx->wall_tri.p, (int32_t*)ctx->wall_hint.p,
                                      (int32_t*)ctx->fb_count.p + 12);
        if (rc) return rc;
    }
    rc = launch_reduce_partials(ctx, a.partials, a.n_partials, a.used_brick, a.used_wave, a.used_generic, a.fb_count,
                                a.uncovered, r.wall_active ? (const int32_t*)ctx->fb_count.p + 12 : nullptr, d_slot);
    span_end(ctx, sp);
    if (rc) return rc;
    r.bufOld = r.bufP; // p_old (src/repel.jl:244)
    r.bufP = o;
    r.can_revert = true;
    r.have_point_data = true;
    r.sweeps_since_rebuild += 1;
    return WTP_OK;
}

static int relax_step_any(wtp_ctx* ctx, int rebuild, wtp_step_stats* d_slot) {
    return ctx->relax.dtype == WTP_F32 ? relax_step_t<float>(ctx, rebuild, d_slot)
                                       : relax_step_t<double>(ctx, rebuild, d_slot);
}
int wtp::relax_step_enqueue(wtp_ctx* ctx, int rebuild, wtp_step_stats* d_slot) { return relax_step_any(ctx, rebuild, d_slot); }

// The block driver knows, before the ghost rows of an iteration have arrived, how many there will be.  When the rebuild that
// follows is going to keep its grid (the same rule as in relax_step_t), the snapshot's own entries are ranked into the
// cells right away, on the context's stream, while the rows travel on another; build_hash then ranks the appended head
// only.  A wrong guess costs one wasted pass, never a wrong result (build_hash checks what it finds).
int wtp::relax_prerank(wtp_ctx* ctx, int64_t n_fixed_new) {
    RelaxState& r = ctx->relax;
    ctx->prerank.valid = false;
    if (!r.active || !r.grid_tuned || !r.have_tree || r.pending.active || r.moved_by_hand || ctx->box_active) return WTP_OK;
    if (r.grid_age >= ctx->grid_reuse_max || !r.shard_grid_reuse || r.grid_fixed <= 0) return WTP_OK;
    if (std::llabs((long long)(n_fixed_new - r.grid_fixed)) * 10 > (long long)r.grid_fixed + 640) return WTP_OK;
    const int64_t n_new = r.n - r.n_fixed + n_fixed_new;
    if (r.cs2_bx > 0 && std::llabs((long long)(n_fixed_new - r.tuned_fixed)) * 20 > (long long)n_new) return WTP_OK;
    const size_t ptsz = r.dtype == WTP_F32 ? sizeof(float4) : sizeof(double4);
    if (ctx->pts[r.bufP].cap < ptsz * (size_t)(r.n + n_fixed_new)) return WTP_OK; // (the head would be rewritten, not appended)
    const int k = (int64_t)r.k_req < n_new ? r.k_req : (int)n_new;
    if (r.dtype == WTP_F32)
        return prerank_old_snapshot<float>(ctx, (const Pt<float>*)ctx->pts[r.bufP].p, r.n, (int32_t)r.n_fixed, n_new, r.n + n_fixed_new,
                                           k, r.last_rho_cs, r.cell_scale);
    return prerank_old_snapshot<double>(ctx, (const Pt<double>*)ctx->pts[r.bufP].p, r.n, (int32_t)r.n_fixed, n_new, r.n + n_fixed_new,
                                        k, r.last_rho_cs, r.cell_scale);
}

// The movable set of a session is replaced as a whole (block decomposition: points migrated in and out).  The caller
// writes the new points {x, y, z, bits(index)} into the buffer relax_swap_begin hands out and commits: the session then
// holds exactly these points, no fixed head, no tree — but keeps what it measured (cell scale, brick geometry, typical
// spacing), so the next rebuild costs one hash build, not a tuning pass.
int wtp::relax_swap_begin(wtp_ctx* ctx, int64_t n_move_new, void** d_buf_out) {
    RelaxState& r = ctx->relax;
    if (!r.active || r.pending.active) return fail(ctx, WTP_ERR_STATE, "relax_swap_begin: no session, or a pending fixed head");
    if (n_move_new < 1 || n_move_new > 2000000000LL) return fail(ctx, WTP_ERR_ARG, "relax_swap_begin: bad point count");
    const size_t ptsz = r.dtype == WTP_F32 ? sizeof(float4) : sizeof(double4);
    const int t = pick_free(r, r.bufP, -1);
    int rc;
    if ((rc = ensure(ctx, ctx->pts[t], ptsz * (size_t)(n_move_new + r.shard_extra)))) return rc;
    r.swap_target = t;
    *d_buf_out = ctx->pts[t].p;
    return WTP_OK;
}

int wtp::relax_swap_commit(wtp_ctx* ctx, int64_t n_move_new) {
    RelaxState& r = ctx->relax;
    if (!r.active || r.swap_target < 0) return fail(ctx, WTP_ERR_STATE, "relax_swap_commit without relax_swap_begin");
    const size_t ts = tsize(r.dtype);
    int rc;
    if (spacing_on_device(r.spacing_kind)) { // hints and certificates belonged to the old set: every point walks once
        if ((rc = ensure(ctx, ctx->spacing_pp, ts * (size_t)n_move_new))) return rc;
        if ((rc = ensure(ctx, ctx->sp_hint, sizeof(int32_t) * (size_t)n_move_new))) return rc;
        if ((rc = ensure(ctx, ctx->sp_cert, 4 * ts * (size_t)n_move_new))) return rc;
        WTP_HIP(ctx, hipMemsetAsync(ctx->sp_hint.p, 0xFF, sizeof(int32_t) * (size_t)n_move_new, ctx->stream));
        WTP_HIP(ctx, hipMemsetAsync(ctx->sp_cert.p, 0xFF, 4 * ts * (size_t)n_move_new, ctx->stream));
        r.aux_off = 0;
    }
    r.bufP = r.swap_target;
    r.swap_target = -1;
    r.n = n_move_new;
    r.n_fixed = 0;
    r.k = (int64_t)r.k_req < n_move_new ? r.k_req : (int)n_move_new;
    r.bufS = -1;
    r.bufOld = -1;
    r.have_tree = false;
    r.can_revert = false;
    r.have_point_data = false;
    r.moved_by_hand = true; // (the kept grid's bounding box is not this set's)
    return WTP_OK;
}

WTP_API int wtp_relax_step(wtp_ctx* ctx, int rebuild, wtp_step_stats* stats) {
    if (!ctx) return WTP_ERR_ARG;
    if (!ctx->relax.active) return fail(ctx, WTP_ERR_STATE, "wtp_relax_step before wtp_relax_init");
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = ensure(ctx, ctx->stats, sizeof(wtp_step_stats)))) return rc;
    if ((rc = relax_step_any(ctx, rebuild, (wtp_step_stats*)ctx->stats.p))) return rc;
    if (stats) {
        if ((rc = ensure_pinned(ctx, sizeof(wtp_step_stats)))) return rc;
        WTP_HIP(ctx, hipMemcpyAsync(ctx->host_pinned, ctx->stats.p, sizeof(wtp_step_stats), hipMemcpyDeviceToHost,
                                    ctx->stream));
        if ((rc = sync(ctx))) return rc;
        memcpy(stats, ctx->host_pinned, sizeof(wtp_step_stats));
        return WTP_OK;
    }
    return sync(ctx);
}

WTP_API int wtp_relax_run(wtp_ctx* ctx, int n_iters, int rebuild_every, double* conv_out, wtp_step_stats* last) {
    if (!ctx) return WTP_ERR_ARG;
    if (!ctx->relax.active) return fail(ctx, WTP_ERR_STATE, "wtp_relax_run before wtp_relax_init");
    if (rebuild_every < 1) return fail(ctx, WTP_ERR_ARG, "rebuild_every must be >= 1"); // src/repel.jl:74
    if (n_iters < 0) return fail(ctx, WTP_ERR_ARG, "n_iters must be >= 0");
    if (n_iters == 0) return WTP_OK;
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = ensure(ctx, ctx->stats, sizeof(wtp_step_stats) * (size_t)n_iters))) return rc;
    wtp_step_stats* d = (wtp_step_stats*)ctx->stats.p;
    for (int i = 0; i < n_iters; ++i)
        if ((rc = relax_step_any(ctx, (i % rebuild_every) == 0, d + i))) return rc;
    if (conv_out || last) {
        if ((rc = ensure_pinned(ctx, sizeof(wtp_step_stats) * (size_t)n_iters))) return rc;
        WTP_HIP(ctx, hipMemcpyAsync(ctx->host_pinned, d, sizeof(wtp_step_stats) * (size_t)n_iters,
                                    hipMemcpyDeviceToHost, ctx->stream));
        if ((rc = sync(ctx))) return rc;
        const wtp_step_stats* h = (const wtp_step_stats*)ctx->host_pinned;
        if (conv_out)
            for (int i = 0; i < n_iters; ++i) conv_out[i] = h[i].max_force;
        if (last) *last = h[n_iters - 1];
        return WTP_OK;
    }
    return sync(ctx);
}

// ---- stop rules on the device (src/repel.jl:305-334) -----------------------------------------------------------
// After every sweep of a batch one thread applies the reference's rules, in the reference's order, to that sweep's
// statistics: cv_target (the caller then reverts p to p_old), the stall counter on the CV of d_NN / s, the tolerance
// on max |F| s.  Once a rule fires, every kernel of the later iterations of the batch that would touch the session's
// state returns at once (SearchArgs::stop / ctx->stop_dev), so the state is that of the stopping iteration.
struct StopState {
    int32_t stopped, reason, n_done, last_impr;
    double best_cv;
};
__global__ void stop_rules_kernel(const wtp_step_stats* __restrict__ st, StopState* __restrict__ s, int iter1, double tol,
                                  int stall_after, double cv_target) {
    if (s->stopped) return;
    s->n_done = iter1;
    const double conv = st->max_force;
    if ((stall_after > 0 || cv_target > 0) && st->n_move > 0) {
        const double n = (double)st->n_move, mu = st->sum_u / n;
        const double var = st->sum_u2 / n - mu * mu;
        const double cv = sqrt(var > 0.0 ? var : 0.0) / mu; // _dnn_cv, src/repel.jl:374-386
        if (cv_target > 0 && cv <= cv_target) {
            s->stopped = 1;
            s->reason = 2;
            return;
        }
        if (stall_after > 0) {
            if (cv < s->best_cv * (1 - 1.0e-3)) {
                s->best_cv = cv;
                s->last_impr = iter1;
            } else if (iter1 - s->last_impr >= stall_after) {
                s->stopped = 1;
                s->reason = 3;
                return;
            }
        }
    }
    if (conv < tol) {
        s->stopped = 1;
        s->reason = 1;
    }
}

WTP_API int wtp_relax_run_until(wtp_ctx* ctx, int max_iters, int rebuild_every, double tol, int stall_after,
                                double cv_target, double* conv_out, int* n_done_out, int* reason_out,
                                wtp_step_stats* last) {
    if (!ctx) return WTP_ERR_ARG;
    RelaxState& r = ctx->relax;
    if (!r.active) return fail(ctx, WTP_ERR_STATE, "wtp_relax_run_until before wtp_relax_init");
    if (rebuild_every < 1) return fail(ctx, WTP_ERR_ARG, "rebuild_every must be >= 1"); // src/repel.jl:74
    if (max_iters < 0) return fail(ctx, WTP_ERR_ARG, "max_iters must be >= 0");
    if (r.wall_active) return fail(ctx, WTP_ERR_STATE, "wtp_relax_run_until: the octree wall rule steps through wtp_relax_step");
    if (n_done_out) *n_done_out = 0;
    if (reason_out) *reason_out = 0;
    if (max_iters == 0) return WTP_OK;
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = ensure(ctx, ctx->stats, sizeof(wtp_step_stats) * (size_t)max_iters))) return rc;
    if ((rc = ensure(ctx, ctx->stop_state, 64))) return rc;
    if ((rc = ensure_pinned(ctx, 64 + sizeof(wtp_step_stats) * (size_t)max_iters))) return rc;
    StopState h0{};
    h0.best_cv = __builtin_huge_val(); // typemax(U), src/repel.jl:238
    memcpy(ctx->host_pinned, &h0, sizeof(h0));
    WTP_HIP(ctx, hipMemcpyAsync(ctx->stop_state.p, ctx->host_pinned, sizeof(h0), hipMemcpyHostToDevice, ctx->stream));
    if ((rc = sync(ctx))) return rc; // (the pinned block is reused for the read-backs below)
    wtp_step_stats* d = (wtp_step_stats*)ctx->stats.p;
    StopState* ds = (StopState*)ctx->stop_state.p;
    const int kBatch = 16; // sweeps enqueued between two looks at the stop state
    std::vector<RelaxState> hist;
    hist.reserve((size_t)kBatch);
    int done = 0, reason = 0;
    ctx->stop_dev = &ds->stopped;
    for (int i0 = 0; i0 < max_iters && !reason; i0 += kBatch) {
        const int i1 = i0 + kBatch < max_iters ? i0 + kBatch : max_iters;
        hist.clear();
        for (int i = i0; i < i1; ++i) {
            if ((rc = relax_step_any(ctx, (i % rebuild_every) == 0, d + i))) {
                ctx->stop_dev = nullptr;
                return rc;
            }
            hist.push_back(r); // what the host believes after this sweep (buffer roles, grid age, ...)
            hipLaunchKernelGGL(stop_rules_kernel, dim3(1), dim3(1), 0, ctx->stream, d + i, ds, i + 1, tol, stall_after, cv_target);
        }
        WTP_HIP(ctx, hipMemcpyAsync(ctx->host_pinned, ds, sizeof(StopState), hipMemcpyDeviceToHost, ctx->stream));
        if ((rc = sync(ctx))) {
            ctx->stop_dev = nullptr;
            return rc;
        }
        StopState hs;
        memcpy(&hs, ctx->host_pinned, sizeof(hs));
        done = hs.n_done;
        if (hs.stopped) {
            reason = hs.reason;
            r = hist[(size_t)(done - 1 - i0)]; // the sweeps after the stop did nothing: forget that they were enqueued
        }
    }
    ctx->stop_dev = nullptr;
    if (reason == 2) { // cv_target: p .= p_old (src/repel.jl:314)
        if ((rc = wtp_relax_revert(ctx))) return rc;
    }
    if (conv_out || last) {
        WTP_HIP(ctx, hipMemcpyAsync(ctx->host_pinned, d, sizeof(wtp_step_stats) * (size_t)done, hipMemcpyDeviceToHost,
                                    ctx->stream));
        if ((rc = sync(ctx))) return rc;
        const wtp_step_stats* hst = (const wtp_step_stats*)ctx->host_pinned;
        if (conv_out)
            for (int i = 0; i < done; ++i) conv_out[i] = hst[i].max_force;
        if (last && done > 0) *last = hst[done - 1];
    }
    if (n_done_out) *n_done_out = done;
    if (reason_out) *reason_out = reason;
    return WTP_OK;
}

WTP_API int wtp_relax_get(wtp_ctx* ctx, void* xyz_out) {
    if (!ctx) return WTP_ERR_ARG;
    RelaxState& r = ctx->relax;
    if (!r.active) return fail(ctx, WTP_ERR_STATE, "wtp_relax_get before wtp_relax_init");
    if (!xyz_out) return fail(ctx, WTP_ERR_ARG, "xyz_out is NULL");
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    if (int rcf = flush_pending(ctx)) return rcf;
    const int64_t n_move = r.n - r.n_fixed;
    if (n_move == 0) return WTP_OK;
    const size_t bytes = tsize(r.dtype) * (size_t)n_move * r.dim;
    int rc;
    if ((rc = ensure(ctx, ctx->scratch, bytes))) return rc;
    if (r.dtype == WTP_F32)
        rc = launch_unpermute<float>(ctx, (const float4*)ctx->pts[r.bufP].p, r.n, r.n_fixed, r.dim, (float*)ctx->scratch.p);
    else
        rc = launch_unpermute<double>(ctx, (const double4*)ctx->pts[r.bufP].p, r.n, r.n_fixed, r.dim, (double*)ctx->scratch.p);
    if (rc) return rc;
    WTP_HIP(ctx, hipMemcpyAsync(xyz_out, ctx->scratch.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return sync(ctx);
}

WTP_API int wtp_relax_get_dev(wtp_ctx* ctx, void* d_xyz_out) {
    if (!ctx) return WTP_ERR_ARG;
    RelaxState& r = ctx->relax;
    if (!r.active) return fail(ctx, WTP_ERR_STATE, "wtp_relax_get_dev before wtp_relax_init");
    if (!d_xyz_out) return fail(ctx, WTP_ERR_ARG, "xyz_out is NULL");
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    if (int rcf = flush_pending(ctx)) return rcf;
    if (r.n - r.n_fixed == 0) return WTP_OK;
    int rc;
    if (r.dtype == WTP_F32)
        rc = launch_unpermute<float>(ctx, (const float4*)ctx->pts[r.bufP].p, r.n, r.n_fixed, r.dim, (float*)d_xyz_out);
    else
        rc = launch_unpermute<double>(ctx, (const double4*)ctx->pts[r.bufP].p, r.n, r.n_fixed, r.dim, (double*)d_xyz_out);
    if (rc) return rc;
    return sync(ctx);
}

WTP_API int wtp_relax_get_point_data(wtp_ctx* ctx, void* forces_out, void* nn_dist_out, int32_t* nn_id_out) {
    if (!ctx) return WTP_ERR_ARG;
    RelaxState& r = ctx->relax;
    if (!r.active || !r.have_point_data || r.bufOld < 0)
        return fail(ctx, WTP_ERR_STATE, "wtp_relax_get_point_data needs a completed wtp_relax_step");
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t n_move = r.n - r.n_fixed;
    if (n_move == 0) return WTP_OK;
    const size_t ts = tsize(r.dtype);
    int rc;
    if ((rc = ensure(ctx, ctx->scratch, (2 * ts + 4) * (size_t)n_move))) return rc;
    char* base = (char*)ctx->scratch.p;
    void* fo = base;
    void* no = base + ts * n_move;
    int32_t* io = (int32_t*)(base + 2 * ts * n_move);
    // per-point arrays are in the slot order of the sweep's query buffer (== bufOld)
    if (r.dtype == WTP_F32)
        rc = launch_unpermute_point_data<float>(ctx, (const float4*)ctx->pts[r.bufOld].p, r.n, r.n_fixed,
                                                (const float*)ctx->forces.p, (const float*)ctx->nn_dist.p,
                                                (const int32_t*)ctx->nn_id.p, (float*)fo, (float*)no, io);
    else
        rc = launch_unpermute_point_data<double>(ctx, (const double4*)ctx->pts[r.bufOld].p, r.n, r.n_fixed,
                                                 (const double*)ctx->forces.p, (const double*)ctx->nn_dist.p,
                                                 (const int32_t*)ctx->nn_id.p, (double*)fo, (double*)no, io);
    if (rc) return rc;
    if (forces_out) WTP_HIP(ctx, hipMemcpyAsync(forces_out, fo, ts * n_move, hipMemcpyDeviceToHost, ctx->stream));
    if (nn_dist_out) WTP_HIP(ctx, hipMemcpyAsync(nn_dist_out, no, ts * n_move, hipMemcpyDeviceToHost, ctx->stream));
    if (nn_id_out) WTP_HIP(ctx, hipMemcpyAsync(nn_id_out, io, 4 * (size_t)n_move, hipMemcpyDeviceToHost, ctx->stream));
    return sync(ctx);
}

WTP_API int wtp_relax_set(wtp_ctx* ctx, int64_t i, const void* xyz) {
    if (!ctx) return WTP_ERR_ARG;
    RelaxState& r = ctx->relax;
    if (!r.active) return fail(ctx, WTP_ERR_STATE, "wtp_relax_set before wtp_relax_init");
    if (!xyz) return fail(ctx, WTP_ERR_ARG, "xyz is NULL");
    if (i < 0 || i >= r.n - r.n_fixed) return fail(ctx, WTP_ERR_ARG, "movable point index out of range");
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    if (int rcf = flush_pending(ctx)) return rcf;
    const size_t ts = tsize(r.dtype);
    int rc;
    if ((rc = ensure(ctx, ctx->scratch, 64))) return rc;
    // P may alias the snapshot (fresh tree): the kick must not move the tree's copy, so give P
    // its own buffer first.
    if (r.bufP == r.bufS) {
        const int t = pick_free(r, r.bufS, r.can_revert ? r.bufOld : -1);
        if (t == r.bufS) return fail(ctx, WTP_ERR_STATE, "no free buffer for wtp_relax_set");
        const size_t ptsz = r.dtype == WTP_F32 ? sizeof(float4) : sizeof(double4);
        if ((rc = ensure(ctx, ctx->pts[t], ptsz * (size_t)r.n))) return rc;
        WTP_HIP(ctx, hipMemcpyAsync(ctx->pts[t].p, ctx->pts[r.bufP].p, ptsz * (size_t)r.n, hipMemcpyDeviceToDevice,
                                    ctx->stream));
        if (r.bufOld == t) r.can_revert = false;
        r.bufP = t;
    }
    WTP_HIP(ctx, hipMemcpyAsync(ctx->scratch.p, xyz, ts * r.dim, hipMemcpyHostToDevice, ctx->stream));
    const int32_t id = (int32_t)(i + r.n_fixed);
    r.moved_by_hand = true;
    if (r.dtype == WTP_F32)
        rc = launch_set_point<float>(ctx, (float4*)ctx->pts[r.bufP].p, r.n, id, r.dim, (const float*)ctx->scratch.p);
    else
        rc = launch_set_point<double>(ctx, (double4*)ctx->pts[r.bufP].p, r.n, id, r.dim, (const double*)ctx->scratch.p);
    if (rc) return rc;
    return sync(ctx);
}

// Many movable points placed at once (the deposition pass of the octree method lands a whole layer of
// escapees in one iteration): idx ascending, strictly increasing; one pass over the snapshot.
WTP_API int wtp_relax_set_batch(wtp_ctx* ctx, const int64_t* idx, const void* xyz, int64_t m) {
    if (!ctx) return WTP_ERR_ARG;
    RelaxState& r = ctx->relax;
    if (!r.active) return fail(ctx, WTP_ERR_STATE, "wtp_relax_set_batch before wtp_relax_init");
    if (m < 0) return fail(ctx, WTP_ERR_ARG, "m must be >= 0");
    if (m == 0) return WTP_OK;
    if (!idx || !xyz) return fail(ctx, WTP_ERR_ARG, "NULL array");
    std::vector<int32_t> ids((size_t)m);
    for (int64_t j = 0; j < m; ++j) {
        if (idx[j] < 0 || idx[j] >= r.n - r.n_fixed) return fail(ctx, WTP_ERR_ARG, "movable point index out of range");
        if (j > 0 && idx[j] <= idx[j - 1]) return fail(ctx, WTP_ERR_ARG, "indices must be strictly increasing");
        ids[(size_t)j] = (int32_t)(idx[j] + r.n_fixed);
    }
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    if (int rcf = flush_pending(ctx)) return rcf;
    const size_t ts = tsize(r.dtype);
    int rc;
    const size_t o_v = (sizeof(int32_t) * (size_t)m + 255) / 256 * 256;
    if ((rc = ensure(ctx, ctx->scratch, o_v + ts * (size_t)m * r.dim))) return rc;
    if (r.bufP == r.bufS) { // P may alias the snapshot (fresh tree): give P its own buffer first (as wtp_relax_set)
        const int t = pick_free(r, r.bufS, r.can_revert ? r.bufOld : -1);
        if (t == r.bufS) return fail(ctx, WTP_ERR_STATE, "no free buffer for wtp_relax_set_batch");
        const size_t ptsz = r.dtype == WTP_F32 ? sizeof(float4) : sizeof(double4);
        if ((rc = ensure(ctx, ctx->pts[t], ptsz * (size_t)r.n))) return rc;
        WTP_HIP(ctx, hipMemcpyAsync(ctx->pts[t].p, ctx->pts[r.bufP].p, ptsz * (size_t)r.n, hipMemcpyDeviceToDevice,
                                    ctx->stream));
        if (r.bufOld == t) r.can_revert = false;
        r.bufP = t;
    }
    char* b = (char*)ctx->scratch.p;
    WTP_HIP(ctx, hipMemcpyAsync(b, ids.data(), sizeof(int32_t) * (size_t)m, hipMemcpyHostToDevice, ctx->stream));
    WTP_HIP(ctx, hipMemcpyAsync(b + o_v, xyz, ts * (size_t)m * r.dim, hipMemcpyHostToDevice, ctx->stream));
    r.moved_by_hand = true;
    rc = r.dtype == WTP_F32 ? launch_set_points<float>(ctx, (float4*)ctx->pts[r.bufP].p, r.n, (const int32_t*)b, m, r.dim,
                                                       (const float*)(b + o_v))
                            : launch_set_points<double>(ctx, (double4*)ctx->pts[r.bufP].p, r.n, (const int32_t*)b, m, r.dim,
                                                        (const double*)(b + o_v));
    if (rc) return rc;
    return sync(ctx); // also keeps `ids` alive until the copy has run
}

WTP_API int wtp_relax_revert(wtp_ctx* ctx) {
    if (!ctx) return WTP_ERR_ARG;
    RelaxState& r = ctx->relax;
    if (!r.active || !r.can_revert || r.bufOld < 0)
        return fail(ctx, WTP_ERR_STATE, "wtp_relax_revert needs a wtp_relax_step to undo");
    r.bufP = r.bufOld; // p .= p_old (src/repel.jl:314)
    r.can_revert = false;
    return WTP_OK;
}

WTP_API int wtp_relax_set_spacing(wtp_ctx* ctx, const void* spacing) {
    if (!ctx) return WTP_ERR_ARG;
    RelaxState& r = ctx->relax;
    if (!r.active) return fail(ctx, WTP_ERR_STATE, "wtp_relax_set_spacing before wtp_relax_init");
    if (r.spacing_kind != WTP_SPACING_PER_POINT) return fail(ctx, WTP_ERR_STATE, "spacing is not PER_POINT");
    if (!spacing) return fail(ctx, WTP_ERR_ARG, "spacing is NULL");
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    WTP_HIP(ctx, hipMemcpyAsync(ctx->spacing_pp.p, spacing, tsize(r.dtype) * (size_t)r.n, hipMemcpyHostToDevice,
                                ctx->stream));
    r.spacing_max = host_max(spacing, r.n, r.dtype);
    return sync(ctx);
}

WTP_API int wtp_relax_get_spacing(wtp_ctx* ctx, void* spacing_out) {
    if (!ctx) return WTP_ERR_ARG;
    RelaxState& r = ctx->relax;
    if (!r.active) return fail(ctx, WTP_ERR_STATE, "wtp_relax_get_spacing before wtp_relax_init");
    if (!spacing_out) return fail(ctx, WTP_ERR_ARG, "spacing_out is NULL");
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t ts = tsize(r.dtype);
    if (r.spacing_kind == WTP_SPACING_CONSTANT) {
        for (int64_t i = 0; i < r.n; ++i) {
            if (r.dtype == WTP_F32) ((float*)spacing_out)[i] = (float)r.spacing_const;
            else ((double*)spacing_out)[i] = r.spacing_const;
        }
        return WTP_OK;
    }
    WTP_HIP(ctx, hipMemcpyAsync(spacing_out, ctx->spacing_pp.p, ts * (size_t)r.n, hipMemcpyDeviceToHost, ctx->stream));
    return sync(ctx);
}

template <typename T>
static int relax_query_knn_t(wtp_ctx* ctx, const void* xyz, int64_t nq, int k, int32_t* idx_out, void* dist_out) {
    RelaxState& r = ctx->relax;
    const size_t o_q = (sizeof(T) * (size_t)nq * r.dim + 255) / 256 * 256;
    int rc;
    if ((rc = ensure(ctx, ctx->scratch, o_q + sizeof(Pt<T>) * (size_t)nq))) return rc;
    if ((rc = ensure(ctx, ctx->idx_out, sizeof(int32_t) * (size_t)nq * k))) return rc;
    if (dist_out && (rc = ensure(ctx, ctx->dist_out, sizeof(T) * (size_t)nq * k))) return rc;
    WTP_HIP(ctx, hipMemcpyAsync(ctx->scratch.p, xyz, sizeof(T) * (size_t)nq * r.dim, hipMemcpyHostToDevice, ctx->stream));
    SearchArgs<T> a{};
    a.grid = (const Grid<T>*)ctx->grid.p;
    a.snap = (const Pt<T>*)ctx->pts[r.bufS].p;
    a.cell_start = (const int32_t*)ctx->cell_start.p;
    a.n = (int32_t)nq;
    a.k = k;
    a.idx_out = (int32_t*)ctx->idx_out.p;
    a.dist_out = dist_out ? (T*)ctx->dist_out.p : nullptr;
    int sp = span_begin(ctx, 2);
    rc = launch_query_knn<T>(ctx, a, (const T*)ctx->scratch.p, r.dim, (Pt<T>*)((char*)ctx->scratch.p + o_q));
    span_end(ctx, sp);
    if (rc) return rc;
    WTP_HIP(ctx, hipMemcpyAsync(idx_out, ctx->idx_out.p, sizeof(int32_t) * (size_t)nq * k, hipMemcpyDeviceToHost, ctx->stream));
    if (dist_out)
        WTP_HIP(ctx, hipMemcpyAsync(dist_out, ctx->dist_out.p, sizeof(T) * (size_t)nq * k, hipMemcpyDeviceToHost, ctx->stream));
    return sync(ctx);
}

WTP_API int wtp_relax_query_knn(wtp_ctx* ctx, const void* xyz, int64_t nq, int k, int32_t* idx_out, void* dist_out) {
    if (!ctx) return WTP_ERR_ARG;
    RelaxState& r = ctx->relax;
    if (!r.active || !r.have_tree || r.bufS < 0 || r.pending.active)
        return fail(ctx, WTP_ERR_STATE, "wtp_relax_query_knn needs a session that has swept at least once");
    if (nq < 0 || nq > 2000000000LL) return fail(ctx, WTP_ERR_ARG, "bad nq");
    if (k < 1 || k > r.n || k > kGenericKMax) return fail(ctx, WTP_ERR_ARG, "k must be in 1..min(n, 128)");
    if (nq == 0) return WTP_OK;
    if (!xyz || !idx_out) return fail(ctx, WTP_ERR_ARG, "NULL array");
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    return r.dtype == WTP_F32 ? relax_query_knn_t<float>(ctx, xyz, nq, k, idx_out, dist_out)
                              : relax_query_knn_t<double>(ctx, xyz, nq, k, idx_out, dist_out);
}

WTP_API int wtp_spacing_eval(wtp_ctx* ctx, const wtp_spacing_desc* spacing, const void* xyz, int64_t n, int dim,
                             int dtype, void* out) {
    int rc = check_cloud(ctx, xyz, n, dim, dtype);
    if (rc) return rc;
    if (!spacing || !out) return fail(ctx, WTP_ERR_ARG, "spacing/out is NULL");
    if (!spacing_on_device(spacing->kind))
        return fail(ctx, WTP_ERR_ARG, "wtp_spacing_eval evaluates LOGLIKE / BOUNDARY_LAYER descriptors");
    if ((rc = check_spacing_law(ctx, spacing))) return rc;
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t ts = tsize(dtype);
    if ((rc = ensure_kd(ctx, spacing, dim, dtype))) return rc;
    if ((rc = ensure(ctx, ctx->ins_in, ts * (size_t)n * dim))) return rc;
    if ((rc = ensure(ctx, ctx->ins_out, ts * (size_t)n))) return rc;
    WTP_HIP(ctx, hipMemcpyAsync(ctx->ins_in.p, xyz, ts * (size_t)n * dim, hipMemcpyHostToDevice, ctx->stream));
    if (dtype == WTP_F32)
        rc = launch_spacing_eval<float>(ctx, (const float*)ctx->ins_in.p, n, dim, ctx->kd_nodes.p, ctx->kd_m, spacing->kind,
                                        spacing->p0, spacing->p1, spacing->p2, (float*)ctx->ins_out.p);
    else
        rc = launch_spacing_eval<double>(ctx, (const double*)ctx->ins_in.p, n, dim, ctx->kd_nodes.p, ctx->kd_m,
                                         spacing->kind, spacing->p0, spacing->p1, spacing->p2, (double*)ctx->ins_out.p);
    if (rc) return rc;
    WTP_HIP(ctx, hipMemcpyAsync(out, ctx->ins_out.p, ts * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    return sync(ctx);
}

WTP_API int wtp_relax_end(wtp_ctx* ctx) {
    if (!ctx) return WTP_ERR_ARG;
    ctx->relax = RelaxState{};
    return WTP_OK;
}

// ---- isinside post-filter (src/repel.jl:90) ------------------------------------------------------------
static size_t align256(size_t b) { return (b + 255) / 256 * 256; }

WTP_API int wtp_isinside_greens(wtp_ctx* ctx, const void* test_xyz, int64_t n, const void* elem_xyz,
                                const void* elem_normal, const void* elem_area, int64_t m, int dtype,
                                uint8_t* inside_out, void* g_out) {
    if (!ctx) return WTP_ERR_ARG;
    if (dtype != WTP_F32 && dtype != WTP_F64) return fail(ctx, WTP_ERR_ARG, "dtype must be WTP_F32 or WTP_F64");
    if (n < 0 || m < 1) return fail(ctx, WTP_ERR_ARG, "need n >= 0 test points and m >= 1 boundary elements");
    if (n > 2000000000LL || m > 2000000000LL) return fail(ctx, WTP_ERR_ARG, "n or m exceeds the int32 index space");
    if (n == 0) return WTP_OK;
    if (!test_xyz || !elem_xyz || !elem_normal || !elem_area || !inside_out)
        return fail(ctx, WTP_ERR_ARG, "NULL array");
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t ts = tsize(dtype);
    const int chunks = isinside_chunks(ctx, n, m, isinside_greens_ppb());
    // ins_in: [test n*3 | p m*3 | normal m*3 | area m]; ins_out: [g n | inside n]
    const size_t o_p = align256(ts * n * 3), o_n = o_p + align256(ts * m * 3), o_a = o_n + align256(ts * m * 3);
    int rc;
    if ((rc = ensure(ctx, ctx->ins_in, o_a + ts * m))) return rc;
    if ((rc = ensure(ctx, ctx->ins_elems, isinside_elem_bytes(dtype) * (size_t)m))) return rc;
    if ((rc = ensure(ctx, ctx->ins_partial, ts * (size_t)n * chunks))) return rc;
    if ((rc = ensure(ctx, ctx->ins_out, align256(ts * n) + (size_t)n))) return rc;
    char* in = (char*)ctx->ins_in.p;
    char* out = (char*)ctx->ins_out.p;
    WTP_HIP(ctx, hipMemcpyAsync(in, test_xyz, ts * n * 3, hipMemcpyHostToDevice, ctx->stream));
    WTP_HIP(ctx, hipMemcpyAsync(in + o_p, elem_xyz, ts * m * 3, hipMemcpyHostToDevice, ctx->stream));
    WTP_HIP(ctx, hipMemcpyAsync(in + o_n, elem_normal, ts * m * 3, hipMemcpyHostToDevice, ctx->stream));
    WTP_HIP(ctx, hipMemcpyAsync(in + o_a, elem_area, ts * m, hipMemcpyHostToDevice, ctx->stream));
    uint8_t* d_inside = (uint8_t*)(out + align256(ts * n));
    int sp = span_begin(ctx, 2);
    if (dtype == WTP_F32)
        rc = launch_isinside_greens<float>(ctx, (const float*)in, n, (const float*)(in + o_p), (const float*)(in + o_n),
                                           (const float*)(in + o_a), m, ctx->ins_elems.p, chunks,
                                           (float*)ctx->ins_partial.p, (float*)out, d_inside);
    else
        rc = launch_isinside_greens<double>(ctx, (const double*)in, n, (const double*)(in + o_p),
                                            (const double*)(in + o_n), (const double*)(in + o_a), m, ctx->ins_elems.p,
                                            chunks, (double*)ctx->ins_partial.p, (double*)out, d_inside);
    span_end(ctx, sp);
    if (rc) return rc;
    WTP_HIP(ctx, hipMemcpyAsync(inside_out, d_inside, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    if (g_out) WTP_HIP(ctx, hipMemcpyAsync(g_out, out, ts * n, hipMemcpyDeviceToHost, ctx->stream));
    return sync(ctx);
}

WTP_API int wtp_isinside_winding(wtp_ctx* ctx, const void* test_xy, int64_t n, const void* poly_xy, int64_t m,
                                 int dtype, uint8_t* inside_out, void* sum_out) {
    if (!ctx) return WTP_ERR_ARG;
    if (dtype != WTP_F32 && dtype != WTP_F64) return fail(ctx, WTP_ERR_ARG, "dtype must be WTP_F32 or WTP_F64");
    if (m < 3) return fail(ctx, WTP_ERR_ARG, "need at least 3 points to define a polygon"); // src/isinside.jl:38-43
    if (n < 0 || n > 2000000000LL || m > 2000000000LL) return fail(ctx, WTP_ERR_ARG, "bad n or m");
    if (n == 0) return WTP_OK;
    if (!test_xy || !poly_xy || !inside_out) return fail(ctx, WTP_ERR_ARG, "NULL array");
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t ts = tsize(dtype);
    const int chunks = isinside_chunks(ctx, n, m, isinside_winding_ppb());
    const size_t o_p = align256(ts * n * 2);
    int rc;
    if ((rc = ensure(ctx, ctx->ins_in, o_p + ts * m * 2))) return rc;
    if ((rc = ensure(ctx, ctx->ins_partial, ts * (size_t)n * chunks))) return rc;
    if ((rc = ensure(ctx, ctx->ins_elems, sizeof(int32_t) * (size_t)n))) return rc; // coincidence flags
    if ((rc = ensure(ctx, ctx->ins_out, align256(ts * n) + (size_t)n))) return rc;
    char* in = (char*)ctx->ins_in.p;
    char* out = (char*)ctx->ins_out.p;
    WTP_HIP(ctx, hipMemcpyAsync(in, test_xy, ts * n * 2, hipMemcpyHostToDevice, ctx->stream));
    WTP_HIP(ctx, hipMemcpyAsync(in + o_p, poly_xy, ts * m * 2, hipMemcpyHostToDevice, ctx->stream));
    uint8_t* d_inside = (uint8_t*)(out + align256(ts * n));
    if (dtype == WTP_F32)
        rc = launch_isinside_winding<float>(ctx, (const float*)in, n, (const float*)(in + o_p), m, chunks,
                                            (float*)ctx->ins_partial.p, (int32_t*)ctx->ins_elems.p, (float*)out, d_inside);
    else
        rc = launch_isinside_winding<double>(ctx, (const double*)in, n, (const double*)(in + o_p), m, chunks,
                                             (double*)ctx->ins_partial.p, (int32_t*)ctx->ins_elems.p, (double*)out,
                                             d_inside);
    if (rc) return rc;
    WTP_HIP(ctx, hipMemcpyAsync(inside_out, d_inside, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    if (sum_out) WTP_HIP(ctx, hipMemcpyAsync(sum_out, out, ts * n, hipMemcpyDeviceToHost, ctx->stream));
    return sync(ctx);
}

// ---- sharded sessions ---------------------------------------------------------------------------------
WTP_API int wtp_set_stream(wtp_ctx* ctx, void* hip_stream, int external) {
    if (!ctx) return WTP_ERR_ARG;
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    spans_collect(ctx); // events of open spans belong to the old stream
    WTP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = external ? (hipStream_t)hip_stream : ctx->own_stream;
    return WTP_OK;
}

// launches the two layer kernels on the current P; *d_tot_out = device address of the four counts
static int enqueue_layers(wtp_ctx* ctx, int axis, double lo_in, double hi_in, double lo_out, double hi_out, void* d_lo4,
                          void* d_hi4, int64_t cap, int32_t** d_tot_out, int slot = 0) {
    RelaxState& r = ctx->relax;
    if (axis < 0 || axis >= r.dim) return fail(ctx, WTP_ERR_ARG, "axis must be in [0, dim)");
    if (cap < 0 || (cap > 0 && (!d_lo4 || !d_hi4))) return fail(ctx, WTP_ERR_ARG, "layer buffers are NULL");
    if (int rcf = flush_pending(ctx)) return rcf;
    int rc;
    const int nblk = layer_blocks(r.n);
    // (three regions: wtp_relax_step_layers3 keeps the layers of all three axes in flight)
    const size_t region = (64 + sizeof(int2) * (size_t)nblk + 63) / 64 * 64;
    if ((rc = ensure(ctx, ctx->scratch, 3 * region))) return rc;
    int32_t* d_tot = (int32_t*)((char*)ctx->scratch.p + (size_t)slot * region);
    int2* d_blk = (int2*)((char*)d_tot + 64);
    // P is in the slot order of the last rebuild's grid (and nobody moved farther than one spacing,
    // src/repel.jl:286-289) whenever a sweep produced it: then only the boundary cell layers are scanned
    const bool slot_ordered = r.have_tree && r.have_point_data && axis == r.dim - 1 && r.bufP != r.bufS &&
                              !r.moved_by_hand;
    const double reach = 1.001 * r.spacing_max * (double)(r.sweeps_since_rebuild > 0 ? r.sweeps_since_rebuild : 1);
    if (r.dtype == WTP_F32)
        rc = launch_layers<float>(ctx, (const float4*)ctx->pts[r.bufP].p, r.n, r.n_fixed, axis, lo_in, hi_in, lo_out,
                                  hi_out, (float4*)d_lo4, (float4*)d_hi4, cap, d_blk, d_tot, slot_ordered, reach);
    else
        rc = launch_layers<double>(ctx, (const double4*)ctx->pts[r.bufP].p, r.n, r.n_fixed, axis, lo_in, hi_in, lo_out,
                                   hi_out, (double4*)d_lo4, (double4*)d_hi4, cap, d_blk, d_tot, slot_ordered, reach);
    *d_tot_out = d_tot;
    return rc;
}

WTP_API int wtp_relax_layers_dev(wtp_ctx* ctx, int axis, double lo_in, double hi_in, double lo_out, double hi_out,
                                 void* d_lo4, void* d_hi4, int64_t cap, int64_t counts[4]) {
    if (!ctx) return WTP_ERR_ARG;
    RelaxState& r = ctx->relax;
    if (!r.active) return fail(ctx, WTP_ERR_STATE, "wtp_relax_layers_dev before wtp_relax_init");
    if (!counts) return fail(ctx, WTP_ERR_ARG, "counts is NULL");
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    int32_t* d_tot = nullptr;
    if ((rc = enqueue_layers(ctx, axis, lo_in, hi_in, lo_out, hi_out, d_lo4, d_hi4, cap, &d_tot))) return rc;
    if ((rc = ensure_pinned(ctx, 64))) return rc;
    WTP_HIP(ctx, hipMemcpyAsync(ctx->host_pinned, d_tot, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = sync(ctx))) return rc;
    for (int j = 0; j < 4; ++j) counts[j] = ((const int32_t*)ctx->host_pinned)[j];
    return WTP_OK;
}

// One sweep and, from the positions it produced, the boundary layers of the NEXT iteration, with a
// single read-back and a single synchronisation for both (a sharded iteration otherwise pays two).
WTP_API int wtp_relax_step_layers(wtp_ctx* ctx, int rebuild, wtp_step_stats* stats, int axis, double lo_in, double hi_in,
                                  double lo_out, double hi_out, void* d_lo4, void* d_hi4, int64_t cap,
                                  int64_t counts[4]) {
    if (!ctx) return WTP_ERR_ARG;
    if (!ctx->relax.active) return fail(ctx, WTP_ERR_STATE, "wtp_relax_step_layers before wtp_relax_init");
    if (!stats || !counts) return fail(ctx, WTP_ERR_ARG, "stats/counts is NULL");
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = ensure(ctx, ctx->stats, sizeof(wtp_step_stats)))) return rc;
    if ((rc = relax_step_any(ctx, rebuild, (wtp_step_stats*)ctx->stats.p))) return rc;
    int32_t* d_tot = nullptr;
    if ((rc = enqueue_layers(ctx, axis, lo_in, hi_in, lo_out, hi_out, d_lo4, d_hi4, cap, &d_tot))) return rc;
    const size_t off = (sizeof(wtp_step_stats) + 63) / 64 * 64;
    if ((rc = ensure_pinned(ctx, off + 64))) return rc;
    WTP_HIP(ctx, hipMemcpyAsync(ctx->host_pinned, ctx->stats.p, sizeof(wtp_step_stats), hipMemcpyDeviceToHost, ctx->stream));
    WTP_HIP(ctx, hipMemcpyAsync((char*)ctx->host_pinned + off, d_tot, 4 * sizeof(int32_t), hipMemcpyDeviceToHost,
                                ctx->stream));
    if ((rc = sync(ctx))) return rc;
    memcpy(stats, ctx->host_pinned, sizeof(wtp_step_stats));
    for (int j = 0; j < 4; ++j) counts[j] = ((const int32_t*)((const char*)ctx->host_pinned + off))[j];
    return WTP_OK;
}

// Materialise a pending input view (wtp_relax_set_fixed_dev below): stale fixed points dropped, the
// appended ones moved to the head, ids renumbered.  Every entry point that reads P calls this first;
// the usual consumer, the next rebuild, never needs it.
static int flush_pending(wtp_ctx* ctx) {
    RelaxState& r = ctx->relax;
    if (!r.pending.active) return WTP_OK;
    const size_t ptsz = r.dtype == WTP_F32 ? sizeof(float4) : sizeof(double4);
    int rc;
    const int t = pick_free(r, r.bufP, -1);
    if ((rc = ensure(ctx, ctx->pts[t], ptsz * (size_t)(r.n + r.shard_extra)))) return rc;
    if ((rc = ensure(ctx, ctx->scratch, 64))) return rc;
    const HashView v = r.pending;
    if (r.dtype == WTP_F32) {
        const float4* P = (const float4*)ctx->pts[r.bufP].p;
        rc = launch_refix<float>(ctx, P, v.n_old, v.fixed_old, r.n_fixed, P + v.n_old, (float4*)ctx->pts[t].p,
                                 (int32_t*)ctx->scratch.p);
    } else {
        const double4* P = (const double4*)ctx->pts[r.bufP].p;
        rc = launch_refix<double>(ctx, P, v.n_old, v.fixed_old, r.n_fixed, P + v.n_old, (double4*)ctx->pts[t].p,
                                  (int32_t*)ctx->scratch.p);
    }
    if (rc) return rc;
    r.bufP = t;
    r.pending.active = false;
    return WTP_OK;
}

// (keep_alive: the caller's array stays valid until the copy has run in stream order — the block driver's own pool)
int wtp::relax_set_fixed_dev_impl(wtp_ctx* ctx, const void* d_fixed4, int64_t n_fixed_new, bool keep_alive) {
    if (!ctx) return WTP_ERR_ARG;
    RelaxState& r = ctx->relax;
    if (!r.active) return fail(ctx, WTP_ERR_STATE, "wtp_relax_set_fixed_dev before wtp_relax_init");
    if (r.spacing_kind == WTP_SPACING_PER_POINT)
        return fail(ctx, WTP_ERR_STATE, "wtp_relax_set_fixed_dev: not with a caller-evaluated (PER_POINT) spacing array");
    if (n_fixed_new < 0 || (n_fixed_new > 0 && !d_fixed4)) return fail(ctx, WTP_ERR_ARG, "bad fixed-point array");
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = flush_pending(ctx))) return rc; // two calls in a row: the first one's view is materialised
    const int64_t n_move = r.n - r.n_fixed, n_new = n_move + n_fixed_new;
    if (n_new < 1) return fail(ctx, WTP_ERR_ARG, "the snapshot would be empty");
    if (n_new > 2000000000LL) return fail(ctx, WTP_ERR_ARG, "n exceeds the int32 index space");
    const size_t ts = tsize(r.dtype);
    const size_t ptsz = r.dtype == WTP_F32 ? sizeof(float4) : sizeof(double4);
    // from now on the point buffers keep room for a replaced head next to the old one
    const int64_t extra = n_fixed_new + n_fixed_new / 4 + 4096;
    if (extra > r.shard_extra) r.shard_extra = extra;
    if ((rc = ensure(ctx, ctx->forces, ts * (size_t)n_new))) return rc;
    if ((rc = ensure(ctx, ctx->nn_dist, ts * (size_t)n_new))) return rc;
    if ((rc = ensure(ctx, ctx->nn_id, sizeof(int32_t) * (size_t)n_new))) return rc;
    if ((rc = ensure(ctx, ctx->fb_list, sizeof(int32_t) * (size_t)n_new))) return rc;
    if ((rc = ensure(ctx, ctx->fb2_list, sizeof(int32_t) * (size_t)n_new))) return rc;
    if ((rc = ensure(ctx, ctx->scratch, 64))) return rc;
    if (spacing_on_device(r.spacing_kind)) {
        // The law is evaluated at the movable points before every sweep (src/repel.jl:260) and nobody reads a fixed
        // point's spacing, so the new head needs no values: the array only has to hold n_new entries.  Hints and
        // certificates stay where they are, addressed by movable index (aux_off follows the head's size).
        if ((rc = ensure(ctx, ctx->spacing_pp, ts * (size_t)n_new))) return rc; // (contents: rewritten before the next sweep)
        r.aux_off += n_fixed_new - r.n_fixed;
    }
    const bool fits = ctx->pts[r.bufP].cap >= ptsz * (size_t)(r.n + n_fixed_new);
    if (fits) {
        // No pass over the cloud: the new head is appended behind the old snapshot and the NEXT hash
        // build reads the array through a view that drops the stale fixed points and renumbers the
        // rest (HashView; 0.16 ms per iteration saved at 11 M points against rewriting the array).
        if (n_fixed_new > 0) {
            if (r.dtype == WTP_F32)
                rc = launch_append_fixed<float>(ctx, (const float4*)d_fixed4, n_fixed_new,
                                                (float4*)ctx->pts[r.bufP].p + r.n);
            else
                rc = launch_append_fixed<double>(ctx, (const double4*)d_fixed4, n_fixed_new,
                                                 (double4*)ctx->pts[r.bufP].p + r.n);
            if (rc) return rc;
        }
        r.pending.active = true;
        r.pending.n_old = r.n;
        r.pending.n_in = r.n + n_fixed_new;
        r.pending.fixed_old = (int32_t)r.n_fixed;
        r.pending.id_shift = (int32_t)(n_fixed_new - r.n_fixed);
    } else { // first call of a session (buffers sized for the plain snapshot): rewrite once
        const int t = pick_free(r, r.bufP, -1);
        if ((rc = ensure(ctx, ctx->pts[t], ptsz * (size_t)(n_new + r.shard_extra)))) return rc;
        if (r.dtype == WTP_F32)
            rc = launch_refix<float>(ctx, (const float4*)ctx->pts[r.bufP].p, r.n, r.n_fixed, n_fixed_new,
                                     (const float4*)d_fixed4, (float4*)ctx->pts[t].p, (int32_t*)ctx->scratch.p);
        else
            rc = launch_refix<double>(ctx, (const double4*)ctx->pts[r.bufP].p, r.n, r.n_fixed, n_fixed_new,
                                      (const double4*)d_fixed4, (double4*)ctx->pts[t].p, (int32_t*)ctx->scratch.p);
        if (rc) return rc;
        r.bufP = t;
    }
    // The brick geometry of the round-2 sweep (and which queries its bricks hand to the exact path, whose sums
    // round differently) was measured on the cloud of the first rebuild: a head that changes the cloud by more
    // than 5 % has it measured again, so that a resident session keeps equalling a fresh one bit for bit.
    if (r.cs2_bx > 0 && std::llabs((long long)(n_fixed_new - r.tuned_fixed)) * 20 > (long long)n_new) {
        r.grid_tuned = false;
        r.cell_scale = 1.0;
    }
    r.n = n_new;
    r.n_fixed = n_fixed_new;
    r.k = (int64_t)r.k_req < n_new ? r.k_req : (int)n_new;
    r.bufS = -1;
    r.bufOld = -1;
    r.have_tree = false;
    r.can_revert = false;
    r.have_point_data = false;
    // the caller's array must outlive the copy: on a lent stream that is stream order, else wait
    return (ctx->stream == ctx->own_stream && !keep_alive) ? sync(ctx) : WTP_OK;
}

WTP_API int wtp_relax_set_fixed_dev(wtp_ctx* ctx, const void* d_fixed4, int64_t n_fixed_new) {
    return relax_set_fixed_dev_impl(ctx, d_fixed4, n_fixed_new, false);
}

WTP_API int wtp_relax_set_coverage(wtp_ctx* ctx, int axis, double lo, double hi) {
    if (!ctx) return WTP_ERR_ARG;
    RelaxState& r = ctx->relax;
    if (!r.active) return fail(ctx, WTP_ERR_STATE, "wtp_relax_set_coverage before wtp_relax_init");
    if (axis >= r.dim) return fail(ctx, WTP_ERR_ARG, "axis must be < dim (negative: unlimited)");
    if (axis >= 0 && !(lo <= hi)) return fail(ctx, WTP_ERR_ARG, "need lo <= hi");
    r.cover_axis = axis < 0 ? -1 : axis;
    r.cover_lo = lo;
    r.cover_hi = hi;
    return WTP_OK;
}

// wtp_relax_step_layers for a block decomposition: the layers of up to three axes (bit a of axes_mask) come home
// with the statistics, one read-back and one synchronisation for everything.  counts[4*a .. 4*a+3] as in
// wtp_relax_layers_dev; d_lo4[a] / d_hi4[a] each hold `cap` rows.
WTP_API int wtp_relax_step_layers3(wtp_ctx* ctx, int rebuild, wtp_step_stats* stats, int axes_mask, const double lo_in[3],
                                   const double hi_in[3], const double lo_out[3], const double hi_out[3], void* const d_lo4[3],
                                   void* const d_hi4[3], int64_t cap, int64_t counts[12]) {
    if (!ctx) return WTP_ERR_ARG;
    if (!ctx->relax.active) return fail(ctx, WTP_ERR_STATE, "wtp_relax_step_layers3 before wtp_relax_init");
    if (!stats || !counts || !lo_in || !hi_in || !lo_out || !hi_out || !d_lo4 || !d_hi4)
        return fail(ctx, WTP_ERR_ARG, "NULL argument");
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = ensure(ctx, ctx->stats, sizeof(wtp_step_stats)))) return rc;
    if ((rc = relax_step_any(ctx, rebuild, (wtp_step_stats*)ctx->stats.p))) return rc;
    int32_t* d_tot[3] = {nullptr, nullptr, nullptr};
    for (int ax = 0; ax < 3; ++ax)
        if (axes_mask & (1 << ax))
            if ((rc = enqueue_layers(ctx, ax, lo_in[ax], hi_in[ax], lo_out[ax], hi_out[ax], d_lo4[ax], d_hi4[ax], cap,
                                     &d_tot[ax], ax)))
                return rc;
    const size_t off = (sizeof(wtp_step_stats) + 63) / 64 * 64;
    if ((rc = ensure_pinned(ctx, off + 3 * 64))) return rc;
    WTP_HIP(ctx, hipMemcpyAsync(ctx->host_pinned, ctx->stats.p, sizeof(wtp_step_stats), hipMemcpyDeviceToHost, ctx->stream));
    for (int ax = 0; ax < 3; ++ax)
        if (d_tot[ax])
            WTP_HIP(ctx, hipMemcpyAsync((char*)ctx->host_pinned + off + 64 * ax, d_tot[ax], 4 * sizeof(int32_t),
                                        hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = sync(ctx))) return rc;
    memcpy(stats, ctx->host_pinned, sizeof(wtp_step_stats));
    for (int ax = 0; ax < 3; ++ax)
        for (int j = 0; j < 4; ++j)
            counts[4 * ax + j] = d_tot[ax] ? ((const int32_t*)((const char*)ctx->host_pinned + off + 64 * ax))[j] : 0;
    return WTP_OK;
}

WTP_API int wtp_relax_set_coverage_box(wtp_ctx* ctx, const double lo[3], const double hi[3]) {
    if (!ctx || !lo || !hi) return WTP_ERR_ARG;
    RelaxState& r = ctx->relax;
    if (!r.active) return fail(ctx, WTP_ERR_STATE, "wtp_relax_set_coverage_box before wtp_relax_init");
    for (int ax = 0; ax < 3; ++ax) {
        if (!(lo[ax] <= hi[ax])) return fail(ctx, WTP_ERR_ARG, "need lo <= hi on every axis");
        r.cover_lo3[ax] = lo[ax];
        r.cover_hi3[ax] = hi[ax];
    }
    r.cover_axis = 3;
    return WTP_OK;
}

WTP_API int wtp_timers_get(wtp_ctx* ctx, double out[4]) {
    if (!ctx || !out) return WTP_ERR_ARG;
    hipSetDevice(ctx->device);
    spans_collect(ctx);
    out[0] = ctx->t_hash;
    out[1] = ctx->t_sweep;
    out[2] = ctx->t_other;
    out[3] = (double)ctx->n_sweep_launches;
    return WTP_OK;
}

WTP_API int wtp_timers_reset(wtp_ctx* ctx) {
    if (!ctx) return WTP_ERR_ARG;
    hipSetDevice(ctx->device);
    spans_collect(ctx);
    ctx->t_hash = ctx->t_sweep = ctx->t_other = 0;
    ctx->n_sweep_launches = 0;
    if (!ctx->timing_forced) ctx->timing = true; // the caller is going to read the timers
    return WTP_OK;
}

// Diagnostic builds (-DWTP_DIAG): per-phase wave-cycle sums of the brick kernel, accumulated
// over all launches since the last call; reading resets them.  Release builds leave zeros.
WTP_API int wtp_debug_diag(wtp_ctx* ctx, unsigned long long out[16]) {
    if (!ctx || !out) return WTP_ERR_ARG;
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = ensure(ctx, ctx->diag, 128))) return rc;
    WTP_HIP(ctx, hipMemcpyAsync(out, ctx->diag.p, 128, hipMemcpyDeviceToHost, ctx->stream));
    WTP_HIP(ctx, hipMemsetAsync(ctx->diag.p, 0, 128, ctx->stream));
    if ((rc = sync(ctx))) return rc;
    if (getenv("WTP_DEBUG_KD")) { // diagnostic builds: node visits of the spacing law's tree walk
        unsigned long long kd[2] = {0, 0};
        wtp::debug_kd_steps(kd);
        fprintf(stderr, "[wtp] kd walk: %llu node visits by %llu wave-walks (%.1f per walk)\n", kd[0], kd[1], kd[1] ? (double)kd[0] / (double)kd[1] : 0.0);
    }
    return WTP_OK;
}

// ---- device-side helpers for bench.py / the sharded driver (not part of the drop-in surface) -----
WTP_API int wtp_gen_uniform_dev(wtp_ctx* ctx, uint64_t seed, int64_t first, int64_t n, int dim, int dtype, void* d_out) {
    if (!ctx || !d_out || n < 0) return WTP_ERR_ARG;
    WTP_HIP(ctx, hipSetDevice(ctx->device));
    int rc = dtype == WTP_F32 ? launch_gen_uniform<float>(ctx, seed, first, n, dim, (float*)d_out)
                              : launch_gen_uniform<double>(ctx, seed, first, n, dim, (double*)d_out);
    if (rc) return rc;
    return sync(ctx);
}

