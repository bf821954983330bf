// Float64, ClippedSpacingForce, variable spacing: the queries whose support is wider than their cell.
//
// The compact-support brick kernel (wtp_brick64.hip) stages 27 cells per query; in the coarse part of a graded cloud the
// cells follow the spacing of the dense part and a support ball reaches two to four cells out, so those queries — a
// seventh of a 64x-graded cloud — were handed to the exact wave-per-query path (6 of 20 ms per 10 M-point iteration).
// fp32 has had the ball kernel of wtp_cs2.hip for them since round 2; it adds its terms in lane order, which Float64
// must not.  This is its Float64 twin: eight lanes per query walk the (2R+1)^3 block that provably holds the ball, the
// ball's points (at most k, self included: more than k is the k-selection's case) are collected in the group's LDS list,
// ranked by (d2, index), their force terms evaluated by the lanes and added by one lane in that order — what the wave
// kernel's compact-support shortcut does with the same points (wtp_wave.hip: cs_done), so the same bits.  A query alone
// in its ball needs only its nearest neighbour, which must lie inside the radius the block certifies.  Anything else
// (more than k points in the ball, a ball beyond four cells, a lonely query whose neighbour is not certified) is left
// for the wave kernel.
#include "wtp_device.hpp"
#include "wtp_internal.hpp"

namespace wtp {

constexpr int kB64Threads = 256;
constexpr int kB64Lanes = 8;                 // lanes per query
constexpr int kB64PerWave = 64 / kB64Lanes;  // queries a wave works on together
constexpr int kB64RMax = 4;
constexpr int kB64Cap = 32;                  // ball population the group's list holds (k <= 31)
constexpr int kB64BlocksMax = 3072;

struct B64Group {
    double d2[kB64Cap];
    double fx[kB64Cap], fy[kB64Cap], fz[kB64Cap];
    int32_t id[kB64Cap];
    int32_t slot[kB64Cap];
    int32_t order[kB64Cap]; // order[rank] = entry
    int32_t count;
    int32_t pad;
};

__global__ __launch_bounds__(kB64Threads) void cs_ball64_kernel(SearchArgs<double> a, const int32_t* __restrict__ list,
                                                                const int32_t* __restrict__ list_count,
                                                                int32_t* __restrict__ out_list, int32_t* __restrict__ out_count,
                                                                int part_base) {
    if (a.stop && *a.stop) return; // wtp_relax_run_until: a stop rule fired earlier in this batch
    __shared__ Acc sacc[kB64Threads / 64];
    __shared__ B64Group groups[kB64Threads / kB64Lanes];
    const Grid<double> g = *a.grid;
    const int n = *list_count;
    const int K = a.k;
    const int lane = threadIdx.x & 63;
    const int grp = lane / kB64Lanes, l8 = lane % kB64Lanes;
    B64Group* sg = &groups[threadIdx.x / kB64Lanes];
    const int wave_g = (blockIdx.x * kB64Threads + threadIdx.x) >> 6, nwaves = (gridDim.x * kB64Threads) >> 6;
    Acc acc = acc_empty();
    for (int i0 = wave_g * kB64PerWave; i0 < n; i0 += nwaves * kB64PerWave) {
        const int i = i0 + grp;
        const bool on = i < n;
        const int gslot = list[on ? i : i0];
        const double4 qp = a.query[gslot];
        const int32_t qid = w_to_id(qp.w);
        const double s = a.spacing_pp ? a.spacing_pp[qid] : a.spacing_const;
        const double lim = (a.u0 * a.u0) * (s * s); // the wave kernel's cs_lim
        const int cx = cell_coord(g, qp.x, 0), cy = cell_coord(g, qp.y, 1), cz = cell_coord(g, qp.z, 2);
        int R = 0; // smallest block that provably holds the ball
        double g2 = 0.0;
#pragma unroll
        for (int r = 1; r <= kB64RMax; ++r) {
            const double t = safe_radius2(g, qp.x, qp.y, qp.z, cx, cy, cz, r);
            const bool take = R == 0 && lim <= t;
            g2 = take ? t : g2;
            R = take ? r : R;
        }
        const bool ok = on && R > 0 && qid >= a.n_fixed;
        const int side = 2 * R + 1, nrows = ok ? side * side : 0;
        const int x0 = cx - R < 0 ? 0 : cx - R, x1 = cx + R > g.n[0] - 1 ? g.n[0] - 1 : cx + R;
        if (l8 == 0) sg->count = 0;
        __builtin_amdgcn_wave_barrier();
        // nearest point other than the query itself inside the certified radius (lexicographic in (d2, index))
        double nd2 = Lim<double>::inf();
        int32_t nid = 0x7FFFFFFF, nslot = 0;
        constexpr int kRowsPerLane = ((2 * kB64RMax + 1) * (2 * kB64RMax + 1) + kB64Lanes - 1) / kB64Lanes;
#pragma unroll 1
        for (int j = 0; j < kRowsPerLane; ++j) {
            const int row = l8 + kB64Lanes * j;
            const int y = cy + row % side - R, z = cz + row / side - R;
            const bool in = row < nrows && y >= 0 && y < g.n[1] && z >= 0 && z < g.n[2];
            if (!in) continue;
            const int base = (z * g.n[1] + y) * g.n[0];
            const int ps = a.cell_start[base + x0], pe = a.cell_start[base + x1 + 1];
            for (int p = ps; p < pe; p += 2) {
                double4 c[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) c[u] = a.snap[p + u < pe ? p + u : p];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (p + u >= pe) continue;
                    const double d = dist2<double>(qp.x, qp.y, qp.z, c[u].x, c[u].y, c[u].z);
                    const int32_t cid = w_to_id(c[u].w);
                    if (d <= lim) { // in the ball (self among them)
                        const int pos = atomicAdd(&sg->count, 1);
                        if (pos < kB64Cap) {
                            sg->d2[pos] = d;
                            sg->id[pos] = cid;
                            sg->slot[pos] = p + u;
                        }
                    }
                    if (cid != qid && d <= g2 && lex_lt(d, cid, nd2, nid)) {
                        nd2 = d;
                        nid = cid;
                        nslot = p + u;
                    }
                }
            }
        }
        // the group's nearest other point
#pragma unroll
        for (int dlt = kB64Lanes / 2; dlt >= 1; dlt >>= 1) {
            const double od = __shfl_xor(nd2, dlt, 64);
            const int32_t oi = __shfl_xor(nid, dlt, 64), os = __shfl_xor(nslot, dlt, 64);
            if (lex_lt(od, oi, nd2, nid)) {
                nd2 = od;
                nid = oi;
                nslot = os;
            }
        }
        __builtin_amdgcn_wave_barrier();
        const int m_lim = sg->count; // population of the ball, self included
        // what this kernel finishes: the ball holds 2 .. k points, or the query is alone in it and its nearest neighbour is certified
        const bool alone = m_lim == 1;
        const bool mine = ok && ((m_lim >= 2 && m_lim <= K && m_lim <= kB64Cap) || (alone && nid != 0x7FFFFFFF));
        if (!mine) {
            if (l8 == 0 && on) out_list[atomicAdd(out_count, 1)] = gslot;
            continue; // (group-uniform)
        }
        if (alone && l8 == 0) { // the list becomes {self, nearest}: the two nearest, as the wave kernel selects them (Kq = 2)
            sg->d2[1] = nd2;
            sg->id[1] = nid;
            sg->slot[1] = nslot;
        }
        __builtin_amdgcn_wave_barrier();
        const int m = alone ? 2 : m_lim;
        // canonical rank of the lane's entries, then their force terms
        for (int e = l8; e < m; e += kB64Lanes) {
            const double md = sg->d2[e];
            const int32_t mi = sg->id[e];
            int rank = 0;
            for (int j = 0; j < m; ++j) rank += lex_lt(sg->d2[j], sg->id[j], md, mi) ? 1 : 0;
            sg->order[rank] = e;
            double fx = 0, fy = 0, fz = 0;
            if (mi != qid) {
                const double4 c = a.snap[sg->slot[e]];
                add_force<double>(a, g.dim, s, qp.x, qp.y, qp.z, qid, c.x, c.y, c.z, mi, md, fx, fy, fz);
            }
            sg->fx[e] = fx;
            sg->fy[e] = fy;
            sg->fz[e] = fz;
        }
        __builtin_amdgcn_wave_barrier();
        if (l8 == 0) {
            double Fx = 0, Fy = 0, Fz = 0, nd = Lim<double>::inf();
            int32_t nn = -1;
            for (int r = 0; r < m; ++r) { // ascending (d2, id), self skipped by index (src/repel.jl:271)
                const int e = sg->order[r];
                if (sg->id[e] == qid) continue;
                if (nn < 0) {
                    nn = sg->id[e];
                    nd = wsqrt(sg->d2[e]);
                }
                Fx = Fx + sg->fx[e];
                Fy = Fy + sg->fy[e];
                Fz = Fz + sg->fz[e];
            }
            double4 o;
            const double f = step_point<double>(a, s, qp.x, qp.y, qp.z, Fx, Fy, Fz, o.x, o.y, o.z);
            o.w = qp.w;
            a.out[gslot] = o;
            a.forces[gslot] = f;
            a.nn_dist[gslot] = nd;
            a.nn_id[gslot] = nn;
            acc_point<double>(acc, f, nd, s, qid, nn);
            // sharded sessions: the answer rests on the support ball, or — alone in it — on the neighbour's distance too
            const double last = sg->d2[sg->order[m - 1]];
            const double need = alone ? (last > lim ? last : lim) : lim;
            if (reaches_past_cover<double>(a, qp.x, qp.y, qp.z, need)) atomicAdd(a.uncovered, 1);
        }
        __builtin_amdgcn_wave_barrier();
    }
    acc_block_reduce(acc, sacc);
    if (threadIdx.x == 0) acc_store(&a.partials[part_base + blockIdx.x], acc);
}

// Runs between the brick kernel and the exact path; on return a.fb_list / a.fb_count name what is left.
int launch_cs_ball64(wtp_ctx* ctx, SearchArgs<double>& a, int32_t* rest_list, int32_t* rest_count) {
    int blocks = (int)((a.n + 1023) / 1024);
    blocks = blocks < 8 ? 8 : (blocks > kB64BlocksMax ? kB64BlocksMax : blocks);
    hipLaunchKernelGGL(cs_ball64_kernel, dim3(blocks), dim3(kB64Threads), 0, ctx->stream, a, (const int32_t*)a.fb_list,
                       (const int32_t*)a.fb_count, rest_list, rest_count, a.used_brick);
    a.used_brick += blocks;
    a.fb_list = rest_list;
    a.fb_count = rest_count;
    WTP_HIP(ctx, hipGetLastError());
    return WTP_OK;
}

} // namespace wtp
