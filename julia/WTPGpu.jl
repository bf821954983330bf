# WTPGpu.jl — the Julia side of the drop-in: the bodies of WhatsThePoint.jl's neighbour/stencil functions
# replaced by ccalls into libwtp (include/wtp.h).  WRITTEN BLIND: the build image has no Julia toolchain,
# so this file has never been parsed or run; it only marshals (every bit of logic it relies on is tested
# from Python through the same C ABI).  INTEGRATION.md explains what replaces what.
#
#   using WhatsThePoint, WTPGpu
#   WTPGpu.build_knn_neighbors(points(cloud), 21)      # body of _build_knn_neighbors  (src/topology.jl:79-84)
#   WTPGpu.build_radius_neighbors(points(cloud), r)    # body of _build_radius_neighbors (src/topology.jl:91-97)
#   WTPGpu.relax!(p, p_old, snap, spacing, force_model; kwargs...)   # loop of _relax! (src/repel.jl:243-339)
module WTPGpu
using WhatsThePoint, Meshes, StaticArrays, Unitful
const lib = get(ENV, "WTP_LIB", "libwtp")          # csrc/libwtp.so on LD_LIBRARY_PATH / DL_LOAD_PATH

struct ForceDesc;  kind::Int32; beta::Float64; u0::Float64; gamma::Float64; end
struct SpacingDesc
    kind::Int32; constant::Float64; per_point::Ptr{Cvoid}
    p0::Float64; p1::Float64; p2::Float64; boundary_xyz::Ptr{Cvoid}; n_boundary::Int64
end
mutable struct StepStats
    max_force::Float64; sum_u::Float64; sum_u2::Float64; n_move::Int64
    argmin_i::Int64; argmin_j::Int64; argmin_r::Float64; n_fallback::Int64; n_uncovered::Int64
    n_escaped::Int64
    StepStats() = new()
end

const ctx = Ref{Ptr{Cvoid}}(C_NULL)
function context()
    if ctx[] == C_NULL
        dev = Cint[0]
        check(C_NULL, ccall((:wtp_create, lib), Cint, (Ptr{Cint}, Cint, Ptr{Ptr{Cvoid}}), dev, 1, ctx))
    end
    return ctx[]
end
function check(c, rc)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:wtp_last_error, lib), Cstring, (Ptr{Cvoid},), c))
    rc == 1 ? throw(ArgumentError(msg)) : error("libwtp status $rc: $msg")   # WTP_ERR_ARG -> ArgumentError (src/repel.jl:74)
end

dtype(::Type{Float32}) = Cint(0)
dtype(::Type{Float64}) = Cint(1)
# Vector{SVector{D,T}} is the AoS n x D layout the ABI wants (src/repel.jl:216)
raw(pts) = [WhatsThePoint._raw_point(p) for p in pts]

# ---- src/topology.jl:79-84 -------------------------------------------------------------------------
function build_knn_neighbors(points, k::Int)
    xs = raw(points); D = length(first(xs)); T = eltype(first(xs)); n = length(xs)
    idx = Matrix{Int32}(undef, k, n)                       # column i = row i of the C result
    check(context(), ccall((:wtp_knn, lib), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Cint, Cint, Cint, Cint, Ptr{Int32}, Ptr{Cvoid}),
        context(), xs, n, D, dtype(T), k, 0, idx, C_NULL))
    return [Int.(view(idx, :, i)) .+ 1 for i in 1:n]       # 0-based int32 -> 1-based Int
end

# search / searchdists (src/neighbors.jl:9-21): self first, distances ascending
function search_with_dists(points, k::Int)
    xs = raw(points); D = length(first(xs)); T = eltype(first(xs)); n = length(xs)
    idx = Matrix{Int32}(undef, k, n); dist = Matrix{T}(undef, k, n)
    check(context(), ccall((:wtp_knn, lib), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Cint, Cint, Cint, Cint, Ptr{Int32}, Ptr{Cvoid}),
        context(), xs, n, D, dtype(T), k, 1, idx, dist))
    return [Int.(view(idx, :, i)) .+ 1 for i in 1:n], [collect(view(dist, :, i)) for i in 1:n]
end

# ---- src/topology.jl:91-97 -------------------------------------------------------------------------
function build_radius_neighbors(points, radius)
    r = ustrip(WhatsThePoint._get_radius(radius, points))
    xs = raw(points); D = length(first(xs)); T = eltype(first(xs)); n = length(xs)
    offsets = Vector{Int64}(undef, n + 1)                  # scanned on the device
    check(context(), ccall((:wtp_radius_offsets, lib), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Cint, Cint, Float64, Ptr{Int64}), context(), xs, n, D, dtype(T), Float64(r), offsets))
    idx = Vector{Int32}(undef, max(offsets[end], 1))
    check(context(), ccall((:wtp_radius_fill, lib), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int32}), context(), C_NULL, idx))
    return [Int.(idx[(offsets[i] + 1):offsets[i + 1]]) .+ 1 for i in 1:n]
end

# ---- src/repel_forces.jl -> wtp_force_desc ----------------------------------------------------------
force_desc(f::InverseDistanceForce) = ForceDesc(0, Float64(f.β), 1.0, 3.0)
force_desc(f::SpacingEquilibriumForce) = ForceDesc(1, Float64(f.β), 1.0, 3.0)
force_desc(f::ClippedSpacingForce) = ForceDesc(2, Float64(f.β), Float64(f.u0), 3.0)
force_desc(f::StrongSpacingForce) = ForceDesc(3, Float64(f.β), 1.0, Float64(f.γ))

# ---- the session behind _relax! (src/repel.jl:207-339) ----------------------------------------------
function relax_init(snap, n_fixed::Int, spacings, force_model, k::Int, α_lo, α_max)
    xs = raw(snap); D = length(first(xs)); T = eltype(first(xs)); n = length(xs)
    sp = spacings isa Number ?
        SpacingDesc(0, Float64(spacings), C_NULL, 0.0, 0.0, 0.0, C_NULL, 0) :
        SpacingDesc(1, 0.0, pointer(spacings), 0.0, 0.0, 0.0, C_NULL, 0)      # Vector{T}, one value per snapshot point
    GC.@preserve spacings check(context(), ccall((:wtp_relax_init, lib), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Cint, Cint, Ref{SpacingDesc}, Ref{ForceDesc}, Cint, Float64, Float64),
        context(), xs, n, n_fixed, D, dtype(T), sp, force_desc(force_model), k, Float64(α_lo), Float64(α_max)))
    return (D, T, n - n_fixed)
end
function relax_step(rebuild::Bool)
    st = StepStats()
    check(context(), ccall((:wtp_relax_step, lib), Cint, (Ptr{Cvoid}, Cint, Ref{StepStats}), context(), rebuild, st))
    return st                                             # argmin_i / argmin_j are 0-based snapshot indices
end
relax_revert() = check(context(), ccall((:wtp_relax_revert, lib), Cint, (Ptr{Cvoid},), context()))
relax_end() = check(context(), ccall((:wtp_relax_end, lib), Cint, (Ptr{Cvoid},), context()))
function relax_set(i::Int, x)                              # i: 1-based movable index (the kick, src/repel.jl:431)
    v = collect(x)
    check(context(), ccall((:wtp_relax_set, lib), Cint, (Ptr{Cvoid}, Int64, Ptr{Cvoid}), context(), i - 1, v))
end
function relax_get(D, T, n_move)
    out = Vector{SVector{D, T}}(undef, n_move)
    check(context(), ccall((:wtp_relax_get, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), context(), out))
    return out
end

function relax_set_spacing(s::Vector)                     # per-point spacings of the whole snapshot, the cloud's eltype
    GC.@preserve s check(context(), ccall((:wtp_relax_set_spacing, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), context(), pointer(s)))
end
# the whole loop with the stop rules evaluated on the device (include/wtp.h: wtp_relax_run_until)
function relax_run_until(max_iters::Int, rebuild_every::Int, tol, stall_after::Int, cv_target)
    conv = zeros(Float64, max(max_iters, 1)); n_done = Ref{Cint}(0); reason = Ref{Cint}(0); st = StepStats()
    check(context(), ccall((:wtp_relax_run_until, lib), Cint,
        (Ptr{Cvoid}, Cint, Cint, Float64, Cint, Float64, Ptr{Float64}, Ref{Cint}, Ref{Cint}, Ref{StepStats}),
        context(), max_iters, rebuild_every, Float64(tol), stall_after, Float64(cv_target), conv, n_done, reason, st))
    return conv[1:n_done[]], Int(reason[])               # reason: 0 max_iters, 1 tol, 2 cv_target (reverted), 3 stall
end

# The loop of _relax! (src/repel.jl:243-339) with rebuild + sweep + reductions on the device.
#  * constant spacing, no kick, no trace: ONE ccall — the stop rules (src/repel.jl:305-334) run on the device;
#  * otherwise one ccall per iteration, the rules below exactly as the reference applies them, the spacing
#    callable re-evaluated before every sweep (s = spacing(xi), src/repel.jl:260; the array the CV monitor and the
#    kick read is the one of src/repel.jl:251), kick and trace as in src/repel.jl:294-304,396-433.
# (`constrain` of the octree method: wtp_relax_set_wall, see INTEGRATION.md.)
function relax!(p, p_old, snap, spacing, force_model; n_fixed, α_lo, α_max, k, max_iters, tol, rebuild_every,
                stall_after = 0, cv_target = 0.0, kick_after = 0, trace = nothing, n_protected = n_fixed)
    rebuild_every >= 1 || throw(ArgumentError("rebuild_every must be ≥ 1"))
    T = eltype(WhatsThePoint._raw_point(first(snap)))
    s = T.(ustrip.(spacing.(snap)))                        # the ABI reads n values of the CLOUD's float type
    constant = all(==(first(s)), s)
    D, _, n_move = relax_init(snap, n_fixed, constant ? first(s) : s, force_model, min(k, length(snap)),
                              ustrip(α_lo), ustrip(α_max))
    conv = T[]
    u = unit(Meshes.to(first(p))[1])
    try
        if constant && kick_after <= 0 && trace === nothing
            c, _ = relax_run_until(max_iters, rebuild_every, tol, stall_after, cv_target)
            append!(conv, T.(c))
        else
            best_cv = Inf; last_improvement = 0; i = 1
            kick_pair = (0, 0); kick_rs = Inf; kick_count = 0
            while i <= max_iters
                if !constant && i > 1                      # s = spacing(xi) at the current positions, every sweep
                    cur = relax_get(D, T, n_move)
                    for j in 1:n_move
                        s[n_fixed + j] = T(ustrip(spacing(Meshes.Point((cur[j] .* u)...))))
                    end
                    relax_set_spacing(s)
                end
                st = relax_step((i - 1) % rebuild_every == 0)
                push!(conv, T(st.max_force))
                if n_move > 0 && (trace !== nothing || kick_after > 0)   # _closest_pair, src/repel.jl:396-403
                    a, b = st.argmin_i + 1, st.argmin_j + 1             # 1-based snapshot indices
                    sp = b > 0 ? (s[a] + s[b]) / 2 : s[a]
                    pair = (r = st.argmin_r, s = sp, r_over_s = st.argmin_r / sp, idx_a = min(a, b), idx_b = max(a, b))
                    trace !== nothing && push!(trace, (iteration = i, pair...))
                    if kick_after > 0                                    # _maybe_kick!, src/repel.jl:415-433
                        frozen = (pair.idx_a, pair.idx_b) == kick_pair && abs(pair.r_over_s - kick_rs) < 1.0e-8
                        kick_count = frozen ? kick_count + 1 : 1
                        kick_pair = (pair.idx_a, pair.idx_b); kick_rs = pair.r_over_s
                        if kick_count >= kick_after
                            t = pair.idx_a > n_protected ? pair.idx_a : (pair.idx_b > n_protected ? pair.idx_b :
                                (pair.idx_a > n_fixed ? pair.idx_a : pair.idx_b))
                            if t > n_fixed
                                cur = relax_get(D, T, n_move)
                                d = randn(T, D); d ./= sqrt(sum(abs2, d))
                                relax_set(t - n_fixed, cur[t - n_fixed] .+ T(0.1) * s[t] .* d)
                            end
                            kick_count = 0
                        end
                    end
                end
                if (stall_after > 0 || cv_target > 0) && st.n_move > 0
                    μ = st.sum_u / st.n_move
                    cv = sqrt(max(st.sum_u2 / st.n_move - μ^2, 0)) / μ
                    if cv_target > 0 && cv <= cv_target
                        relax_revert(); break
                    end
                    if stall_after > 0
                        if cv < best_cv * (1 - 1.0e-3)
                            best_cv, last_improvement = cv, i
                        elseif i - last_improvement >= stall_after
                            break
                        end
                    end
                end
                conv[end] < tol && break
                i += 1
            end
        end
        raws = relax_get(D, T, n_move)
        for j in eachindex(p)
            p[j] = Meshes.Point((raws[j] .* u)...)
        end
    finally
        relax_end()
    end
    return conv
end

# ---- multi-GPU: the context's RCCL communicator (include/wtp.h: wtp_comm_*; one Julia process per GPU) ------------
# The caller's own transport carries the id once (MPI.Bcast!, a file); INTEGRATION.md lists the block iteration.
function comm_unique_id()
    id = Vector{UInt8}(undef, 128)
    check(context(), ccall((:wtp_comm_unique_id, lib), Cint, (Ptr{Cvoid}, Ptr{UInt8}), context(), id))
    return id
end
comm_init(id::Vector{UInt8}, rank::Int, nranks::Int) =
    check(context(), ccall((:wtp_comm_init, lib), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Cint, Cint), context(), id, rank, nranks))
comm_finalize() = check(context(), ccall((:wtp_comm_finalize, lib), Cint, (Ptr{Cvoid},), context()))
# device pointers (Ptr{Cvoid}) to packed 16-byte rows; returns the received counts, rows are stream-ordered
function comm_exchange_rows(peer_lo::Int, peer_hi::Int, send_lo, n_lo::Int, send_hi, n_hi::Int, recv_lo, recv_hi, cap::Int)
    got_lo = Ref{Int64}(0); got_hi = Ref{Int64}(0)
    check(context(), ccall((:wtp_comm_exchange_rows, lib), Cint,
        (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ref{Int64}, Ref{Int64}),
        context(), peer_lo, peer_hi, send_lo, n_lo, send_hi, n_hi, recv_lo, recv_hi, cap, got_lo, got_hi))
    return got_lo[], got_hi[]
end
comm_allreduce_stats!(st::StepStats) =
    check(context(), ccall((:wtp_comm_allreduce_stats, lib), Cint, (Ptr{Cvoid}, Ref{StepStats}), context(), st))

# ---- set_topology / rebuild_topology! (src/cloud.jl:200-228, src/surface.jl:167-205, src/volume.jl:97-125) ----------
# The containers stay the reference's; only the adjacency comes from the device.
knn_topology(pts, k::Int) = WhatsThePoint.KNNTopology(build_knn_neighbors(pts, k), k)
radius_topology(pts, radius) = WhatsThePoint.RadiusTopology(build_radius_neighbors(pts, radius), radius)
function rebuild_topology!(topo::WhatsThePoint.KNNTopology, pts)
    topo.neighbors = build_knn_neighbors(pts, topo.k)     # in place, like src/topology.jl:116-121
    return nothing
end
function rebuild_topology!(topo::WhatsThePoint.RadiusTopology, pts)
    topo.neighbors = build_radius_neighbors(pts, topo.radius)
    return nothing
end


# ---- multi-GPU, round 3: the whole sharded iteration behind the ABI (include/wtp.h: wtp_block_*) -------------------------
# One call per iteration replaces the body of `_relax!`'s loop (src/repel.jl:243-334) on a cloud cut into the boxes of an
# orthtree partition; INTEGRATION.md "Driving the multi-GPU iteration from Julia".  Written blind like the rest of this file.
struct BlockDesc
    rank::Int32; nranks::Int32; boxes::Ptr{Float64}; ghost_width::Float64; margin::Float64
end
mutable struct BlockInfo
    n_owned::Int64; n_ghost::Int64; n_sent_rows::Int64; n_recv_rows::Int64; n_emigrated::Int64; n_immigrated::Int64
    n_peers::Int32; widened::Int32; host_syncs::Int32; redone::Int32; ghost_width::Float64
    overlapped::Int32; reserved::Int32
    BlockInfo() = new()
end
block_grid(nranks::Int) = (p = Vector{Cint}(undef, 3); ccall((:wtp_block_grid, lib), Cint, (Cint, Ptr{Cint}), nranks, p); Tuple(Int.(p)))
block_morton_rank(ix, iy, iz, p) = Int(ccall((:wtp_block_morton_rank, lib), Cint, (Cint, Cint, Cint, Ptr{Cint}), ix, iy, iz, Cint[p...]))

# boxes: 6 x nranks Float64 ({lo xyz, hi xyz} per rank, identical on every rank); d_owned / d_gid: device pointers
function block_open(rank::Int, boxes::Matrix{Float64}, d_owned::Ptr{Cvoid}, d_gid::Ptr{Int64}, n_owned::Int, sd::SpacingDesc,
                    fd::ForceDesc, k::Int, alpha_lo, alpha_max; ghost_width::Float64, margin::Float64 = -1.0)
    GC.@preserve boxes begin
        desc = BlockDesc(Int32(rank), Int32(size(boxes, 2)), pointer(boxes), ghost_width, margin)
        check(context(), ccall((:wtp_block_open, lib), Cint,
              (Ptr{Cvoid}, Ref{BlockDesc}, Ptr{Cvoid}, Ptr{Int64}, Int64, Ref{SpacingDesc}, Ref{ForceDesc}, Cint, Cdouble, Cdouble),
              context(), desc, d_owned, d_gid, n_owned, sd, fd, k, alpha_lo, alpha_max))
    end
end
function block_step!(st::StepStats, info::BlockInfo)           # st: the GLOBAL statistics, the same on every rank
    check(context(), ccall((:wtp_block_step, lib), Cint, (Ptr{Cvoid}, Ref{StepStats}, Ref{BlockInfo}), context(), st, info))
    st
end
function block_run_until(max_iters::Int; tol = 1e-6, stall_after = 50, cv_target = 0.0)   # src/repel.jl:305-334 on the global numbers
    conv = Vector{Float64}(undef, max(max_iters, 1)); nd = Ref{Cint}(0); why = Ref{Cint}(0); st = StepStats()
    check(context(), ccall((:wtp_block_run_until, lib), Cint,
          (Ptr{Cvoid}, Cint, Cdouble, Cint, Cdouble, Ptr{Float64}, Ref{Cint}, Ref{Cint}, Ref{StepStats}),
          context(), max_iters, tol, stall_after, cv_target, conv, nd, why, st))
    conv[1:nd[]], Int(why[]), st
end
function block_get(d_xyz::Ptr{Cvoid}, d_gid::Ptr{Int64}, cap::Int)
    n = Ref{Int64}(0)
    check(context(), ccall((:wtp_block_get, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Int64}, Int64, Ref{Int64}), context(), d_xyz, d_gid, cap, n))
    Int(n[])
end
block_close() = check(context(), ccall((:wtp_block_close, lib), Cint, (Ptr{Cvoid},), context()))

end # module
