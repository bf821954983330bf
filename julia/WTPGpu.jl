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

# The loop of _relax! with the sweep on the device; stop rules exactly as src/repel.jl:305-334.
# (`constrain`, kick and trace stay with the caller's code path: see INTEGRATION.md for the octree method.)
function relax!(p, p_old, snap, spacing, force_model; n_fixed, α_lo, α_max, k, max_iters, tol, rebuild_every,
                stall_after = 0, cv_target = 0.0)
    rebuild_every >= 1 || throw(ArgumentError("rebuild_every must be ≥ 1"))
    s = ustrip.(spacing.(snap))
    D, T, n_move = relax_init(snap, n_fixed, all(==(first(s)), s) ? first(s) : s, force_model, min(k, length(snap)),
                              ustrip(α_lo), ustrip(α_max))
    conv = T[]; best_cv = Inf; last_improvement = 0; i = 1
    try
        while i <= max_iters
            st = relax_step((i - 1) % rebuild_every == 0)
            push!(conv, T(st.max_force))
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
        raws = relax_get(D, T, n_move)
        u = unit(Meshes.to(first(p))[1])
        for j in eachindex(p)
            p[j] = Meshes.Point((raws[j] .* u)...)
        end
    finally
        relax_end()
    end
    return conv
end

end # module
