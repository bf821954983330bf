# bench_reference.jl — the reference's own OhMyThreads repel sweep on the benchmark cloud
# (SURVEY.md §8d "CPU baseline").  bench.py runs this ONLY when `julia` and an already installed
# WhatsThePoint resolve on the box; it never installs or fetches anything.  UNTESTED here: the build
# image has no Julia toolchain (same caveat as the shim in INTEGRATION.md).
#
#   julia --threads=auto julia/bench_reference.jl <n_points> <iters>
# prints one line:  WTP_REFERENCE <Mpoints_per_s> <threads> <seconds>
using WhatsThePoint
using Meshes
using Unitful: m

# counter-based generator of SURVEY.md §8d (identical to oracle/wtp_oracle.c and synth.py)
function splitmix64(z::UInt64)
    z += 0x9E3779B97F4A7C15
    z = (z ⊻ (z >> 30)) * 0xBF58476D1CE4E5B9
    z = (z ⊻ (z >> 27)) * 0x94D049BB133111EB
    return z ⊻ (z >> 31)
end
coord(seed, i, a) = Float32(splitmix64((UInt64(seed) << 40) + UInt64(3 * i + a)) >> 40) * (1.0f0 / 16777216.0f0)

function main()
    n = parse(Int, ARGS[1])
    iters = parse(Int, ARGS[2])
    seed = 20260821
    pts = [Meshes.Point(coord(seed, i, 0) * m, coord(seed, i, 1) * m, coord(seed, i, 2) * m) for i in 0:(n - 1)]
    s = Float32(n)^(-1.0f0 / 3.0f0)
    spacing = ConstantSpacing(s * m)
    p, p_old, snap = copy(pts), copy(pts), copy(pts)
    run(k) = WhatsThePoint._relax!(
        p, p_old, snap, spacing, ClippedSpacingForce(0.2), (id, xi, x) -> x;
        n_fixed = 0, n_protected = 0, α_lo = s / 2000, α_max = s / 20, k = 21, max_iters = k, tol = 0.0,
        rebuild_every = 1, kick_after = 0, trace = nothing, stall_after = 0, cv_target = 0.0,
    )
    run(1)                                   # compile
    p .= pts; p_old .= pts; snap .= pts
    t = @elapsed run(iters)
    println("WTP_REFERENCE ", n * iters / t / 1.0e6, " ", Threads.nthreads(), " ", t)
end

main()
