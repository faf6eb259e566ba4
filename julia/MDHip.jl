# MDHip.jl -- thin Julia binding of libmdhip.so (include/mdhip.h) that keeps the reference's
# names: Parameters, NVT/NVE, Potential, evaluate, LennardJones, PseudoHS, initialize_state,
# initialize_velocities, run_simulation!.  It is the drop-in for MolecularDynamics.jl's
# src/simulation.jl:40-178 path: user scripts keep their `run_simulation!(state, params, ens,
# total_steps, frequency, pathname)` call and the step loop runs on the GPU.
#
# NOT RUN in the build image (no Julia there).  It mirrors moleculardynamics/jl_amd/*.py one-to-one, which is
# what the test-suite drives through the same C ABI; tests/test_host_api.py parses every ccall / @ccall in this
# file and checks its symbol, argument count and argument types against include/mdhip.h.
module MDHip

using Random, Printf, LinearAlgebra, Statistics
using Distributions: Gamma
import CodecZstd                                    # compress=true: src/io.jl:207-223 (the reference's own dependency)

export Parameters, NVT, NVE, Brownian, Potential, evaluate, LennardJones, PseudoHS, Polydisperse,
       initialize_state, initialize_velocities, run_simulation!, LinearRamp, ExponentialRamp, fire_minimize!, minimize!,
       LennardJonesShifted, LennardJonesForceShifted, LennardJonesXPLOR, device_spec

const LIB = get(ENV, "MDHIP_LIB", joinpath(@__DIR__, "..", "moleculardynamics", "jl_amd", "csrc", "libmdhip.so"))

# ---- types: src/types.jl ---------------------------------------------------------------
abstract type Potential end
evaluate(pot::Potential, r::Real, s1::Real, s2::Real) =
    error("evaluate not implemented for potential type: $(typeof(pot))")          # src/types.jl:4-6
"""
Device description of a potential.  Either `(kind::Int, params::Vector{Float64})` for a built-in kind
(0 LennardJones, 1 PseudoHS, 2 Polydisperse, 3 modified LJ), or `(hip_source::String, entry::String, params)` for a
user potential: HIP source of `__device__ void entry(double r, double s1, double s2, const double* p, double* u,
double* f)` with f = -dU/dr -- the positional `evaluate(pot, r, sigma1, sigma2)` contract of src/pairwise.jl:31 --
compiled at run time (md_set_potential_source).  A subtype without a device form raises; nothing falls back to the CPU.
"""
device_spec(pot::Potential) = error("$(typeof(pot)) has no device form: define MDHip.device_spec")
energy_lrc(::Potential, N, V) = 0.0                                                # src/potentials.jl:281-293
pressure_lrc(::Potential, N, V) = 0.0

struct Parameters{P<:Potential,T<:AbstractFloat,N<:Integer}                        # src/types.jl:8-13
    ρ::T
    n_particles::N
    dt::T
    potential::P
end

abstract type Ensemble end
struct NVE <: Ensemble end
struct NVT{U,T<:AbstractFloat} <: Ensemble                                         # src/types.jl:36-44
    ktemp::U
    tau::T
end
NVT(ktemp::T, tau::T) where {T<:AbstractFloat} = NVT(step -> ktemp, tau)
struct Brownian{T<:AbstractFloat} <: Ensemble                                       # src/types.jl:46-49
    ktemp::T
end

# ---- potentials: src/potentials.jl ------------------------------------------------------
Base.@kwdef struct LennardJones <: Potential
    epsilon::Float64 = 1.0
    sigma::Float64 = 1.0
    r_cut::Float64 = 2.5
    tail_correction::Bool = false
end
function evaluate(p::LennardJones, r::Float64, s1::Float64, s2::Float64)          # src/potentials.jl:160-164,66-77
    σ = (s1 + s2) / 2.0
    r >= p.r_cut && return (0.0, 0.0)
    sr = σ / r; sr2 = sr * sr; sr6 = sr2 * sr2 * sr2; sr12 = sr6 * sr6
    return (4.0 * p.epsilon * (sr12 - sr6), 24.0 * p.epsilon * (2.0 * sr12 - sr6) / r)
end
device_spec(p::LennardJones) = (0, [p.epsilon, p.sigma, p.r_cut])
function energy_lrc(p::LennardJones, N, V)                                         # src/potentials.jl:111-141
    p.tail_correction || return 0.0
    ρ = N / V; x = p.sigma / p.r_cut
    return N * (8.0 * pi * ρ / 3.0) * (x^9 / 3.0 - x^3)
end
function pressure_lrc(p::LennardJones, N, V)
    p.tail_correction || return 0.0
    ρ = N / V; sr3 = (p.sigma / p.r_cut)^3
    return (16.0 * pi * ρ^2 / 3.0) * (2.0 * sr3^3 / 3.0 - sr3)
end

struct PseudoHS <: Potential
    lambda::Float64
end
PseudoHS() = PseudoHS(50.0)
const B_PARAM = 1.0204081632653061                                                  # src/potentials.jl:2-3
const A_PARAM = 134.5526623421209
function evaluate(p::PseudoHS, r::Float64, s1::Float64, s2::Float64)              # src/potentials.jl:11-29
    σ = (s1 + s2) / 2.0
    uij = 0.0; fij = 0.0
    if r < B_PARAM                                                                 # (the cutoff ignores sigma, as in the reference)
        uij = A_PARAM * ((σ / r)^p.lambda - (σ / r)^(p.lambda - 1.0)) + 1.0
        fij = A_PARAM * (p.lambda * (σ / r)^(p.lambda + 1.0) - (p.lambda - 1.0) * (σ / r)^p.lambda)
    end
    return (uij, fij)
end
device_spec(p::PseudoHS) = (1, [p.lambda])

Base.@kwdef struct Polydisperse <: Potential                                        # README.md:89-145
    rcut::Float64 = 1.25
    non_additivity::Float64 = 0.2
end
function evaluate(p::Polydisperse, r::Float64, s1::Float64, s2::Float64)          # README.md:89-145, positional (SURVEY.md D6)
    σ = 0.5 * (s1 + s2) * (1.0 - p.non_additivity * abs(s1 - s2))
    rc = p.rcut
    r < rc * σ || return (0.0, 0.0)
    c0 = -28.0 / rc^12; c2 = 48.0 / rc^14; c4 = -21.0 / rc^16
    u = (σ / r)^12 + c0 + c2 * (r / σ)^2 + c4 * (r / σ)^4
    f = 12.0 * σ^12 / r^13 - 2.0 * c2 * r / σ^2 - 4.0 * c4 * r^3 / σ^4
    return (u, f)
end
device_spec(p::Polydisperse) = (2, [p.rcut, p.non_additivity])

# shifted / force-shifted / XPLOR Lennard-Jones (src/potentials.jl:79-103,176-249; dead code in the reference):
# device kind 3 = MD_POT_LJ_MODIFIED, params {epsilon, sigma, r_cut, mode, r_on}; see include/mdhip.h
Base.@kwdef struct LennardJonesShifted <: Potential
    epsilon::Float64 = 1.0; sigma::Float64 = 1.0; r_cut::Float64 = 2.5
end
Base.@kwdef struct LennardJonesForceShifted <: Potential
    epsilon::Float64 = 1.0; sigma::Float64 = 1.0; r_cut::Float64 = 2.5
end
Base.@kwdef struct LennardJonesXPLOR <: Potential
    ϵ::Float64 = 1.0; σ::Float64 = 1.0; r_on::Float64 = 2.0; r_cut::Float64 = 2.5; tail_correction::Bool = false
end
device_spec(p::LennardJonesShifted) = (3, [p.epsilon, p.sigma, p.r_cut, 0.0, 0.0])
device_spec(p::LennardJonesForceShifted) = (3, [p.epsilon, p.sigma, p.r_cut, 1.0, 0.0])
device_spec(p::LennardJonesXPLOR) = (3, [p.ϵ, p.σ, p.r_cut, 2.0, p.r_on])

# ---- ramps: src/temperature_ramps.jl ----------------------------------------------------
struct LinearRamp; T_initial::Float64; T_final::Float64; n_steps::Int; end
function (r::LinearRamp)(step::Int)
    step > r.n_steps && return r.T_final
    step = clamp(step, 1, r.n_steps)
    r.n_steps == 1 && return r.T_final
    return r.T_initial + (r.T_final - r.T_initial) * (step - 1) / (r.n_steps - 1)
end
struct ExponentialRamp; T_initial::Float64; T_final::Float64; n_steps::Int; end
function (r::ExponentialRamp)(step::Int)
    step > r.n_steps && return r.T_final
    step = clamp(step, 1, r.n_steps)
    (r.n_steps == 1 || r.T_initial == r.T_final) && return r.T_final
    return r.T_initial * exp(log(r.T_final / r.T_initial) * (step - 1) / (r.n_steps - 1))
end

# ---- the handle -------------------------------------------------------------------------
mutable struct Device
    h::Ptr{Cvoid}
    dim::Int
    n::Int
end
function check(dev, rc)
    rc == 0 || error(unsafe_string(ccall((:md_last_error, LIB), Cstring, (Ptr{Cvoid},), dev === nothing ? C_NULL : dev.h)))
end
function Device(dim, n, unitcell::AbstractMatrix, cutoff; device_id=-1)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    box = Matrix{Float64}(unitcell)                                  # column-major d x d, as the ABI wants
    rc = ccall((:md_create, LIB), Cint, (Cint, Int64, Ptr{Float64}, Float64, Cint, Ptr{Ptr{Cvoid}}),
               dim, n, box, cutoff, device_id, h)
    check(nothing, rc)
    dev = Device(h[], dim, n)
    finalizer(d -> ccall((:md_destroy, LIB), Cint, (Ptr{Cvoid},), d.h), dev)
    return dev
end

# Vector{<:AbstractVector} (the reference's Vector{MVector}: an array of pointers) <-> d x N matrix
pack(v, d) = (M = Matrix{Float64}(undef, d, length(v)); for (i, x) in enumerate(v); M[:, i] .= x; end; M)
unpack!(v, M) = (for i in eachindex(v); v[i] .= view(M, :, i); end; v)

mutable struct EnergyAndForces                                                      # src/types.jl:53-57
    energy::Float64
    virial::Float64
    forces::Vector{Vector{Float64}}
end
mutable struct ParticleSystem
    positions::Vector{Vector{Float64}}
    unitcell::Matrix{Float64}
    cutoff::Float64
    energy_and_forces::EnergyAndForces
    device::Device
end
mutable struct SimulationState                                                      # src/types.jl:15-32
    system::ParticleSystem
    diameters::Vector{Float64}
    rng::AbstractRNG
    unitcell::Matrix{Float64}
    velocities::Vector{Vector{Float64}}
    images::Matrix{Int32}
    dimension::Int
    nf::Float64
end

function initialize_velocities(ktemp, rng, n_particles, dimension)                  # src/initialization.jl:32-47
    V = randn(rng, dimension, n_particles)
    V .-= mean(V; dims=2)
    fs = sqrt(ktemp / (sum(abs2, V) / ((n_particles - 1) * dimension)))
    V .*= fs
    return [V[:, i] for i in 1:n_particles]
end

"initialize_state(params, pathname; dimension, cutoff, rng, unitcell, positions, diameters): src/initialization.jl:112-157.
Positions must be supplied (Packmol is not a dependency here)."
function initialize_state(params::Parameters, pathname::String; dimension::Int=3, cutoff=1.5,
                          rng::AbstractRNG=Random.Xoshiro(), unitcell=nothing, positions, diameters=nothing)
    n = length(positions)
    nf = dimension * (params.n_particles - 1.0)
    cell = unitcell === nothing ? Matrix{Float64}(I, dimension, dimension) .* (n / params.ρ)^(1.0 / dimension) :
           (unitcell isa Number ? Matrix{Float64}(I, dimension, dimension) .* unitcell : Matrix{Float64}(unitcell))
    diam = diameters === nothing ? ones(n) : Vector{Float64}(diameters)
    pos = [Vector{Float64}(p) for p in positions]
    forces = [zeros(dimension) for _ in 1:n]                                       # zero forces: src/initialization.jl:97-99
    dev = Device(dimension, n, cell, cutoff)
    sys = ParticleSystem(pos, cell, cutoff, EnergyAndForces(0.0, 0.0, forces), dev)
    return SimulationState(sys, diam, rng, cell, Vector{Vector{Float64}}(), zeros(Int32, dimension, n), dimension, nf)
end

function sum_noises(nf, rng)                                                        # src/thermostat.jl:1-18
    nf == 0.0 && return 0.0
    nf == 1.0 && return randn(rng)^2
    mod(nf, 2) == 0 && return 2.0 * rand(rng, Gamma(nf ÷ 2))
    return 2.0 * rand(rng, Gamma((nf - 1) ÷ 2)) + randn(rng)^2
end

# ---- device configuration: Potential -> md_set_potential / md_set_potential_source --------------------------------
function configure!(dev::Device, pot::Potential)
    spec = device_spec(pot)
    if length(spec) == 2
        kind, pp = spec
        p = Vector{Float64}(pp)
        check(dev, ccall((:md_set_potential, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Cint), dev.h, kind, p, length(p)))
    else
        src, entry, pp = spec
        p = Vector{Float64}(pp)
        check(dev, ccall((:md_set_potential_source, LIB), Cint, (Ptr{Cvoid}, Cstring, Cstring, Ptr{Float64}, Cint),
                         dev.h, String(src), String(entry), p, length(p)))
    end
    return nothing
end

function upload!(dev::Device, state::SimulationState; velocities::Bool=true)
    d = state.dimension
    X = pack(state.system.positions, d); F = pack(state.system.energy_and_forces.forces, d)
    V = velocities ? pack(state.velocities, d) : nothing
    check(dev, ccall((:md_upload, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Float64}),
                     dev.h, X, velocities ? V : C_NULL, F, state.images, state.diameters))
    return nothing
end

"positions (wrapped), velocities, forces, images as d x N matrices"
function download(dev::Device)
    X = Matrix{Float64}(undef, dev.dim, dev.n); V = similar(X); F = similar(X)
    IM = Matrix{Int32}(undef, dev.dim, dev.n)
    check(dev, ccall((:md_download, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
                     dev.h, X, V, F, IM))
    return X, V, F, IM
end

"start the asynchronous export of one frame (positions + images); collect it with `snapshot_end` after the next segment"
function snapshot_begin(dev::Device)
    check(dev, ccall((:md_snapshot_begin, LIB), Cint, (Ptr{Cvoid},), dev.h))
    return nothing
end

function snapshot_end(dev::Device)
    X = Matrix{Float64}(undef, dev.dim, dev.n)
    IM = Matrix{Int32}(undef, dev.dim, dev.n)
    check(dev, ccall((:md_snapshot_end, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Int32}), dev.h, X, IM))
    return X, IM
end

# ---- output: src/io.jl ---------------------------------------------------------------------------------------------
function generate_log_times(; max_iter::Int=10000, logn::Int=40, logbase::Float64=1.35)   # src/io.jl:17-36
    dtime = Int[]
    maxlog = floor(Int, logbase^logn)
    for j in 0:max_iter, i in 0:logn
        push!(dtime, floor(Int, j * maxlog + logbase^i))
    end
    logs = sort(unique(dtime))
    open("new-log-times.txt", "w") do file                                          # src/io.jl:1-15 (in the CWD, as the reference)
        write(file, "#maxsnap=$logn,base=$logbase\n")
        for l in logs; write(file, "$l\n"); end
    end
    return logs
end

"extended XYZ, src/io.jl:42-70 (X: d x N matrix; \"%lf\" prints 6 decimals)"
function write_to_file(filepath, step, unitcell, n_particles, X::AbstractMatrix, diameters, dimension; mode="a")
    open(filepath, mode) do io
        println(io, n_particles)
        flat = join([string(unitcell[i, j]) for i in 1:dimension, j in 1:dimension], " ")
        @printf(io, "Lattice=\"%s\" Properties=type:I:1:id:I:1:radius:R:1:pos:R:%d Time=%.6g\n", flat, dimension, step)
        for i in 1:n_particles
            @printf(io, "%d %d %f", 1, i, diameters[i] / 2.0)
            for d in 1:dimension; @printf(io, " %f", X[d, i]); end
            @printf(io, "\n")
        end
    end
    return nothing
end

"LAMMPS dump with wrapped and unwrapped coordinates, src/io.jl:96-170 (id = i, type = 1, radius = sigma/2)"
function write_to_file_lammps(filepath, step, unitcell, n_particles, X::AbstractMatrix, IM::AbstractMatrix, diameters,
                              dimension; mode="w")
    open(filepath, mode) do io
        @printf(io, "ITEM: TIMESTEP\n%d\n", step)
        @printf(io, "ITEM: NUMBER OF ATOMS\n%d\n", n_particles)
        boxmat = zeros(3, 3); boxmat[1:dimension, 1:dimension] .= unitcell
        if dimension == 2
            @printf(io, "ITEM: BOX BOUNDS xy pp pp\n")
            @printf(io, "%f %f %f\n", 0.0, norm(boxmat[:, 1]), boxmat[1, 2])
            @printf(io, "%f %f 0.0\n", 0.0, norm(boxmat[:, 2]))
            @printf(io, "%f %f 0.0\n", 0.0, 1.0)
            @printf(io, "ITEM: ATOMS id type radius x y xu yu\n")
        elseif dimension == 3
            @printf(io, "ITEM: BOX BOUNDS xy xz yz pp pp pp\n")
            @printf(io, "%f %f %f\n", 0.0, norm(boxmat[:, 1]), boxmat[1, 2])
            @printf(io, "%f %f %f\n", 0.0, norm(boxmat[:, 2]), boxmat[2, 3])
            @printf(io, "%f %f %f\n", 0.0, norm(boxmat[:, 3]), boxmat[1, 3])
            @printf(io, "ITEM: ATOMS id type radius x y z xu yu zu\n")
        else
            error("Unsupported dimension: $dimension")
        end
        for i in 1:n_particles
            uw = X[:, i] .+ unitcell * IM[:, i]                                       # unwrapped: src/io.jl:77-85
            if dimension == 2
                @printf(io, "%d %d %f %f %f %f %f\n", i, 1, diameters[i] / 2.0, X[1, i], X[2, i], uw[1], uw[2])
            else
                @printf(io, "%d %d %f %f %f %f %f %f %f\n", i, 1, diameters[i] / 2.0, X[1, i], X[2, i], X[3, i],
                        uw[1], uw[2], uw[3])
            end
        end
    end
    return nothing
end

function compress_zstd(filepath)                                                     # src/io.jl:207-223
    open(filepath, "r") do infile
        open(CodecZstd.ZstdCompressorStream, filepath * ".zst", "w") do outfile
            write(outfile, read(infile))
        end
    end
    rm(filepath)
    return nothing
end

function open_files(pathname, traj_name, thermo_name)                                # src/io.jl:225-239
    files = (joinpath(pathname, traj_name), joinpath(pathname, thermo_name))
    for f in files; isfile(f) && rm(f); end
    return files
end

"""
run_simulation!(state, params, ensemble, total_steps, frequency, pathname; traj_name, thermo_name, compress, log_times)
-- src/simulation.jl:40-178 (NVE / NVT) and :181-308 (Brownian).  Mutates `state`, returns nothing.  The step loop
runs device-resident inside libmdhip; this driver cuts the run into segments that end on the reference's output
steps (step % frequency == 0, 0-based), draws the thermostat's random numbers on the host in the reference's
order, and writes the thermo line, the LAMMPS frames, the log-spaced snapshots and final.xyz.
"""
function run_simulation!(state::SimulationState, params::Parameters, ensemble::Ensemble, total_steps::Int,
                         frequency::Int, pathname::String; traj_name::String="trajectory.xyz",
                         thermo_name::String="thermo.txt", compress::Bool=false, log_times::Bool=false)
    dev = state.system.device; d = state.dimension; n = params.n_particles
    brownian = ensemble isa Brownian
    configure!(dev, params.potential)
    upload!(dev, state; velocities=!brownian)
    trajectory_file, thermo_file = open_files(pathname, traj_name, thermo_name)
    open(io -> println(io, "# Step Energy Temperature Pressure"), thermo_file, "a")
    volume = abs(det(state.unitcell))                                                # src/simulation.jl:7-9
    nvt = ensemble isa NVT
    # Brownian method: the device's noise stream is keyed by one draw of state.rng; the virial is sampled every 10th
    # step and averaged at the output steps (src/simulation.jl:253-266)
    brown_seed = brownian ? rand(state.rng, UInt64) >> 1 : UInt64(0)
    vir_sum = 0.0; vir_cnt = 0.0
    snapshot_times = log_times ? vcat(0, generate_log_times()) : Int[]                # src/simulation.jl:80-87
    snap_i = 1
    step = 0
    uwk = zeros(3); bout = zeros(4)
    # A frame is exported asynchronously (snapshot_begin) and collected after the NEXT segment has run: its device-to-host
    # copy overlaps that segment.  `pending` = the files the frame in flight goes to: (path, step, mode).
    pending = Tuple{String,Int,String}[]
    function collect_frame!()
        isempty(pending) && return
        X, IM = snapshot_end(dev)
        for (path, at, mode) in pending
            write_to_file_lammps(path, at, state.unitcell, n, X, IM, state.diameters, d; mode=mode)
        end
        empty!(pending)
    end
    while step < total_steps
        next_out = mod(step, frequency) == 0 ? step : (step ÷ frequency + 1) * frequency
        if log_times
            while snap_i <= length(snapshot_times) && snapshot_times[snap_i] < step; snap_i += 1; end
            snap_i <= length(snapshot_times) && (next_out = min(next_out, snapshot_times[snap_i]))
        end
        last = min(next_out, total_steps - 1)
        ns = last - step + 1
        if brownian
            rc = @ccall gc_safe=true LIB.md_run_brownian(dev.h::Ptr{Cvoid}, ns::Int64, params.dt::Float64,
                              ensemble.ktemp::Float64, brown_seed::UInt64, step::Int64, 10::Int64, bout::Ptr{Float64})::Cint
            check(dev, rc)
            uwk[1] = bout[1]; uwk[2] = bout[2]; uwk[3] = 0.0
            vir_sum += bout[3]; vir_cnt += bout[4]
        else
            kt = nvt ? Float64[ensemble.ktemp(s + 1) for s in step:last] : Float64[]   # step+1: src/simulation.jl:108
            r1 = zeros(nvt ? ns : 0); r2 = zeros(nvt ? ns : 0)
            if nvt
                for s in 1:ns
                    r1[s] = randn(state.rng)                                       # draw order: src/thermostat.jl:32-33
                    r2[s] = sum_noises(state.nf - 1, state.rng)
                end
            end
            # long-running and allocation-free on the Julia side: safe to run GC-safe
            rc = @ccall gc_safe=true LIB.md_run(dev.h::Ptr{Cvoid}, ns::Int64, params.dt::Float64, (nvt ? 1 : 0)::Cint,
                              (nvt ? ensemble.tau : 0.0)::Float64, state.nf::Float64, kt::Ptr{Float64}, r1::Ptr{Float64},
                              r2::Ptr{Float64}, uwk::Ptr{Float64})::Cint
            check(dev, rc)
        end
        collect_frame!()                 # the frame exported before this segment
        step = last + 1
        if mod(last, frequency) == 0                                               # src/simulation.jl:118-136
            if brownian
                T = ensemble.ktemp                                                 # src/simulation.jl:259-266
                e = uwk[1] / n
                P = vir_sum / (d * max(vir_cnt, 1.0) * volume) + params.ρ * ensemble.ktemp
                vir_sum = 0.0; vir_cnt = 0.0
            else
                T = 2.0 * uwk[3] / state.nf
                e = (uwk[1] + energy_lrc(params.potential, n, volume)) / n         # :120-124
                P = uwk[2] / (d * volume) + params.ρ * T + pressure_lrc(params.potential, n, volume)   # :128-131
            end
            open(io -> @printf(io, "%d %.6f %.6f %.6f\n", last, e, T, P), thermo_file, "a")
            state.system.energy_and_forces.energy = uwk[1]; state.system.energy_and_forces.virial = uwk[2]
            push!(pending, (trajectory_file, last, "a"))                           # src/simulation.jl:139-151
        end
        if log_times && snap_i <= length(snapshot_times) && snapshot_times[snap_i] == last   # :153-171
            push!(pending, (joinpath(pathname, "snapshot.$(last)"), last, "w"))
            snap_i += 1
        end
        isempty(pending) || snapshot_begin(dev)      # gather + copy to pinned memory: overlaps the next segment
    end
    collect_frame!()
    X, V, F, IM = download(dev)
    unpack!(state.system.positions, X); unpack!(state.system.energy_and_forces.forces, F)
    brownian || (state.velocities = [V[:, i] for i in 1:n])
    state.images .= IM
    # finalize_simulation!: src/simulation.jl:11-36
    write_to_file(joinpath(pathname, "final.xyz"), total_steps, state.unitcell, n, X, state.diameters, d; mode="w")
    compress && isfile(trajectory_file) && compress_zstd(trajectory_file)
    return nothing
end

# ---- fire_minimize!: src/minimize.jl:31-135 (same keywords and defaults; returns (energy, true) or nothing) ----
function fire_minimize!(state::SimulationState, params::Parameters; dimension::Int=2, max_steps::Int=10000,
                        tol::Float64=1e-6, dt_initial::Float64=0.01, dt_max::Float64=0.1, alpha0::Float64=0.1,
                        f_inc::Float64=1.2, f_dec::Float64=0.2, Nmin::Int=5)
    dev = state.system.device; d = state.dimension; n = params.n_particles
    configure!(dev, params.potential)
    X = pack(state.system.positions, d); F = pack(state.system.energy_and_forces.forces, d)
    check(dev, ccall((:md_upload, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Float64}),
                     dev.h, X, C_NULL, F, state.images, state.diameters))
    steps = Ref{Int64}(0); conv = Ref{Cint}(0); energy = Ref{Float64}(0.0); frms = Ref{Float64}(0.0)
    rc = @ccall gc_safe=true LIB.md_fire_minimize(dev.h::Ptr{Cvoid}, max_steps::Int64, tol::Float64, dt_initial::Float64,
                      dt_max::Float64, alpha0::Float64, f_inc::Float64, f_dec::Float64, Nmin::Cint, steps::Ptr{Int64},
                      conv::Ptr{Cint}, energy::Ptr{Float64}, frms::Ptr{Float64})::Cint
    check(dev, rc)
    check(dev, ccall((:md_download, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
                     dev.h, X, C_NULL, F, state.images))
    unpack!(state.system.positions, X); unpack!(state.system.energy_and_forces.forces, F)
    state.system.energy_and_forces.energy = energy[]
    conv[] != 0 && return energy[], true
    @warn "FIRE did not converge after $(max_steps) steps; final F_norm = $(frms[])"
    return nothing
end

"minimize!(state, params, pathname, dimension; method=:FIRE, save_config=\"minimized.xyz\", kwargs...): src/minimize.jl:166-197"
function minimize!(state::SimulationState, params::Parameters, pathname::String, dimension::Int; method::Symbol=:FIRE,
                   save_config::String="minimized.xyz", kwargs...)
    if method == :FIRE
        fire_minimize!(state, params; dimension=dimension, kwargs...)
    else
        error("Unknown minimization method: $method")
    end
    X = pack(state.system.positions, state.dimension)
    write_to_file(joinpath(pathname, save_config), 0, state.unitcell, params.n_particles, X, state.diameters, dimension)
    return nothing
end

end # module
