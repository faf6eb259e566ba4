# MDHip.jl -- thin Julia binding of libmdhip.so (include/mdhip.h) that keeps the reference's
# names: Parameters, NVT/NVE, Potential, evaluate, LennardJones, PseudoHS, initialize_state,
# initialize_velocities, run_simulation!.  It is the drop-in for MolecularDynamics.jl's
# src/simulation.jl:40-178 path: user scripts keep their `run_simulation!(state, params, ens,
# total_steps, frequency, pathname)` call and the step loop runs on the GPU.
#
# NOT TESTED in the build image (no Julia there).  It mirrors moleculardynamics/jl_amd/*.py
# one-to-one, which is what the test-suite drives through the same C ABI.
module MDHip

using Random, Printf, LinearAlgebra, Statistics
using Distributions: Gamma

export Parameters, NVT, NVE, Potential, evaluate, LennardJones, PseudoHS, Polydisperse,
       initialize_state, initialize_velocities, run_simulation!, LinearRamp, ExponentialRamp, fire_minimize!,
       LennardJonesShifted, LennardJonesForceShifted, LennardJonesXPLOR

const LIB = get(ENV, "MDHIP_LIB", joinpath(@__DIR__, "..", "moleculardynamics", "jl_amd", "csrc", "libmdhip.so"))

# ---- types: src/types.jl ---------------------------------------------------------------
abstract type Potential end
evaluate(pot::Potential, r::Real, s1::Real, s2::Real) =
    error("evaluate not implemented for potential type: $(typeof(pot))")          # src/types.jl:4-6
"Device description of a potential: (kind, params) for a built-in, or (hip_source, entry, params)."
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
device_spec(p::PseudoHS) = (1, [p.lambda])

Base.@kwdef struct Polydisperse <: Potential                                        # README.md:89-145
    rcut::Float64 = 1.25
    non_additivity::Float64 = 0.2
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

"run_simulation!: src/simulation.jl:40-178 (NVE / NVT method).  Mutates `state`, returns nothing."
function run_simulation!(state::SimulationState, params::Parameters, ensemble::Ensemble, total_steps::Int,
                         frequency::Int, pathname::String; thermo_name::String="thermo.txt")
    dev = state.system.device; d = state.dimension; n = params.n_particles
    kind, pp = device_spec(params.potential)
    check(dev, ccall((:md_set_potential, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Cint), dev.h, kind, pp, length(pp)))
    X = pack(state.system.positions, d); V = pack(state.velocities, d); F = pack(state.system.energy_and_forces.forces, d)
    check(dev, ccall((:md_upload, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Float64}),
                     dev.h, X, V, F, state.images, state.diameters))
    thermo_file = joinpath(pathname, thermo_name)
    isfile(thermo_file) && rm(thermo_file)
    open(io -> println(io, "# Step Energy Temperature Pressure"), thermo_file, "a")
    volume = abs(det(state.unitcell))
    nvt = ensemble isa NVT
    step = 0
    uwk = zeros(3)
    while step < total_steps
        next_out = mod(step, frequency) == 0 ? step : (step ÷ frequency + 1) * frequency
        last = min(next_out, total_steps - 1)
        ns = last - step + 1
        kt = nvt ? Float64[ensemble.ktemp(s + 1) for s in step:last] : Float64[]   # step+1: src/simulation.jl:108
        r1 = zeros(nvt ? ns : 0); r2 = zeros(nvt ? ns : 0)
        if nvt
            for s in 1:ns
                r1[s] = randn(state.rng)                                           # draw order: src/thermostat.jl:32-33
                r2[s] = sum_noises(state.nf - 1, state.rng)
            end
        end
        # long-running and allocation-free on the Julia side: safe to run GC-safe
        rc = @ccall gc_safe=true LIB.md_run(dev.h::Ptr{Cvoid}, ns::Int64, params.dt::Float64, (nvt ? 1 : 0)::Cint,
                          (nvt ? ensemble.tau : 0.0)::Float64, state.nf::Float64, kt::Ptr{Float64}, r1::Ptr{Float64},
                          r2::Ptr{Float64}, uwk::Ptr{Float64})::Cint
        check(dev, rc)
        step = last + 1
        if mod(last, frequency) == 0                                               # src/simulation.jl:118-136
            T = 2.0 * uwk[3] / state.nf
            e = (uwk[1] + energy_lrc(params.potential, n, volume)) / n
            P = uwk[2] / (d * volume) + params.ρ * T + pressure_lrc(params.potential, n, volume)
            open(io -> @printf(io, "%d %.6f %.6f %.6f\n", last, e, T, P), thermo_file, "a")
            state.system.energy_and_forces.energy = uwk[1]; state.system.energy_and_forces.virial = uwk[2]
        end
    end
    check(dev, ccall((:md_download, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
                     dev.h, X, V, F, state.images))
    unpack!(state.system.positions, X); unpack!(state.system.energy_and_forces.forces, F)
    state.velocities = [V[:, i] for i in 1:n]
    return nothing
end

# ---- fire_minimize!: src/minimize.jl:31-135 (same keywords and defaults; returns (energy, true) or nothing) ----
function fire_minimize!(state::SimulationState, params::Parameters; dimension::Int=2, max_steps::Int=10000,
                        tol::Float64=1e-6, dt_initial::Float64=0.01, dt_max::Float64=0.1, alpha0::Float64=0.1,
                        f_inc::Float64=1.2, f_dec::Float64=0.2, Nmin::Int=5)
    dev = state.system.device; d = state.dimension; n = params.n_particles
    kind, pp = device_spec(params.potential)
    check(dev, ccall((:md_set_potential, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Cint), dev.h, kind, pp, length(pp)))
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

end # module
