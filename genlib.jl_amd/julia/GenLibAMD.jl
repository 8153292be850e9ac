# GenLibAMD.jl -- drop-in MI355X path for GenLib.jl's dense kinship matrix.
#
# `GenLibAMD.phi` has the signature, keyword arguments, printed lines, return type and error
# behaviour of `GenLib.phi(pedigree::Pedigree, probandIDs; verbose, compute)`
# (GenLib.jl v0.1.4, src/compute.jl:233-304); the level sweep (src/compute.jl:269-303) runs
# in libgenphi.so (include/genphi.h) on the GPU instead of under Threads.@threads.
# Everything else (gen.genealogy, gen.pro, the Pedigree container) stays GenLib.jl's own.
#
# NOT exercised in the build container (no Julia there, nor on the GPU boxes); see INTEGRATION.md.
module GenLibAMD

import GenLib

const libgenphi = get(ENV, "GENPHI_LIB", joinpath(@__DIR__, "..", "lib", "libgenphi.so"))

struct GenphiOpts              # mirrors genphi_opts (include/genphi.h)
    device::Int32
    kernel::Int32
    row_begin::Int64
    row_end::Int64
    timing::Int32
    flags::Int32               # GENPHI_FLAG_*: 1 = no hipGraph replay, 2 = Float64 level matrices (GENPHI_FLAG_STORAGE_F64)
end
const FLAG_STORAGE_F64 = Int32(2)

last_error() = unsafe_string(ccall((:genphi_last_error, libgenphi), Cstring, ()))

function check(rc::Cint)
    rc == 0 && return
    msg = last_error()
    # GENPHI_ERR_UNKNOWN_ID / GENPHI_ERR_ORDER are KeyErrors in the reference
    (rc == 1 || rc == 2) ? throw(KeyError(msg)) : error("libgenphi: $msg (code $rc)")
end

# flatten in rank order (the traversal of GenLib.genout, src/output.jl:24-29, kept at 64 bit)
function flatten(pedigree::GenLib.Pedigree)
    n = length(pedigree)
    ind = Vector{Int64}(undef, n); father = zeros(Int64, n); mother = zeros(Int64, n)
    sex = Vector{Int64}(undef, n)
    for (k, individual) in enumerate(values(pedigree))
        ind[k] = individual.ID
        sex[k] = individual.sex
        isnothing(individual.father) || (father[k] = individual.father.ID)
        isnothing(individual.mother) || (mother[k] = individual.mother.ID)
    end
    ind, father, mother, sex
end

# The plans of the last calls, per pedigree: the host prologue of GenLib.phi (src/compute.jl:236-262: levelisation, cut sets, index copy)
# depends on (pedigree, probandIDs) alone, so a repeated call skips planning, upload and the calibration of the sparse cuts and pays the
# sweep and the copy (genea140: 6.2 ms for a first call, 0.7 ms for a repeated one).  A plan that holds more than PLAN_CACHE_DEVICE_BYTES
# of device memory is destroyed at the end of its call.  `release_cached()` drops the plans and what the library itself keeps.
const PLAN_CACHE_ENTRIES = 4
const PLAN_CACHE_DEVICE_BYTES = 2 << 30
const plan_cache = WeakKeyDict{GenLib.Pedigree, Vector{Tuple{Vector{Int}, Int, Ptr{Cvoid}}}}()   # (probandIDs, device, plan), least recent first

destroy_plan(plan::Ptr{Cvoid}) = ccall((:genphi_plan_destroy, libgenphi), Cvoid, (Ptr{Cvoid},), plan)

function release_cached()
    for entries in values(plan_cache), (_, _, plan) in entries
        destroy_plan(plan)
    end
    empty!(plan_cache)
    ccall((:genphi_release_cached, libgenphi), Cvoid, ())      # device blocks, streams and pinned staging the library keeps between calls
end

# tuning: nothing (the library's defaults; GENPHI_* environment hooks only under GENPHI_ENV_HOOKS=1) or settings for this plan alone,
# e.g. Dict("SPARSE_K" => -1) (genphi_plan_create_tuned; such plans are not cached)
function create_plan(pedigree::GenLib.Pedigree, probandIDs::Vector{Int}, tuning)
    ind, father, mother, _ = flatten(pedigree)
    plan = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve ind father mother probandIDs begin
        if isnothing(tuning)
            check(ccall((:genphi_plan_create, libgenphi), Cint,
                        (Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Int64, Ptr{Int64}, Ptr{Ptr{Cvoid}}),
                        length(ind), ind, father, mother, length(probandIDs), probandIDs, plan))
        else
            t = ccall((:genphi_tuning_create, libgenphi), Ptr{Cvoid}, ())
            try
                for (name, value) in tuning
                    check(ccall((:genphi_tuning_set, libgenphi), Cint, (Ptr{Cvoid}, Cstring, Cstring), t, string(name), string(value)))
                end
                check(ccall((:genphi_plan_create_tuned, libgenphi), Cint,
                            (Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Int64, Ptr{Int64}, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}),
                            length(ind), ind, father, mother, length(probandIDs), probandIDs, t, plan))
            finally
                ccall((:genphi_tuning_destroy, libgenphi), Cvoid, (Ptr{Cvoid},), t)
            end
        end
    end
    plan[]
end

"""
    phi(pedigree::GenLib.Pedigree, probandIDs::Vector{Int} = GenLib.pro(pedigree);
        verbose::Bool = false, compute::Bool = true, device::Integer = -1, tuning = nothing)

Square `Matrix{Float32}` of pairwise kinship coefficients between probands, bit-identical to
`GenLib.phi`, computed on an MI355X.
"""
function phi(pedigree::GenLib.Pedigree, probandIDs::Vector{Int} = GenLib.pro(pedigree);
             verbose::Bool = false, compute::Bool = true, device::Integer = -1, tuning = nothing)
    entries = get!(() -> Tuple{Vector{Int}, Int, Ptr{Cvoid}}[], plan_cache, pedigree)
    hit = isnothing(tuning) ? findfirst(e -> e[2] == device && e[1] == probandIDs, entries) : nothing
    plan = Ref{Ptr{Cvoid}}(isnothing(hit) ? create_plan(pedigree, probandIDs, tuning) : entries[hit][3])
    isnothing(hit) || deleteat!(entries, hit)              # (re-inserted as the most recent one below, if the call succeeds)
    keep = false
    try
        nlev = Ref{Int32}(0); sizes = Ref{Ptr{Int64}}(C_NULL); both = Ref{Ptr{Int64}}(C_NULL)
        check(ccall((:genphi_plan_levels, libgenphi), Cint,
                    (Ptr{Cvoid}, Ptr{Int32}, Ptr{Ptr{Int64}}, Ptr{Ptr{Int64}}), plan[], nlev, sizes, both))
        nsteps = max(nlev[] - 1, 0)
        cut = unsafe_wrap(Array, sizes[], Int(nlev[])); dragged = unsafe_wrap(Array, both[], nsteps)
        if verbose || !compute                           # lines of src/compute.jl:257-260
            for i in 1:nsteps
                println("Step $i of $nsteps: $(cut[i]) founders, $(cut[i+1]) probands, $(dragged[i]) both.")
            end
        end
        compute || return nothing                        # src/compute.jl:264-266
        hook = nothing
        if verbose                                       # lines of src/compute.jl:281-284, printed INSIDE the level loop as there:
            # the library calls back right before it hands each level step to the GPU (genphi_plan_set_step_hook)
            hook = @cfunction($((step, n, _) -> (println("Running step $(step + 1) of $n ($(cut[step+1]) founders, $(cut[step+2]) probands, $(dragged[step+1]) both."); nothing)),
                              Cvoid, (Int32, Int32, Ptr{Cvoid}))
            check(ccall((:genphi_plan_set_step_hook, libgenphi), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), plan[], hook, C_NULL))
        end
        N = Int(ccall((:genphi_plan_n_probands, libgenphi), Int64, (Ptr{Cvoid},), plan[]))
        Φ = Matrix{Float32}(undef, N, N)                 # symmetric: row-major == column-major
        opts = Ref(GenphiOpts(Int32(device), 0, 0, 0, 0, 0))
        GC.@preserve Φ hook check(ccall((:genphi_compute_f32, libgenphi), Cint,
                                   (Ptr{Cvoid}, Ptr{Float32}, Ptr{GenphiOpts}, Ptr{Cvoid}),
                                   plan[], Φ, opts, C_NULL))
        isnothing(hook) || check(ccall((:genphi_plan_set_step_hook, libgenphi), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), plan[], C_NULL, C_NULL))
        if isnothing(tuning) && ccall((:genphi_plan_device_bytes, libgenphi), Int64, (Ptr{Cvoid},), plan[]) <= PLAN_CACHE_DEVICE_BYTES
            push!(entries, (copy(probandIDs), Int(device), plan[]))
            keep = true
            while length(entries) > PLAN_CACHE_ENTRIES
                destroy_plan(popfirst!(entries)[3])
            end
        end
        return Φ
    finally
        keep || destroy_plan(plan[])
    end
end

"""
    phi(individualᵢ::GenLib.Individual, individualⱼ::GenLib.Individual, pedigree::GenLib.Pedigree; device = -1)

Float64 kinship of a pair, as `GenLib.phi(individualᵢ, individualⱼ)` (src/compute.jl:66-95), from one
Float64 level sweep on the GPU instead of the un-memoised recursion.  (The reference's method needs no
pedigree argument because its `Individual`s carry pointers to their parents; the flat arrays the
library takes are built from the pedigree.)  Bit-identical while kinships are exactly representable in
Float64 (pedigrees less than ~26 generations deep), within 1e-15 relative beyond.
"""
function phi(individualᵢ::GenLib.Individual, individualⱼ::GenLib.Individual, pedigree::GenLib.Pedigree;
             device::Integer = -1)
    ind, father, mother, _ = flatten(pedigree)
    a = Int64[individualᵢ.ID]; b = Int64[individualⱼ.ID]; out = Vector{Float64}(undef, 1)
    GC.@preserve ind father mother a b out check(ccall((:genphi_phi_pairs, libgenphi), Cint,
        (Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Int32),
        length(ind), ind, father, mother, 1, a, b, out, Int32(device)))
    out[1]
end

"""
    f(pedigree::GenLib.Pedigree, IDs::Vector{Int}; device::Integer = -1)

Coefficients of inbreeding (`Vector{Float32}`), as `GenLib.f` (src/compute.jl:500-511), from ONE
Float64 level sweep over the parents on the GPU plus point lookups, instead of one un-memoised
pairwise recursion per individual; rounded to Float32 once, like the reference.
"""
function f(pedigree::GenLib.Pedigree, IDs::Vector{Int}; device::Integer = -1)
    coefficients = zeros(Float32, length(IDs))
    pairs = [(pedigree[ID].father, pedigree[ID].mother) for ID in IDs]       # KeyError on unknown ID
    known = findall(p -> !isnothing(p[1]) && !isnothing(p[2]), pairs)
    isempty(known) && return coefficients
    parents = sort(unique(vcat([pairs[k][1].ID for k in known], [pairs[k][2].ID for k in known])))
    rows = Int64[searchsortedfirst(parents, pairs[k][1].ID) - 1 for k in known]
    cols = Int64[searchsortedfirst(parents, pairs[k][2].ID) - 1 for k in known]
    ind, father, mother, _ = flatten(pedigree)
    plan = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve ind father mother parents begin
        check(ccall((:genphi_plan_create, libgenphi), Cint,
                    (Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Int64, Ptr{Int64}, Ptr{Ptr{Cvoid}}),
                    length(ind), ind, father, mother, length(parents), parents, plan))
    end
    try
        opts = Ref(GenphiOpts(Int32(device), 0, 0, 0, 0, FLAG_STORAGE_F64))
        check(ccall((:genphi_compute_device, libgenphi), Cint, (Ptr{Cvoid}, Ptr{GenphiOpts}, Ptr{Cvoid}),
                    plan[], opts, C_NULL))
        values64 = Vector{Float64}(undef, length(known))
        GC.@preserve rows cols values64 check(ccall((:genphi_result_entries, libgenphi), Cint,
            (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}), plan[], length(known), rows, cols, values64))
        coefficients[known] .= Float32.(values64)
    finally
        ccall((:genphi_plan_destroy, libgenphi), Cvoid, (Ptr{Cvoid},), plan[])
    end
    coefficients
end

"""
    KinshipMatrix, sparse_phi(pedigree, probandIDs = GenLib.pro(pedigree); device = -1)

As `GenLib.sparse_phi` / `GenLib.KinshipMatrix` (src/compute.jl:321-447, :31-46): `ϕ[ID₁, ID₂]`,
`show`, `phiMean(ϕ)`; computed on the GPU one depth at a time (csrc/sparse_phi.hip).
"""
mutable struct KinshipMatrix
    handle::Ptr{Cvoid}
    function KinshipMatrix(h::Ptr{Cvoid})
        ϕ = new(h)
        finalizer(x -> ccall((:genphi_sparse_destroy, libgenphi), Cvoid, (Ptr{Cvoid},), x.handle), ϕ)
    end
end

function sparse_phi(pedigree::GenLib.Pedigree, probandIDs::Vector{Int} = GenLib.pro(pedigree); device::Integer = -1)
    ind, father, mother, _ = flatten(pedigree)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve ind father mother probandIDs check(ccall((:genphi_sparse_phi, libgenphi), Cint,
        (Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Int64, Ptr{Int64}, Int32, Ptr{Ptr{Cvoid}}),
        length(ind), ind, father, mother, length(probandIDs), probandIDs, Int32(device), h))
    KinshipMatrix(h[])
end

function info(ϕ::KinshipMatrix)
    n = Ref{Int64}(0); nz = Ref{Int64}(0); total = Ref{Float64}(0); diagonal = Ref{Float64}(0)
    check(ccall((:genphi_sparse_info, libgenphi), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}),
                ϕ.handle, n, nz, total, diagonal))
    n[], nz[], total[], diagonal[]
end

function Base.getindex(ϕ::KinshipMatrix, ID₁::Int, ID₂::Int)
    a = Int64[ID₁]; b = Int64[ID₂]; out = Vector{Float64}(undef, 1)
    GC.@preserve a b out check(ccall((:genphi_sparse_get, libgenphi), Cint,
        (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}), ϕ.handle, 1, a, b, out))
    out[1]
end

function Base.show(io::IO, ::MIME"text/plain", ϕ::KinshipMatrix)
    n, nz, _, _ = info(ϕ)
    print(io, "$(n)×$(n) KinshipMatrix with $nz stored entries.")
end

function phiMean(ϕ::KinshipMatrix)::Float32
    n, _, total, diagonal = info(ϕ)
    (total - diagonal) / (n * (n - 1) / 2)
end

"""
    branching(pedigree::GenLib.Pedigree; pro = nothing, ancestors = nothing)

As `GenLib.branching` (src/extract.jl:65-186): the pedigree of the individuals on the paths
between the selected probands and ancestors; two linear sweeps in libgenphi instead of
recursive marking over a copied pointer graph.
"""
function branching(pedigree::GenLib.Pedigree; pro::Union{Vector{Int}, Nothing} = nothing,
                   ancestors::Union{Vector{Int}, Nothing} = nothing)
    ind, father, mother, sex = flatten(pedigree)
    n = Ref{Int64}(0)
    out = [Ref{Ptr{Int64}}(C_NULL) for _ in 1:4]
    nothing_or(v) = isnothing(v) ? Ptr{Int64}(C_NULL) : (isempty(v) ? pointer(ind) : pointer(v))
    GC.@preserve ind father mother sex pro ancestors begin
        check(ccall((:genphi_branching, libgenphi), Cint,
                    (Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Int64, Ptr{Int64}, Int64, Ptr{Int64},
                     Ptr{Int64}, Ptr{Ptr{Int64}}, Ptr{Ptr{Int64}}, Ptr{Ptr{Int64}}, Ptr{Ptr{Int64}}),
                    length(ind), ind, father, mother, sex,
                    isnothing(pro) ? 0 : length(pro), nothing_or(pro),
                    isnothing(ancestors) ? 0 : length(ancestors), nothing_or(ancestors),
                    n, out[1], out[2], out[3], out[4]))
    end
    cols = [copy(unsafe_wrap(Array, o[], Int(n[]))) for o in out]
    foreach(o -> ccall((:genphi_free, libgenphi), Cvoid, (Ptr{Cvoid},), o[]), out)
    # rebuild through GenLib's own constructor (already in rank order: sort = false)
    GenLib.genealogy(GenLib.DataFrame(ind = cols[1], father = cols[2], mother = cols[3], sex = cols[4]), sort = false)
end

end # module
