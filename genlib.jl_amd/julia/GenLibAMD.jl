# GenLibAMD.jl -- drop-in MI355X path for GenLib.jl's dense kinship matrix.
#
# `GenLibAMD.phi` has the signature, keyword arguments, printed lines, return type and error
# behaviour of `GenLib.phi(pedigree::Pedigree, probandIDs; verbose, compute)`
# (GenLib.jl v0.1.4, src/compute.jl:233-304); the level sweep (src/compute.jl:269-303) runs
# in libgenphi.so (include/genphi.h) on the GPU instead of under Threads.@threads.
# Everything else (gen.genealogy, gen.pro, the Pedigree container) stays GenLib.jl's own.
#
# NOT exercised in the build container (no Julia there); see INTEGRATION.md.
module GenLibAMD

import GenLib

const libgenphi = get(ENV, "GENPHI_LIB", joinpath(@__DIR__, "..", "lib", "libgenphi.so"))

struct GenphiOpts              # mirrors genphi_opts (include/genphi.h)
    device::Int32
    kernel::Int32
    row_begin::Int64
    row_end::Int64
    timing::Int32
    reserved::Int32
end

last_error() = unsafe_string(ccall((:genphi_last_error, libgenphi), Cstring, ()))

function check(rc::Cint)
    rc == 0 && return
    msg = last_error()
    # GENPHI_ERR_UNKNOWN_ID / GENPHI_ERR_ORDER are KeyErrors in the reference
    (rc == 1 || rc == 2) ? throw(KeyError(msg)) : error("libgenphi: $msg (code $rc)")
end

"""
    phi(pedigree::GenLib.Pedigree, probandIDs::Vector{Int} = GenLib.pro(pedigree);
        verbose::Bool = false, compute::Bool = true, device::Integer = -1)

Square `Matrix{Float32}` of pairwise kinship coefficients between probands, bit-identical to
`GenLib.phi`, computed on an MI355X.
"""
function phi(pedigree::GenLib.Pedigree, probandIDs::Vector{Int} = GenLib.pro(pedigree);
             verbose::Bool = false, compute::Bool = true, device::Integer = -1)
    # flatten in rank order (the traversal of GenLib.genout, src/output.jl:24-29, kept at 64 bit)
    n = length(pedigree)
    ind = Vector{Int64}(undef, n); father = zeros(Int64, n); mother = zeros(Int64, n)
    for (k, individual) in enumerate(values(pedigree))
        ind[k] = individual.ID
        isnothing(individual.father) || (father[k] = individual.father.ID)
        isnothing(individual.mother) || (mother[k] = individual.mother.ID)
    end
    plan = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve ind father mother probandIDs begin
        check(ccall((:genphi_plan_create, libgenphi), Cint,
                    (Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Int64, Ptr{Int64}, Ptr{Ptr{Cvoid}}),
                    n, ind, father, mother, length(probandIDs), probandIDs, plan))
    end
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
        if verbose                                       # lines of src/compute.jl:281-284
            for k in 1:nsteps
                println("Running step $k of $nsteps ($(cut[k]) founders, $(cut[k+1]) probands, $(dragged[k]) both).")
            end
        end
        N = Int(ccall((:genphi_plan_n_probands, libgenphi), Int64, (Ptr{Cvoid},), plan[]))
        Φ = Matrix{Float32}(undef, N, N)                 # symmetric: row-major == column-major
        opts = Ref(GenphiOpts(Int32(device), 0, 0, 0, 0, 0))
        GC.@preserve Φ check(ccall((:genphi_compute_f32, libgenphi), Cint,
                                   (Ptr{Cvoid}, Ptr{Float32}, Ptr{GenphiOpts}, Ptr{Cvoid}),
                                   plan[], Φ, opts, C_NULL))
        return Φ
    finally
        ccall((:genphi_plan_destroy, libgenphi), Cvoid, (Ptr{Cvoid},), plan[])
    end
end

end # module
