# IsokannHIP.jl -- binds libisokann_hip.so (include/isokann_hip.h, ABI version 2) into an ISOKANN.jl v2.0.0 host.
#
# The reference has no FFI on the train-step path: it is multiple dispatch on a Flux.Chain.  This file is the seam a maintainer
# adds (e.g. as ext/IsokannHIPExt.jl); nothing in src/ changes.  It
#   (1) owns the engine handle(s),
#   (2) puts a `HipModel` where `iso.model` (a Flux.Chain) stands and a `HipOpt` where `iso.opt` (an Optimisers state tree) stands,
#   (3) adds methods to the generic functions the reference's own code calls: run!, train_batch!, isotarget, chis, koopman,
#       chicoords, outputdim, gpu / cpu, save.
# `file:line` comments name the reference function each method takes the role of.
#
# STATUS: there is no Julia toolchain in the build container, so this file has been reviewed by reading only; every executable
# test of this repository drives the same C ABI through ctypes with column-major arrays exactly as `ccall` passes them
# (isokann.jl_amd/capi.py is its line-by-line twin, isokann.jl_amd/api.py the twin of the methods below).
module IsokannHIP

using ISOKANN, Flux, Optimisers
import ISOKANN: Iso, run!, train_batch!, isotarget, chis, chicoords, koopman, outputdim, inputdim, iscuda, gpu, cpu, save
import ISOKANN: SimulationData, FeaturesCoords, FeaturesAll, FeaturesAtoms, FeaturesPairs, TransformShiftscale, TransformISA, halfinds

const LIB = Ref{String}("libisokann_hip.so")      # set to isokann.jl_amd/lib/libisokann_hip.so of the build tree
const ABI_VERSION = Int32(2)
const MAXL = 8

# ---------------------------------------------------------------------------------------------------------------------
# struct isk_config, field by field (include/isokann_hip.h)
# ---------------------------------------------------------------------------------------------------------------------
struct IskConfig
    abi_version::Int32; device::Int32
    n_layers::Int32; widths::NTuple{MAXL + 1,Int32}
    activation::Int32; lastactivation::Int32
    layernorm::Int32; ln_mode::Int32; ln_eps::Float32
    opt_kind::Int32; lr::Float32; reg::Float32; beta1::Float32; beta2::Float32; opt_eps::Float32; rho::Float32
    featurizer::Int32; n_atoms::Int32; n_pairs::Int32; pairs::Ptr{Int32}
    featurize_every_iteration::Int32
    target_kind::Int32; isa_permute::Int32; isa_whitening::Int32
    minibatch::Int32; seed::UInt64
    forward_mode::Int32        # 0 f32-input MFMA | 1 split-bf16 | 3 split-f16 (both f32-equivalent) | 2 plain bf16 (C5; ~1e-2)
    force_layered::Int32
end

# isk_status values the shim cares about
const ISK_ECONSTANT_CHI, ISK_ECOLLAPSED, ISK_ESINGULAR = 4, 5, 6

function check(st::Integer)
    st == 0 && return nothing
    msg = unsafe_string(ccall((:isk_last_error, LIB[]), Cstring, ()))
    # the reference's three DomainErrors: src/isotarget.jl:39, src/iso.jl:186-189, src/isotarget.jl:94-97
    st in (ISK_ECONSTANT_CHI, ISK_ECOLLAPSED, ISK_ESINGULAR) && throw(DomainError(st, msg))
    error("isokann_hip status $st: $msg")
end

# ---------------------------------------------------------------------------------------------------------------------
# Flux / Optimisers objects -> isk_config
# ---------------------------------------------------------------------------------------------------------------------
act_id(f) = f === identity ? 0 : f === Flux.sigmoid || f === Flux.sigmoid_fast ? 1 : f === tanh || f === Flux.tanh_fast ? 2 :
            f === Flux.relu ? 3 : f === Flux.swish ? 4 : error("activation $f is not covered by the HIP kernels")

"the target `Iso(...)` would pick (src/iso.jl:28-34)"
default_target(model) = ISOKANN.outputdim(model) == 1 ? TransformShiftscale() : TransformISA()

"(kind, lr, reg, beta1, beta2, eps, rho) of OptimiserChain(WeightDecay(reg), Adam(lr) | Nesterov(lr, rho)) (src/models.jl:12,20)"
function opt_fields(opt::Optimisers.OptimiserChain)
    reg, rule = 0.0f0, nothing
    for o in opt.opts
        o isa Optimisers.WeightDecay ? (reg = Float32(getfield(o, 1))) : (rule = o)      # field name: gamma (<= 0.3) / lambda (0.4)
    end
    if rule isa Optimisers.Adam
        return (Int32(0), Float32(rule.eta), reg, Float32(rule.beta[1]), Float32(rule.beta[2]), Float32(rule.epsilon), 0.9f0)
    elseif rule isa Optimisers.Nesterov
        return (Int32(1), Float32(rule.eta), reg, 0.9f0, 0.999f0, 1f-8, Float32(rule.rho))
    end
    error("optimizer $opt: the engine implements WeightDecay + Adam and WeightDecay + Nesterov")
end

"pair list (0-based, flat a0,b0,a1,b1,...) of a featurizer struct (src/utils/features.jl:22-35); `nothing` for all-pairs / identity"
feat_pairs(f::FeaturesPairs) = Int32.(reduce(vcat, ([a - 1, b - 1] for (a, b) in f.pairs)))
feat_pairs(f::FeaturesAtoms) = (ix = f.atominds; Int32.(reduce(vcat, ([ix[c[1]] - 1, ix[c[2]] - 1] for c in halfinds(length(ix))))))
feat_pairs(::Any) = nothing

"""
    make_config(data, model, opt, target, minibatch; kwargs...) -> (IskConfig, keepalive)

`model` is the Flux.Chain `densenet` / `pairnet` build (src/models.jl:65-92): `[LayerNorm,] Dense..., Dense`.
"""
function make_config(data, model::Flux.Chain, opt, target, minibatch;
                     featurize_every_iteration=true, device=0, seed=0, forward_mode=3, force_layered=false)
    layers = collect(model.layers)
    ln = first(layers) isa Flux.LayerNorm
    dense = ln ? layers[2:end] : layers
    all(l -> l isa Flux.Dense, dense) || error("HipModel covers Chain([LayerNorm,] Dense...)")
    length(dense) <= MAXL || error("at most $MAXL Dense layers")
    w = zeros(Int32, MAXL + 1)
    w[1] = size(dense[1].weight, 2)
    for (i, l) in enumerate(dense); w[i+1] = size(l.weight, 1); end
    act = length(dense) > 1 ? act_id(dense[1].σ) : Int32(1)
    all(l -> act_id(l.σ) == act, dense[1:end-1]) || error("hidden layers must share one activation")
    kind, lr, reg, b1, b2, eps, rho = opt_fields(opt)
    f = data.featurizer
    D = size(data.coords[1], 1)
    pl = feat_pairs(f)
    fkind, natoms, npairs, pptr = (f === identity || f isa FeaturesCoords) ? (Int32(0), Int32(0), Int32(0), Ptr{Int32}(C_NULL)) :
        f isa FeaturesAll ? (Int32(1), Int32(D ÷ 3), Int32(0), Ptr{Int32}(C_NULL)) :
        pl !== nothing ? (Int32(2), Int32(D ÷ 3), Int32(length(pl) ÷ 2), pointer(pl)) :
        error("featurizer $f: the engine implements identity / FeaturesCoords, FeaturesAll, FeaturesAtoms and FeaturesPairs")
    tkind, perm, whit = target isa TransformShiftscale ? (0, 1, 0) :
        target isa TransformISA ? (1, Int(target.permute), Int(target.whitening)) : (2, 1, 0)   # 2: host-side transform
    cfg = IskConfig(ABI_VERSION, device, length(dense), Tuple(w), act, act_id(dense[end].σ),
                    ln, 0, ln ? Float32(first(layers).ϵ) : 1f-5,
                    kind, lr, reg, b1, b2, eps, rho, fkind, natoms, npairs, pptr,
                    featurize_every_iteration, tkind, perm, whit, minibatch, seed, forward_mode, force_layered)
    cfg, pl          # `pl` must stay alive until isk_create has copied the pair list
end

"Flux trainables order == the engine's flat parameter vector ([LN.scale, LN.bias, W1, b1, ...], W column-major (out, in))"
flat(chain) = Float32.(reduce(vcat, vec.(Flux.trainables(chain))))

# ---------------------------------------------------------------------------------------------------------------------
# the objects that stand in `iso.model` and `iso.opt`
# ---------------------------------------------------------------------------------------------------------------------
"Stands where `iso.model` stands; callable on (F,N) and (F,K,N) Float32 arrays.  The handle never leaves this struct."
mutable struct HipModel
    h::Ptr{Cvoid}
    chain::Flux.Chain          # host copy: `iso.model.layers` in show() (src/iso.jl:261), cpu(iso), save
    rule                        # the Optimisers rule the engine was configured with (for cpu(iso))
    nout::Int
end
Base.getproperty(m::HipModel, s::Symbol) = s === :layers ? getfield(m, :chain).layers : getfield(m, s)
outputdim(m::HipModel) = m.nout                       # src/models.jl:30-31
inputdim(m::HipModel) = ISOKANN.inputdim(m.chain)     # src/models.jl:26-27
iscuda(::HipModel) = false                            # chicoords (src/iso.jl:211) then hands over host arrays

"`iso.opt`: the optimizer state lives in the engine (isk_get/set_opt_state); this is what `optimizerstring` / show() see"
struct HipOpt; rule; end
ISOKANN.optimizerstring(o::HipOpt) = o.rule

function (m::HipModel)(x::AbstractArray{<:Real})      # iso.model(x): src/isotarget.jl:18,101; src/iso.jl:185,203
    x = Float32.(x)
    F = size(x, 1); ncols = length(x) ÷ F
    out = Array{Float32}(undef, m.nout, ncols)
    check(ccall((:isk_forward, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64, Int64, Ptr{Float32}), m.h, x, F, ncols, out))
    reshape(out, m.nout, size(x)[2:end]...)
end

set_option!(m::HipModel, key::AbstractString, v::Integer) =
    check(ccall((:isk_set_option, LIB[]), Cint, (Ptr{Cvoid}, Cstring, Int64), m.h, key, v))
function get_option(m::HipModel, key::AbstractString)
    v = Ref{Int64}(0)
    check(ccall((:isk_get_option, LIB[]), Cint, (Ptr{Cvoid}, Cstring, Ref{Int64}), m.h, key, v)); v[]
end

upload_data!(m::HipModel, data) = begin                # SimulationData(xs, ys; featurizer)  src/simulation.jl:100-114
    xs, ys = Float32.(data.coords[1]), Float32.(data.coords[2])          # (D,N), (D,K,N) column-major
    D, K, N = size(ys)
    check(ccall((:isk_set_data, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Int64, Int64, Int64), m.h, xs, ys, D, K, N))
end

"""
`iso.data = new` where `new` is the resident data minus its first start points and / or plus appended ones -- what addcoords!
(src/iso.jl:238) and the cutoff window of run_kde! (src/iso.jl:288-290) produce: `(drop_first, n_add)`, else `nothing`.
The comparison is a host memcmp of the coordinates (~10 ms at villin size against ~0.3 s for re-uploading them).
"""
function data_delta(old, new)
    (oxs, oys), (nxs, nys) = old.coords, new.coords
    (size(oys)[1:2] == size(nys)[1:2] && size(oxs, 1) == size(nxs, 1)) || return nothing
    no, nn = size(oxs, 2), size(nxs, 2)
    nn >= no && view(nxs, :, 1:no) == oxs && view(nys, :, :, 1:no) == oys && return (0, nn - no)
    nn < no && nxs == view(oxs, :, no-nn+1:no) && nys == view(oys, :, :, no-nn+1:no) && return (no - nn, 0)
    nothing
end

"only the NEW columns cross PCIe; the resident start points move device-to-device (isk_append_data)"
function append_data!(m::HipModel, old, new, drop::Integer, nadd::Integer)
    D, K, _ = size(old.coords[2]); no = size(old.coords[1], 2)
    xs = Float32.(new.coords[1][:, no-drop+1:end]); ys = Float32.(new.coords[2][:, :, no-drop+1:end])
    # xs / ys are locals: `pointer(x)` alone does not root them, and isk_append_data copies host -> device synchronously (GBs at
    # villin size) -- keep them alive across the call
    GC.@preserve xs ys begin
        check(ccall((:isk_append_data, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Int64, Int64, Int64, Int64),
                    m.h, nadd > 0 ? pointer(xs) : Ptr{Float32}(C_NULL), nadd > 0 ? pointer(ys) : Ptr{Float32}(C_NULL), D, K, nadd, drop))
    end
end

"""
    HipEngine(; device=0, featurize_every_iteration=true, forward_mode=3, seed=0)

Names this engine as the destination of `gpu`: `gpu(iso, HipEngine())`.  A TWO-argument method on purpose: the reference's
own `gpu(iso::Iso)` (src/iso.jl:256, Flux.gpu -> CUDA) keeps its meaning for everybody who merely loads this extension, and so
does `Iso(data; gpu=true)` (src/iso.jl:24,39).
"""
Base.@kwdef struct HipEngine
    device::Int = 0
    featurize_every_iteration::Bool = true
    forward_mode::Int = 3
    seed::Int = 0
end

"""
    gpu(iso::Iso{<:Flux.Chain}, eng::HipEngine) -> Iso{HipModel}

The role of `gpu(iso)` (src/iso.jl:256) for this engine: the model, the optimizer state and the data move into an engine on
`eng.device`.  `data.coords` (not the cached features) is uploaded: the featurizer runs on the GPU inside every iteration.
"""
function gpu(iso::Iso{<:Flux.Chain}, eng::HipEngine)
    (; device, featurize_every_iteration, forward_mode, seed) = eng
    rule = iso.opt isa HipOpt ? iso.opt.rule : optimiser_rule(iso.opt)
    cfg, keep = make_config(iso.data, iso.model, rule, iso.target, iso.minibatch; featurize_every_iteration, device, seed, forward_mode)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve keep check(ccall((:isk_create, LIB[]), Cint, (Ref{IskConfig}, Ref{Ptr{Cvoid}}), cfg, h))
    hm = HipModel(h[], iso.model, rule, ISOKANN.outputdim(iso.model))
    finalizer(m -> (m.h == C_NULL || ccall((:isk_destroy, LIB[]), Cvoid, (Ptr{Cvoid},), m.h); m.h = C_NULL), hm)
    p = flat(iso.model)
    check(ccall((:isk_set_params, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64), hm.h, p, length(p)))
    push_opt_state!(hm, iso.opt)
    upload_data!(hm, iso.data)
    Iso(hm, HipOpt(rule), iso.data, iso.target, iso.losses, iso.loggers, iso.minibatch)
end

"""
    HipIso(data; opt=NesterovRegularized(), model=defaultmodel(data), target=nothing, minibatch=100, kwargs...)

`Iso(data; ...)` (src/iso.jl:17-43) on this engine (`Iso(data; gpu=true)` itself keeps meaning Flux.gpu).
"""
function HipIso(data; opt=ISOKANN.NesterovRegularized(), model=ISOKANN.defaultmodel(data), target=nothing, minibatch=100,
                loggers=Any[], device=0, featurize_every_iteration=true, forward_mode=3, seed=0, kwargs...)
    t = something(target, default_target(model))
    host = Iso(; model, opt=Flux.setup(opt, model), data, target=t, loggers=Vector{Any}(loggers), minibatch, kwargs...)
    host.opt = HipOpt(opt)                                  # fresh state: nothing to push
    gpu(host, HipEngine(; device, featurize_every_iteration, forward_mode, seed))
end

# ---- optimizer state <-> Optimisers state tree (Flux.setup order == trainables order) ----------------------------------
# Flux.setup(rule, model) returns a tree of Optimisers.Leaf; every leaf carries the same rule
# (walked by hand -- Tuples and NamedTuples in field order, which is the order of Flux.trainables -- rather than through
# Functors.fmapstructure: the extension then needs nothing beyond ISOKANN, Flux and Optimisers in scope)
collect_leaves!(out, x::Optimisers.Leaf) = (push!(out, x); out)
collect_leaves!(out, x::Union{Tuple,NamedTuple}) = (foreach(v -> collect_leaves!(out, v), x); out)
collect_leaves!(out, x) = out

"leaves of an Optimisers state tree in trainables order"
leaves(tree) = collect_leaves!(Optimisers.Leaf[], tree)

optimiser_rule(tree::NamedTuple) = (ls = leaves(tree); isempty(ls) ? nothing : first(ls).rule)

"engine <- state tree: OptimiserChain state per leaf is (nothing | (), rule state); Adam (mt, vt, betat), Nesterov velocity"
function push_opt_state!(m::HipModel, tree)
    tree isa HipOpt && return nothing
    ls = leaves(tree)
    isempty(ls) && return nothing
    st = [l.state[end] for l in ls]                        # WeightDecay has no state; the rule is the chain's last member
    if m.rule.opts[end] isa Optimisers.Adam
        mv = Float32.(reduce(vcat, vec.(s[1] for s in st))); vv = Float32.(reduce(vcat, vec.(s[2] for s in st)))
        bt = Float64[st[1][3][1], st[1][3][2]]
        check(ccall((:isk_set_opt_state, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Int64, Ptr{Float64}), m.h, mv, vv, length(mv), bt))
    else
        mv = Float32.(reduce(vcat, vec.(st)))
        check(ccall((:isk_set_opt_state, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Int64, Ptr{Float64}), m.h, mv, C_NULL, length(mv), C_NULL))
    end
end

"state tree <- engine (for cpu(iso) / save)"
function pull_opt_state(m::HipModel)
    n = length(flat(m.chain))
    mv, vv, bt = Vector{Float32}(undef, n), Vector{Float32}(undef, n), zeros(Float64, 2)
    check(ccall((:isk_get_opt_state, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Int64, Ptr{Float64}), m.h, mv, vv, n, bt))
    tree = Flux.setup(m.rule, m.chain)
    off = 0
    adam = m.rule.opts[end] isa Optimisers.Adam
    for l in leaves(tree)
        s = l.state[end]
        k = adam ? length(s[1]) : length(s)
        if adam
            copyto!(s[1], view(mv, off+1:off+k)); copyto!(s[2], view(vv, off+1:off+k))
            l.state = (l.state[1:end-1]..., (s[1], s[2], (Float32(bt[1]), Float32(bt[2]))))
        else
            copyto!(s, view(mv, off+1:off+k))
        end
        off += k
    end
    tree
end

"pull the parameters back into the host Flux.Chain"
function sync_to_host!(m::HipModel)
    p = Vector{Float32}(undef, length(flat(m.chain)))
    check(ccall((:isk_get_params, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64), m.h, p, length(p)))
    off = 0
    for a in Flux.trainables(m.chain); n = length(a); copyto!(a, reshape(view(p, off+1:off+n), size(a))); off += n; end
    m.chain
end

"cpu(iso) (src/iso.jl:257): an ordinary reference Iso again -- Flux.Chain, Optimisers state tree, no handle inside"
cpu(iso::Iso{<:HipModel}) = Iso(sync_to_host!(iso.model), pull_opt_state(iso.model), iso.data, iso.target, iso.losses, iso.loggers, iso.minibatch)

"save(path, iso) (src/iso.jl:405-408) goes through cpu(iso): what JLD2 serialises contains no Ptr{Cvoid}"
save(path::String, iso::Iso{<:HipModel}) = save(path, cpu(iso))
# load(path) returns the CPU Iso (src/iso.jl:417-420); `gpu(load(path), HipEngine())` puts it back on the engine.

# ---------------------------------------------------------------------------------------------------------------------
# the generic functions of the train step
# ---------------------------------------------------------------------------------------------------------------------
# run!(iso, n, epochs)  src/iso.jl:72-94.  With device-side targets the n iterations are ONE ccall when no logger is installed
# (no host code between two iterations); with loggers one ccall per outer iteration so that log! still runs where it did.
function run!(iso::Iso{<:HipModel}, n=1, epochs=1; showprogress=true)
    if !(iso.target isa TransformShiftscale || iso.target isa TransformISA)
        # host-side transform (TransformPseudoInv ..., src/isotarget.jl:145ff): the reference's own loop; model(xs) and
        # expectation(model, ys) inside isotarget(t, model, xs, ys) are HIP forwards through HipModel, training is isk_train_batch
        for _ in 1:n
            target = isotarget(iso)
            for _ in 1:epochs
                push!(iso.losses, train_batch!(iso.model, nothing, target, iso.opt, iso.minibatch))
            end
            for logger in iso.loggers; ISOKANN.log!(logger; iso); end
        end
        return iso
    end
    chunk = isempty(iso.loggers) ? n : 1
    for _ in 1:cld(n, max(chunk, 1))
        losses = zeros(Float64, chunk * epochs)
        check(ccall((:isk_run, LIB[]), Cint, (Ptr{Cvoid}, Int32, Int32, Ptr{Int32}, Ptr{Float64}), iso.model.h, chunk, epochs, C_NULL, losses))
        append!(iso.losses, losses)
        for logger in iso.loggers; ISOKANN.log!(logger; iso); end
    end
    iso
end

# isotarget(iso)  src/isotarget.jl:10-12
function isotarget(iso::Iso{<:HipModel})
    if iso.target isa TransformShiftscale || iso.target isa TransformISA
        t = Array{Float32}(undef, iso.model.nout, length(iso.data))
        check(ccall((:isk_target, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Float32}), iso.model.h, t)); return t
    end
    isotarget(iso.target, iso.model, ISOKANN.features(iso.data), ISOKANN.propfeatures(iso.data))
end

# train_batch!(model, xs, target, opt, minibatch)  src/iso.jl:179-194 (xs: the resident start points)
function train_batch!(m::HipModel, xs, target::AbstractMatrix, opt, minibatch; kw...)
    loss = Ref{Float64}(0)
    check(ccall((:isk_train_batch, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int32, Ptr{Int32}, Ref{Float64}),
                m.h, Float32.(target), minibatch, C_NULL, loss)); loss[]
end

function fetch_rows(f::Symbol, iso)              # (not `fetch`: that is Base's)
    o = Array{Float32}(undef, iso.model.nout, length(iso.data))
    check(ccall((f, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Float32}), iso.model.h, o)); o
end
chis(iso::Iso{<:HipModel}) = fetch_rows(:isk_chis, iso)             # src/iso.jl:203
koopman(iso::Iso{<:HipModel}) = fetch_rows(:isk_koopman, iso)       # src/isotarget.jl:20
"K chi before its rounding to Float32 (the K-means of the Float32 outputs summed in Float64): what the engine's TransformISA solve reads"
function koopman_f64(iso::Iso{<:HipModel})
    o = Array{Float64}(undef, iso.model.nout, length(iso.data))
    check(ccall((:isk_koopman_f64, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Float64}), iso.model.h, o)); o
end
function chicoords(iso::Iso{<:HipModel}, xs)                   # src/iso.jl:211
    x = Float32.(xs); D = size(x, 1); n = length(x) ÷ D
    o = Array{Float32}(undef, iso.model.nout, n)
    check(ccall((:isk_chicoords, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64, Int64, Ptr{Float32}), iso.model.h, x, D, n, o))
    reshape(o, iso.model.nout, size(x)[2:end]...)
end

# `Iso` is a mutable struct: `iso.minibatch = mb` (src/iso.jl:10) and `iso.data = newdata` (src/iso.jl:238,289: adaptive
# sampling appends start points and cuts a window off the front) are legal in the reference; forward both to the engine
# FIRST, the field follows on success.  Data that only grew at the end / lost start points at the front keep their resident
# part on the device (each run_kde! generation would otherwise re-upload the whole set: 1.5 GB at villin size)
function Base.setproperty!(iso::Iso{<:HipModel}, f::Symbol, v)
    m = getfield(iso, :model)
    f === :minibatch && check(ccall((:isk_set_minibatch, LIB[]), Cint, (Ptr{Cvoid}, Int32), m.h, v))
    if f === :data
        old = getfield(iso, :data)
        d = data_delta(old, v)
        d === nothing ? upload_data!(m, v) : append_data!(m, old, v, d[1], d[2])
    end
    setfield!(iso, f, convert(fieldtype(typeof(iso), f), v))
end

# ---------------------------------------------------------------------------------------------------------------------
# all GPUs of a node from this ONE Julia process (SURVEY.md 8b): isk_dp_create_all / isk_dp_run_all
# ---------------------------------------------------------------------------------------------------------------------
"Stands where `iso.model` stands when the data are sharded over several GPUs: engines[i] holds the N-slab i"
mutable struct HipModelGroup
    hs::Vector{Ptr{Cvoid}}
    chain::Flux.Chain
    rule
    nout::Int
end
outputdim(m::HipModelGroup) = m.nout

function HipIsoAll(data, devices::AbstractVector{<:Integer}; opt=ISOKANN.NesterovRegularized(), model=ISOKANN.defaultmodel(data),
                   target=nothing, minibatch=100, loggers=Any[], featurize_every_iteration=true, forward_mode=3, seed=0)
    t = something(target, default_target(model))
    cfg, keep = make_config(data, model, opt, t, minibatch; featurize_every_iteration, seed, forward_mode)
    n = length(devices); devs = Int32.(devices); hs = fill(Ptr{Cvoid}(C_NULL), n)
    GC.@preserve keep check(ccall((:isk_dp_create_all, LIB[]), Cint, (Ref{IskConfig}, Ptr{Int32}, Int32, Ptr{Ptr{Cvoid}}), cfg, devs, n, hs))
    g = HipModelGroup(hs, model, opt, ISOKANN.outputdim(model))
    finalizer(g -> (isempty(g.hs) || ccall((:isk_dp_destroy_all, LIB[]), Cvoid, (Ptr{Ptr{Cvoid}}, Int32), g.hs, length(g.hs)); empty!(g.hs)), g)
    p = flat(model)
    check(ccall((:isk_dp_set_params_all, LIB[]), Cint, (Ptr{Ptr{Cvoid}}, Int32, Ptr{Float32}, Int64), hs, n, p, length(p)))
    xs, ys = Float32.(data.coords[1]), Float32.(data.coords[2]); D, K, N = size(ys)
    check(ccall((:isk_dp_set_data_all, LIB[]), Cint, (Ptr{Ptr{Cvoid}}, Int32, Ptr{Float32}, Ptr{Float32}, Int64, Int64, Int64), hs, n, xs, ys, D, K, N))
    Iso(g, HipOpt(opt), data, t, Float64[], Vector{Any}(loggers), minibatch)
end

function run!(iso::Iso{<:HipModelGroup}, n=1, epochs=1; showprogress=true)
    g = iso.model
    chunk = isempty(iso.loggers) ? n : 1
    for _ in 1:cld(n, max(chunk, 1))
        losses = zeros(Float64, chunk * epochs)
        check(ccall((:isk_dp_run_all, LIB[]), Cint, (Ptr{Ptr{Cvoid}}, Int32, Int32, Int32, Int32, Ptr{Ptr{Int32}}, Ptr{Float64}),
                    g.hs, length(g.hs), chunk, epochs, iso.minibatch, C_NULL, losses))
        append!(iso.losses, losses)
        for logger in iso.loggers; ISOKANN.log!(logger; iso); end
    end
    iso
end

function chis(iso::Iso{<:HipModelGroup})                      # the engines' slabs in rank order
    g = iso.model; nl = length(iso.data) ÷ length(g.hs)
    reduce(hcat, map(g.hs) do h
        o = Array{Float32}(undef, g.nout, nl)
        check(ccall((:isk_chis, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Float32}), h, o)); o
    end)
end

# One Julia process PER GPU instead (MPI.jl): rank 0 calls isk_dp_comm_id(id), MPI.Bcast!(id, 0, comm), every rank
# isk_dp_comm_init(h, id, rank, world); isk_dp_run(h, n, epochs, local_batch, nbatches, global_batch, C_NULL, losses) is then
# the multi-GPU run! (INTEGRATION.md section 4).

end # module
