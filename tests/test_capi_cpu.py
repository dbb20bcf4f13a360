"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol that
include/isokann_hip.h declares, struct layout is what the C side sees, host-side logic of
the reference-interface mirror, and the product path fails loudly without a device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.capi.load()
    hdr = open(os.path.join(ROOT, "include", "isokann_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(isk_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(pkg.capi.PROTOTYPES), declared ^ set(pkg.capi.PROTOTYPES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.isk_abi_version() == pkg.capi.ISK_ABI_VERSION


def test_config_layout_and_reference_defaults(pkg):
    lib = pkg.capi.load()
    cfg = pkg.capi.Config()
    C.memset(C.byref(cfg), 0xFF, C.sizeof(cfg))
    lib.isk_config_default(C.byref(cfg))
    assert cfg.abi_version == pkg.capi.ISK_ABI_VERSION == 2 and cfg.device == 0 and cfg.n_layers == 0
    assert cfg.activation == pkg.capi.ACT["sigmoid"] and cfg.lastactivation == pkg.capi.ACT["identity"]   # src/models.jl:65
    assert cfg.layernorm == 1 and cfg.ln_mode == 0 and cfg.ln_eps == pytest.approx(1e-5)
    assert cfg.opt_kind == pkg.capi.OPT["nesterov"]                                                       # src/iso.jl:18
    assert cfg.lr == pytest.approx(1e-3) and cfg.reg == pytest.approx(1e-4) and cfg.rho == pytest.approx(0.9)
    assert cfg.beta1 == pytest.approx(0.9) and cfg.beta2 == pytest.approx(0.999) and cfg.opt_eps == pytest.approx(1e-8)
    assert cfg.minibatch == 100 and cfg.isa_permute == 1 and cfg.isa_whitening == 0                       # src/iso.jl:10
    assert cfg.seed == 0 and not cfg.pairs and cfg.forward_mode == 0 and cfg.force_layered == 0


def test_status_strings_name_the_reference_domain_errors(pkg):
    lib = pkg.capi.load()
    assert b"chi function is constant" in lib.isk_status_string(pkg.capi.ECONSTANT_CHI)
    assert b"collapsed under training" in lib.isk_status_string(pkg.capi.ECOLLAPSED)
    assert b"simplex transformation" in lib.isk_status_string(pkg.capi.ESINGULAR)


def test_product_path_fails_loudly_without_a_device(pkg):
    if pkg.capi.device_count() > 0:
        pytest.skip("a GPU is present")
    xs = np.random.default_rng(0).normal(size=(4, 8))
    ys = np.random.default_rng(1).normal(size=(4, 2, 8))
    with pytest.raises(pkg.IskError):
        pkg.Iso(pkg.SimulationData(xs, ys))


def test_missing_library_is_an_error_not_a_fallback(pkg, monkeypatch):
    monkeypatch.setattr(pkg.capi, "_lib", None)
    monkeypatch.setattr(pkg.capi, "LIB_PATH", "/nonexistent/libisokann_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pkg.capi.load()


def test_host_mirror_shapes_and_defaults(pkg):
    assert pkg.pairnet_widths(2) == [2, 2, 1, 1]
    assert pkg.pairnet_widths(231) == [231, 38, 6, 1]
    assert pkg.pairnet_widths(1700, nout=3) == [1700, 142, 12, 3]
    assert pkg.halfinds(4) == [(0, 1), (0, 2), (1, 2), (0, 3), (1, 3), (2, 3)]
    m = pkg.pairnet(n=231)
    assert m.layernorm and m.activation == "sigmoid" and m.lastactivation == "identity"
    assert m.nparams() == 9519 and m.params.shape == (9519,)
    assert pkg.pairnet(n=1700, nout=3).nparams() == 246697
    assert not pkg.densenet([5, 3, 1]).layernorm                             # src/models.jl:87 default
    assert pkg.NesterovRegularized().kind == "nesterov" and pkg.AdamRegularized(1e-4, 1e-5).lr == 1e-4
    fa = pkg.FeaturesAtoms([4, 1, 7])
    assert fa.pairs(10) == [(4, 1), (4, 7), (1, 7)] and fa.featuredim(30) == 3
    assert pkg.FeaturesAll().featuredim(66) == 231

    xs = np.arange(12.0).reshape(6, 2)
    ys = np.arange(36.0).reshape(6, 3, 2)
    d = pkg.SimulationData(xs, ys, pkg.FeaturesAll())
    assert len(d) == 2 and d.nk == 3 and pkg.featuredim(d) == 1
    assert d.coords[0].dtype == np.float32 and d.coords[0].flags.f_contiguous and d.coords[1].flags.f_contiguous
    sub = d[1:2]
    assert len(sub) == 1 and np.array_equal(sub.coords[1][:, :, 0], ys[:, :, 1].astype(np.float32))
    with pytest.raises(ValueError):
        pkg.SimulationData(xs, ys[:, :, :1])
    with pytest.raises(ValueError):
        pkg.SimulationData(xs[:5], ys[:5], pkg.FeaturesAll())


def test_data_from_trajectory_matches_reference_slicing(pkg):
    """src/data.jl:88-100, in 0-based terms: reverse=true pairs frame t with (t-lag, t+lag)"""
    xs = np.arange(2 * 9, dtype=np.float64).reshape(2, 9)
    x, y = pkg.data_from_trajectory(xs, reverse=True, stride=2, lag=1)
    assert x.shape == (2, 4) and y.shape == (2, 2, 4)
    np.testing.assert_array_equal(x, xs[:, [1, 3, 5, 7]])
    np.testing.assert_array_equal(y[:, 0], xs[:, [0, 2, 4, 6]])
    np.testing.assert_array_equal(y[:, 1], xs[:, [2, 4, 6, 8]])
    x, y = pkg.data_from_trajectory(xs, reverse=False, lag=2)
    assert x.shape == (2, 7) and y.shape == (2, 1, 7)
    np.testing.assert_array_equal(y[:, 0], xs[:, 2:])
    xx, yy = pkg.data_from_trajectories([xs, xs[:, :5]], reverse=True)
    assert xx.shape == (2, 7 + 3) and yy.shape == (2, 2, 10)
    d = pkg.SimulationData(*pkg.data_from_trajectory(np.random.default_rng(0).normal(size=(6, 20))), pkg.FeaturesAll())
    assert len(d) == 18 and d.nk == 2


def test_config_struct_agrees_between_header_ctypes_and_julia_shim(pkg):
    """isk_config is mirrored by hand in two places -- capi.Config (ctypes) and `struct IskConfig` of julia/IsokannHIP.jl (which
    cannot be executed here: no Julia toolchain): the field NAMES and their ORDER must be the header's"""
    hdr = open(os.path.join(ROOT, "include", "isokann_hip.h")).read()
    start = hdr.index("typedef struct isk_config {") + len("typedef struct isk_config {")
    body = re.sub(r"/\*.*?\*/", "", hdr[start:hdr.index("} isk_config;")], flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl or decl.startswith("typedef"):
            continue
        for part in decl.split(","):
            m = re.search(r"([A-Za-z_][A-Za-z0-9_]*)\s*(\[[^\]]*\])?\s*$", part.strip())
            names.append(m.group(1))
    assert names[0] == "abi_version" and names[-1] == "force_layered" and len(names) == len(pkg.capi.Config._fields_)
    assert [f[0] for f in pkg.capi.Config._fields_] == names
    jl = open(os.path.join(ROOT, "julia", "IsokannHIP.jl")).read()
    sbody = jl[jl.index("struct IskConfig"):]
    sbody = re.sub(r"#.*", "", sbody[:sbody.index("\nend")])
    jl_names = re.findall(r"([A-Za-z_][A-Za-z0-9_]*)::", sbody)
    assert jl_names == names
    assert f"const ABI_VERSION = Int32({pkg.capi.ISK_ABI_VERSION})" in jl
    # every entry point the shim calls is declared in the header
    called = set(re.findall(r"\(:(isk_[a-z0-9_]+), LIB\[\]\)", jl)) | set(re.findall(r"fetch_rows\(:(isk_[a-z0-9_]+)", jl))
    assert called and called <= set(pkg.capi.PROTOTYPES), called - set(pkg.capi.PROTOTYPES)


# ---- types across the three hand-written mirrors of the C ABI: the header, capi.py (ctypes) and julia/IsokannHIP.jl (which no
# toolchain here can execute).  Every type is reduced to a kind: the width / signedness / pointee that the calling convention sees.
_C_KINDS = [
    (r"^(const\s+)?isk_engine\s*\*\s*\*$", "handle*"), (r"^(const\s+)?isk_engine\s*\*$", "handle"),
    (r"^(const\s+)?isk_config\s*\*$", "config*"),
    (r"^(const\s+)?int32_t\s*\*\s*const\s*\*$", "i32**"), (r"^(const\s+)?int32_t\s*\*$", "i32*"), (r"^(const\s+)?int64_t\s*\*$", "i64*"),
    (r"^(const\s+)?float\s*\*$", "f32*"), (r"^(const\s+)?double\s*\*$", "f64*"), (r"^(const\s+)?uint8_t\s*\*$", "u8*"),
    (r"^(const\s+)?char\s*\*$", "cstr"), (r"^(const\s+)?void\s*\*\s*\*$", "ptr*"), (r"^(const\s+)?void\s*\*$", "ptr"),
    (r"^int32_t$", "i32"), (r"^int$", "i32"), (r"^int64_t$", "i64"), (r"^uint64_t$", "u64"), (r"^float$", "f32"), (r"^double$", "f64"),
    (r"^void$", "void"),
]


def _c_kind(t):
    t = re.sub(r"\s+", " ", t.strip())
    for pat, kind in _C_KINDS:
        if re.match(pat, t):
            return kind
    raise AssertionError("unmapped C type: %r" % t)


def _header_prototypes(hdr):
    """name -> (return kind, [argument kinds]) of every function declared in the header; an array parameter is a pointer"""
    code = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    out = {}
    for m in re.finditer(r"^\s*((?:const\s+)?[A-Za-z_][A-Za-z0-9_]*(?:\s*\*)?)\s+(isk_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", code, flags=re.M | re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        kinds = []
        for a in [x.strip() for x in args.replace("\n", " ").split(",")]:
            if a in ("", "void"):
                continue
            am = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)?\s*(\[[^\]]*\])?$", a)
            base, arr = am.group(1).strip(), am.group(3)
            if not base:                                   # an unnamed parameter: the whole thing is the type
                base = a
            kinds.append(_c_kind(base + ("*" if arr else "")))
        out[name] = (_c_kind(ret), kinds)
    return out


def _ctypes_kind(t):
    if t is None:
        return "void"
    table = {C.c_int: "i32", C.c_int32: "i32", C.c_int64: "i64", C.c_uint64: "u64", C.c_float: "f32", C.c_double: "f64", C.c_char_p: "cstr",
             C.c_void_p: "ptr"}
    if t in table:
        return table[t]
    if hasattr(t, "_type_") and not isinstance(t._type_, str):                  # POINTER(x)
        inner = t._type_
        if inner is C.c_void_p:
            return "ptr*"
        if hasattr(inner, "_fields_"):
            return "config*"
        if hasattr(inner, "_type_") and not isinstance(inner._type_, str):    # POINTER(POINTER(c_int32))
            return _ctypes_kind(inner)[:-1] + "**"
        return {"i32": "i32*", "i64": "i64*", "f32": "f32*", "f64": "f64*", "u8": "u8*"}[
            {C.c_int32: "i32", C.c_int: "i32", C.c_int64: "i64", C.c_float: "f32", C.c_double: "f64", C.c_uint8: "u8"}[inner]]
    raise AssertionError("unmapped ctypes type: %r" % (t,))


_JL_KINDS = {"Ptr{Cvoid}": "ptr", "Ref{Ptr{Cvoid}}": "ptr*", "Ptr{Ptr{Cvoid}}": "ptr*", "Ref{IskConfig}": "config*", "Ptr{Float32}": "f32*",
             "Ptr{Float64}": "f64*", "Ref{Float64}": "f64*", "Ptr{Int32}": "i32*", "Ptr{Ptr{Int32}}": "i32**", "Ref{Int64}": "i64*",
             "Ptr{Int64}": "i64*", "Int64": "i64", "Int32": "i32", "Cint": "i32", "Cstring": "cstr", "Float32": "f32", "Cvoid": "void"}


def _same_kind(a, b):
    """an opaque handle is a void pointer on the binding side, and so is a typed DEVICE pointer (ctypes passes raw addresses for the
    `_dev` parameters); two TYPED pointers must agree in their pointee"""
    norm = {"handle": "void*", "handle*": "void**", "ptr": "void*", "ptr*": "void**", "config*": "config*"}
    a, b = norm.get(a, a), norm.get(b, b)
    if a == b:
        return True
    level = lambda k: len(k) - len(k.rstrip("*"))
    return (a.rstrip("*") == "void" or b.rstrip("*") == "void") and level(a) == level(b) and level(a) > 0


def test_argument_types_agree_between_header_ctypes_and_julia_shim(pkg):
    """the NAMES were checked above; this is the calling convention: field types of isk_config and the argument tuple of every
    prototype, header vs capi.PROTOTYPES vs each `ccall` of julia/IsokannHIP.jl"""
    hdr = open(os.path.join(ROOT, "include", "isokann_hip.h")).read()
    protos = _header_prototypes(hdr)
    assert set(protos) == set(pkg.capi.PROTOTYPES), set(protos) ^ set(pkg.capi.PROTOTYPES)
    for name, (ret, args) in protos.items():
        cres, cargs = pkg.capi.PROTOTYPES[name]
        assert _same_kind(ret, _ctypes_kind(cres)), (name, ret, cres)
        assert len(args) == len(cargs), (name, args, cargs)
        for i, (h, c) in enumerate(zip(args, cargs)):
            assert _same_kind(h, _ctypes_kind(c)), (name, i, h, c)
    # isk_config: C field types vs ctypes vs Julia
    start = hdr.index("typedef struct isk_config {") + len("typedef struct isk_config {")
    body = re.sub(r"/\*.*?\*/", "", hdr[start:hdr.index("} isk_config;")], flags=re.S)
    cfields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        tm = re.match(r"^((?:const\s+)?[A-Za-z_][A-Za-z0-9_]*\s*\*?)\s*(.*)$", decl)
        for part in tm.group(2).split(","):
            pm = re.match(r"^\s*(\*?)\s*([A-Za-z_][A-Za-z0-9_]*)\s*(\[[^\]]*\])?\s*$", part)
            cfields.append((pm.group(2), _c_kind(tm.group(1) + pm.group(1)) + ("[]" if pm.group(3) else "")))
    ct = []
    for fname, ftype in pkg.capi.Config._fields_:
        ct.append((fname, (_ctypes_kind(ftype._type_) + "[]") if hasattr(ftype, "_length_") else _ctypes_kind(ftype)))
    assert cfields == ct, [x for x in zip(cfields, ct) if x[0] != x[1]]
    jl = open(os.path.join(ROOT, "julia", "IsokannHIP.jl")).read()
    sbody = jl[jl.index("struct IskConfig"):]
    sbody = re.sub(r"#.*", "", sbody[:sbody.index("\nend")])
    jf = re.findall(r"([A-Za-z_][A-Za-z0-9_]*)::((?:NTuple\{[^}]*\})|(?:[A-Za-z0-9]+(?:\{[A-Za-z0-9]+\})?))", sbody)
    jl_field_kind = {"Int32": "i32", "UInt64": "u64", "Float32": "f32", "Ptr{Int32}": "i32*", "NTuple{MAXL + 1,Int32}": "i32[]"}
    assert [(n, jl_field_kind[t]) for n, t in jf] == cfields
    assert "const MAXL = %d" % pkg.capi.ISK_MAX_LAYERS in jl
    # every ccall of the shim: return type and argument tuple against the header's prototype
    calls = re.findall(r"ccall\(\((:isk_[a-z0-9_]+|f), LIB\[\]\),\s*([A-Za-z]+),\s*\(([^()]*(?:\{[^()]*\}[^()]*)*)\)", jl)
    assert len(calls) >= 25
    seen = set()
    for sym, ret, tup in calls:
        jargs = [t.strip() for t in re.split(r",\s*(?![^{]*\})", tup) if t.strip()]
        names = [sym[1:]] if sym != "f" else ["isk_chis", "isk_koopman"]           # fetch_rows(f, iso) is called with these two
        for name in names:
            hret, hargs = protos[name]
            seen.add(name)
            assert _same_kind(hret, _JL_KINDS[ret]), (name, hret, ret)
            assert len(hargs) == len(jargs), (name, hargs, jargs)
            for i, (h, j) in enumerate(zip(hargs, jargs)):
                assert _same_kind(h, _JL_KINDS[j]), (name, i, h, j)
    assert {"isk_create", "isk_set_data", "isk_append_data", "isk_run", "isk_dp_run_all", "isk_set_opt_state"} <= seen
    # the engine must not take over the reference's one-argument gpu(iso) (src/iso.jl:256): opt-in through a second argument
    assert re.search(r"^function gpu\(iso::Iso\{<:Flux\.Chain\}, eng::HipEngine\)", jl, flags=re.M)
    assert not re.search(r"^(function )?gpu\(iso::Iso(\{[^}]*\})?\s*[;)]", jl, flags=re.M)


def test_options_are_per_engine_and_validated(pkg):
    lib = pkg.capi.load()
    assert lib.isk_set_option(None, b"train_graph", 0) == pkg.capi.EINVAL
    v = C.c_int64(0)
    assert lib.isk_get_option(None, b"train_graph", C.byref(v)) == pkg.capi.EINVAL
    # the documented keys of the header are the keys of the option table (isk_get_option on a live engine is a GPU test)
    hdr = open(os.path.join(ROOT, "include", "isokann_hip.h")).read()
    doc = hdr[hdr.index("/* ---- options"):hdr.index("int  isk_set_option")]
    keys = set(re.findall(r"^ \*   ([a-z0-9_]+) ", doc, flags=re.M)) - {"key"}
    src = open(os.path.join(ROOT, "isokann.jl_amd", "csrc", "engine_options.hip")).read()
    shipped, experimental = src.split("#ifdef ISK_EXPERIMENTAL", 1)
    table = set(re.findall(r'\{"([a-z0-9_]+)", &IskOptions::', shipped))
    assert table == keys, (table ^ keys)
    assert len(table) <= 20                                          # the shipped option table stays small
    # knobs of rejected variants / diagnostics exist in -DISK_EXPERIMENTAL builds only, and the header names them
    extra = set(re.findall(r'\{"([a-z0-9_]+)", &IskOptions::', experimental.split("#endif", 1)[0]))
    assert extra and not (extra & table) and all(k in doc for k in extra), extra


def test_no_vendor_blas_is_reachable_from_the_library(pkg):
    """the rocBLAS route of the wide-network training path is gone -- no rocblas / hipblas symbol or
    DT_NEEDED entry, no dlopen of one"""
    import subprocess
    lib = os.path.join(ROOT, "isokann.jl_amd", "lib", "libisokann_hip.so")
    pkg.capi.load()
    dyn = subprocess.run(["readelf", "-d", lib], capture_output=True, text=True).stdout.lower()
    assert "blas" not in dyn, dyn
    data = open(lib, "rb").read().lower()
    assert b"rocblas" not in data and b"hipblas" not in data
    for f in os.listdir(os.path.join(ROOT, "isokann.jl_amd", "csrc")):
        txt = open(os.path.join(ROOT, "isokann.jl_amd", "csrc", f)).read().lower()
        assert "rocblas" not in txt and "hipblas" not in txt, f


def test_product_pseudoinv_host_algebra_has_analytic_answers(pkg):
    """TransformPseudoInv's d x d algebra runs on the host in the product too (api._pseudoinv_target, as in the reference's `|> cpu`,
    src/isotarget.jl:152-179).  Checked against known answers, not against its twin in the oracle: if chi = M * kchi for an
    invertible M, then chi * pinv(kchi) * kchi == chi (no eigenvectors / normalisation / permutation); the Schur vectors are
    orthogonal (the transform only rotates that target); normalize makes every row's L1 norm N; permute returns rows of the
    un-permuted result"""
    api = pkg.api if hasattr(pkg, "api") else __import__("isokann_jl_amd.api", fromlist=["api"])
    rng = np.random.default_rng(3)
    d, n = 3, 200
    kchi = rng.uniform(size=(d, n)).astype(np.float32)
    M = np.array([[0.9, 0.1, 0.0], [0.05, 0.8, 0.15], [0.0, 0.2, 0.8]])
    chi = (M @ kchi.astype(np.float64)).astype(np.float32)
    plain = api.TransformPseudoInv(normalize=False, direct=True, eigenvecs=False, permute=False)
    t0 = api._pseudoinv_target(plain, chi, kchi)
    np.testing.assert_allclose(t0, chi, atol=2e-6)
    t1 = api._pseudoinv_target(api.TransformPseudoInv(normalize=False, direct=True, eigenvecs=True, permute=False), chi, kchi)
    # t1 = Z * t0 with Z orthogonal: the Gram matrix of the columns is unchanged
    np.testing.assert_allclose(t1.T.astype(np.float64) @ t1, t0.T.astype(np.float64) @ t0, atol=1e-4)
    t2 = api._pseudoinv_target(api.TransformPseudoInv(normalize=True, direct=True, eigenvecs=False, permute=False), chi, kchi)
    np.testing.assert_allclose(np.abs(t2).sum(axis=1), n, rtol=1e-5)
    t3 = api._pseudoinv_target(api.TransformPseudoInv(normalize=True, direct=True, eigenvecs=True, permute=True), chi, kchi)
    t3u = api._pseudoinv_target(api.TransformPseudoInv(normalize=True, direct=True, eigenvecs=True, permute=False), chi, kchi)
    assert sorted(map(tuple, np.round(t3, 5))) == sorted(map(tuple, np.round(t3u, 5)))
    # fixperm picks the row order closest to chi in the entrywise 1-norm: no other order of the same rows is closer
    import itertools
    best = min(np.abs(t3u[list(p)] - chi).sum() for p in itertools.permutations(range(d)))
    assert np.abs(t3 - chi).sum() <= best * (1 + 1e-6)
    with pytest.raises(ValueError):
        api._pseudoinv_target(plain, chi[:1], kchi[:1])


@pytest.mark.parametrize("d", [9, 12, 20])
def test_isa_solve_beyond_eight_outputs_uses_an_assignment_solve(pkg, d):
    """isk_isa_solve is host code (no GPU): TransformISA for d > 8 (src/isotarget.jl:81-127).  fixperm there is the linear
    assignment problem the reference's TODO names (src/isotarget.jl:121) instead of d! permutations; checked against (1) the
    oracle, whose solver is scipy's, (2) the planted answer: chi = a known row permutation of the ISA target, (3) at d = 9 the
    reference's own brute force over all 362 880 permutations"""
    from oracle import isokann_oracle as O
    lib = pkg.capi.load()
    rng = np.random.default_rng(40 + d)
    N = 600
    # membership-like K chi: a noisy convex mixture of d vertices, so the simplex is well conditioned
    V = rng.uniform(0.2, 1.0, size=(d, d)) + 2.0 * np.eye(d)
    Wt = rng.dirichlet(np.ones(d) * 0.3, size=N)
    ks = (Wt @ V).T.astype(np.float32)                                     # (d, N)
    A = O.myisa(ks.T.astype(np.float64))
    t = A.T @ ks.astype(np.float64)
    planted = rng.permutation(d)
    chi = (t[planted] + 0.01 * rng.normal(size=t.shape)).astype(np.float32)     # old = the target in another row order, plus noise
    At = np.zeros(d * d, np.float64)
    rp = np.zeros(d, np.int32)
    pkg.capi.check(lib.isk_isa_solve(pkg.capi.ptr_f32(np.asfortranarray(ks)), pkg.capi.ptr_f32(np.asfortranarray(chi)), N, d, 1, 0,
                                     pkg.capi.ptr_f64(At), pkg.capi.ptr_i32(rp)))
    np.testing.assert_allclose(At.reshape(d, d), A.T, rtol=1e-9, atol=1e-12)             # target = At * ks
    assert list(rp) == list(planted)
    assert tuple(rp) == O.fixperm_index(t, chi.astype(np.float64))
    if d == 9:
        import itertools
        C = np.abs(t[:, None, :] - chi.astype(np.float64)[None, :, :]).sum(axis=2)       # C[a, i]
        best = min(itertools.permutations(range(d)), key=lambda p: sum(C[p[i], i] for i in range(d)))
        assert tuple(rp) == best
    assert lib.isk_isa_solve(pkg.capi.ptr_f32(np.asfortranarray(ks)), pkg.capi.ptr_f32(np.asfortranarray(chi)), N, 65, 1, 0,
                             pkg.capi.ptr_f64(np.zeros(65 * 65)), pkg.capi.ptr_i32(np.zeros(65, np.int32))) == pkg.capi.EINVAL
