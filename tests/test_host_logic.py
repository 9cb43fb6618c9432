"""CPU tests of the host side: the C-ABI library loads and exports every declared symbol,
host-only helpers (partition, CSR split, tridiagonal solver), input parsing, loud failure without a GPU."""
import os
import re

import numpy as np
import pytest

import oracle
from helpers import chain
import lanczosplusplus_amd as lp
from lanczosplusplus_amd import _capi, geometry

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "lpp_engine.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(lpp_[a-z0-9_]+)\s*\(", header)) - {"lpp_status"}
    assert declared, "no declarations parsed"
    L = _capi.lib()
    for name in sorted(declared):
        assert hasattr(L, name), name
    # the Python binding covers exactly the declared surface
    assert declared == set(_capi.SYMBOLS), declared ^ set(_capi.SYMBOLS)
    assert L.lpp_abi_version() == _capi.LPP_ABI_VERSION


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(lp.LppError) as ei:
        lp.LanczosEngine()
    assert "no CPU fallback" in str(ei.value)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "lanczosplusplus_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in txt and "lpp_oracle" not in txt and "lppo_" not in txt, os.path.join(dirpath, f)


def test_partition_rows():
    s = lp.partition_rows(100, 3, 10)
    assert list(s) == [0, 40, 80, 100]
    s = lp.partition_rows(12870 * 12870, 8, 12870)
    assert s[0] == 0 and s[-1] == 12870 * 12870 and all(x % 12870 == 0 for x in s)
    assert max(np.diff(s)) == -(-12870 // 8) * 12870
    s = lp.partition_rows(30, 8, 10)  # fewer blocks than ranks: trailing ranks are empty
    assert list(s) == [0, 10, 20, 30, 30, 30, 30, 30, 30]
    with pytest.raises(lp.LppError):
        lp.partition_rows(101, 2, 10)


@pytest.mark.parametrize("cplx", [False, True])
def test_split_csr_reassembles_the_product(cplx):
    L = 6
    if cplx:
        A = oracle.tj_csr(L, 2, 2, chain(L, -1.0), chain(L, 0.4), chain(L, 0.4), chain(L, -0.1), force_complex=True)
        block = 1
    else:
        A = oracle.hubbard_csr(L, 3, 3, chain(L, -1.0, True), np.full(L, 4.0))
        block = 20  # N_up
    P = 3
    starts = lp.partition_rows(A.nrows, P, block)
    stride = int(max(np.diff(starts)))
    y = oracle.fill_random(A.nrows, 11, cplx)
    full = oracle.spmv_acc(A, np.zeros_like(y), y)
    gath = np.zeros(P * stride, y.dtype)
    for r in range(P):
        gath[r * stride:r * stride + (starts[r + 1] - starts[r])] = y[starts[r]:starts[r + 1]]
    nnz_total = 0
    for r in range(P):
        lo, hi = starts[r], starts[r + 1]
        rp = A.rowptr[lo:hi + 1] - A.rowptr[lo]
        ci = A.colind[A.rowptr[lo]:A.rowptr[hi]]
        va = A.values[A.rowptr[lo]:A.rowptr[hi]]
        (rpl, cl, vl), (rpr, cr, vr) = lp.split_csr(r, P, starts, stride, rp, ci, va)
        nnz_total += len(cl) + len(cr)
        assert cl.size == 0 or (cl.min() >= 0 and cl.max() < hi - lo)
        loc = oracle.Csr(rpl, cl, vl)
        rem = oracle.Csr(rpr, cr, vr)
        out = np.zeros(hi - lo, y.dtype)
        # local part reads the rank's own slice, remote part the padded gathered vector
        ys = np.ascontiguousarray(y[lo:hi])
        if loc.nnz:
            lib = oracle.lib()
            import ctypes as C
            lib.lppo_spmv_acc(loc.nrows, loc.rowptr, loc.colind, loc.values.ctypes.data_as(C.c_void_p), int(cplx),
                              out.ctypes.data_as(C.c_void_p), ys.ctypes.data_as(C.c_void_p), 1)
        if rem.nnz:
            lib = oracle.lib()
            import ctypes as C
            lib.lppo_spmv_acc(rem.nrows, rem.rowptr, rem.colind, rem.values.ctypes.data_as(C.c_void_p), int(cplx),
                              out.ctypes.data_as(C.c_void_p), gath.ctypes.data_as(C.c_void_p), 1)
        assert np.abs(out - full[lo:hi]).max() < 1e-13
        if not cplx:
            # Hubbard partition at multiples of N_up: the diagonal and every up-hop are rank-local
            up_and_diag = sum(1 for i in range(lo, hi) for c in A.colind[A.rowptr[i]:A.rowptr[i + 1]] if c // block == i // block)
            assert len(cl) >= up_and_diag
    assert nnz_total == A.nnz


def test_tridiag_lowest_vs_numpy():
    rng = np.random.default_rng(9)
    for n in (1, 2, 5, 40, 200):
        d, e = rng.normal(size=n), np.abs(rng.normal(size=max(n - 1, 0))) + 0.01
        T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
        ref = np.linalg.eigvalsh(T)
        k = min(3, n)
        w = lp.tridiag_lowest(d, e, k)
        assert np.abs(w - ref[:k]).max() < 1e-12 * max(1.0, np.abs(ref).max())
        w2, z = lp.tridiag_lowest(d, e, k, vectors=True)
        assert np.abs(w2 - ref[:k]).max() < 1e-12 * max(1.0, np.abs(ref).max())
        assert np.abs(T @ z - z * w2).max() < 1e-10
        # the engine's solver and the oracle's are independent implementations: they must agree
        assert np.abs(w - oracle.tridiag_eig(d, e)[:k]).max() < 1e-12 * max(1.0, np.abs(ref).max())


def test_input0_parsing_and_geometry():
    """tests/golden/input0.inp is the reference's TestSuite/inputs/input0.inp (a data file)."""
    inp = geometry.parse_input(open(os.path.join(ROOT, "tests", "golden", "input0.inp")).read())
    assert inp["Model"] == "HubbardOneBand" and int(inp["TotalNumberOfSites"]) == 4
    assert int(inp["TargetElectronsUp"]) == 2 and int(inp["TargetElectronsDown"]) == 2
    assert len(inp["hubbardU"]) == 4 and len(inp["potentialV"]) == 8
    (hop,) = geometry.terms_from_input(inp)
    assert np.array_equal(hop, chain(4, -1.0))
    A = oracle.hubbard_csr(4, 2, 2, hop, inp["hubbardU"], inp["potentialV"])
    assert abs(np.linalg.eigvalsh(A.to_scipy().toarray())[0] + 2 * np.sqrt(5)) < 1e-12
    lad = geometry.ladder(8, 2, -1.0, -0.5)
    assert lad[0, 2] == -1.0 and lad[0, 1] == -0.5 and lad[1, 2] == 0 and np.array_equal(lad, lad.T)


def test_product_template_is_bank_conflict_free_and_lossless():
    """lpp_pb_pack_template (host part of the product-basis layout): the per-slice, per-value streams of 16-bit window indices
    reproduce the in-block matrix entry for entry, the filling only reads zero slots, and within every slot the 32 lanes of a
    half-wave ask no LDS bank (bank = element index mod 32 for ds_read_b64) for more than `bank_ways` different addresses."""
    import ctypes as C
    import oracle
    import scipy.sparse as sp
    from helpers import chain, square
    from lanczosplusplus_amd import _capi
    L = _capi.lib()

    def vp(a):
        return a.ctypes.data_as(C.c_void_p)

    def pack(A, pitch, ways):
        n = A.nrows
        ng, sl, nw, ent, slots = C.c_int32(), C.c_int32(), C.c_int64(), C.c_int64(), C.c_int64()
        gv = np.zeros(8)
        args = (n, pitch, vp(A.rowptr), vp(A.colind), vp(A.values), C.byref(ng), vp(gv), C.byref(sl), C.byref(nw))
        rc = L.lpp_pb_pack_template(*args, None, None, None, C.byref(ent), C.byref(slots), ways)
        if rc != 0:
            return rc, None
        G, spb = ng.value, sl.value
        off, ln, words = np.zeros(spb * G, np.int32), np.zeros(spb * G, np.uint16), np.zeros(nw.value, np.uint32)
        _capi.check(L.lpp_pb_pack_template(*args, vp(off), vp(ln), vp(words), C.byref(ent), C.byref(slots), ways))
        return 0, (G, gv, spb, off, ln, words, ent.value, slots.value)

    cases = [(12, 6, chain(12, -1.0, True)), (10, 5, square(2, 5, -0.7)), (9, 3, chain(9, 1.3, True) + np.diag([0.5] * 8, 1) + np.diag([0.5] * 8, -1))]
    for (nsites, nup, hop) in cases:
        A = oracle.hubbard_csr(nsites, nup, 0, hop, np.zeros(nsites), np.linspace(-1, 1, nsites))  # one species: T plus a diagonal to be skipped
        n = A.nrows
        pitch = (n + 15) // 16 * 16
        R = A.to_scipy().tolil()
        R.setdiag(0)
        R = R.tocsr()
        R.eliminate_zeros()
        nslots = {}
        for ways in (1, 2):
            rc, (G, gv, spb, off, ln, words, ent, slots) = pack(A, pitch, ways)
            assert rc == 0 and spb == (n + 63) // 64 and G == len(np.unique(R.data))
            nslots[ways] = slots
            rows, cols, vals = [], [], []
            for j in range(spb):
                for g in range(G):
                    w = words[off[j * G + g] * 128:(off[j * G + g] + int(ln[j * G + g])) * 128].reshape(-1, 64, 2)  # [chunk][lane][word]
                    idx = np.stack([w[:, :, 0] & 0xffff, w[:, :, 0] >> 16, w[:, :, 1] & 0xffff, w[:, :, 1] >> 16], axis=2)  # [chunk][lane][k]
                    for chunk in idx:
                        for k in range(4):
                            slot = chunk[:, k]
                            for h in (0, 1):
                                addr = np.unique(slot[h * 32:(h + 1) * 32])
                                assert np.bincount(addr & 31, minlength=32).max() <= ways
                            real = slot < n
                            assert np.all((slot[~real] >= pitch) & (slot[~real] < pitch + 32))
                            lanes = np.nonzero(real)[0]
                            assert np.all(j * 64 + lanes < n)
                            rows += list(j * 64 + lanes)
                            cols += list(slot[real])
                            vals += [gv[g]] * len(lanes)
            M = sp.csr_matrix((vals, (rows, cols)), shape=(n, n))
            assert len(vals) == ent == R.nnz and abs(M - R).max() == 0
        assert nslots[2] <= nslots[1]
    # more than 8 distinct in-block values: refused (the engine then keeps the general layout)
    rng = np.random.default_rng(0)
    hop = chain(10, -1.0) * (1 + 0.1 * rng.standard_normal((10, 10)))
    hop = (hop + hop.T) / 2
    A = oracle.hubbard_csr(10, 5, 0, hop, np.zeros(10))
    rc, _ = pack(A, 256, 2)
    assert rc == _capi.LPP_ERR_INVALID


def test_rccl_communicator_library_exports_its_symbols():
    """liblpp_comm_rccl.so (include/lpp_comm_rccl.h): loads next to librccl and exports every declared entry point (no GPU call)."""
    import ctypes as C
    import re
    path = os.path.join(ROOT, "lanczosplusplus_amd", "csrc", "liblpp_comm_rccl.so")
    assert os.path.exists(path), "run __graft_entry__.build()"
    lib = C.CDLL(path)
    hdr = open(os.path.join(ROOT, "include", "lpp_comm_rccl.h")).read()
    names = set(re.findall(r"\b(lpp_rccl_\w+)\s*\(", hdr))
    assert {"lpp_rccl_unique_id", "lpp_rccl_comm_create", "lpp_rccl_comm_get", "lpp_rccl_comm_destroy", "lpp_rccl_comm_selftest",
            "lpp_rccl_last_error"} <= names
    for n in names:
        assert hasattr(lib, n), n


def test_exchange_chunk_helper():
    """lpp_xchg_chunk (include/lpp_engine.h): ceil(N_down/P) down configurations x (ceil(N_up/P) rounded up to 16) up indices"""
    from lanczosplusplus_amd._capi import lib
    L = lib()
    assert L.lpp_xchg_chunk(12870, 12870, 8) == 1609 * 1616
    assert L.lpp_xchg_chunk(924, 495, 4) == 124 * 240
    assert L.lpp_xchg_chunk(252, 210, 4) == 53 * 64
    assert L.lpp_xchg_chunk(16, 16, 1) == 16 * 16
    assert L.lpp_xchg_chunk(0, 5, 2) == 0 and L.lpp_xchg_chunk(5, 5, 0) == 0
    for n_up, n_dn, P in ((12870, 12870, 2), (77520, 38760, 8), (15, 6, 2)):
        c = L.lpp_xchg_chunk(n_up, n_dn, P)
        per = -(-n_dn // P)
        assert c % per == 0 and (c // per) % 16 == 0 and (c // per) * P >= n_up and (c // per - 16) * P < n_up + 16 * P


def test_bench_workloads_are_well_formed():
    """bench.py's workload table: Hermitian hopping matrices, fillings within the lattice, and every golden fixture it names exists with the
    fields the coefficient gate and the e0 check read."""
    import importlib.util
    import json

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)  # imports numpy only at module level; torch and the engine are imported inside main()
    for name, (model, p) in bench.WORKLOADS.items():
        if model == "hubbard":
            hop = np.asarray(p["hop"]())
            assert hop.shape == (p["L"], p["L"]) and np.allclose(hop, hop.conj().T), name
            assert 0 < p["nup"] <= p["L"] and 0 < p["ndown"] <= p["L"], name
            assert ("complex" in name) == np.iscomplexobj(hop), name
        elif model == "tj":
            assert p["lx"] * p["ly"] == p["L"] and p["nup"] + p["ndown"] <= p["L"], name
        elif model == "heisenberg":
            assert 0 <= p["sz"] <= p["L"], name
    assert bench.WORKLOADS["hubbard_4x4_half_filling_pbc_U4"][1]["nup"] == 8  # BASELINE config 2, the default line
    for name, fixture in bench.GOLDEN.items():
        assert name in bench.WORKLOADS
        g = json.load(open(os.path.join(root, "tests", "golden", fixture)))
        assert len(g["a"]) == len(g["b"]) >= 30 and "e0" in g, fixture


def test_segmented_in_block_form_reproduces_the_hopping_matrix():
    """lpp_pb_seg_plan_stats (host part of k_pb_up_seg, csrc/lpp_pbseg.h): the in-block matrix of one species -- the oracle's restatement of
    HubbardHelper.h:191-243 in the basis of BasisOneSpin.h:53-61 -- is read as (L sites, n particles, hopping amplitudes), cut into
    segments by the high sites of the basis word and packed as low-low lists per item type, cross tables per class and high-high pairs
    per segment; the library expands that description again and compares it with the matrix entry by entry (out[0] == 1).  BASELINE
    config 5's species (4x5 lattice, 7 and 6 particles) at full size; matrices that are not of that form are refused, not mangled."""
    import ctypes as C
    import oracle
    from helpers import chain, square
    from math import comb
    L = _capi.lib()

    def vp(a):
        return a.ctypes.data_as(C.c_void_p)

    def plan(A, wcap):
        out = (C.c_int64 * 16)()
        perm = np.full(A.nrows, -1, np.int32)
        _capi.check(L.lpp_pb_seg_plan_stats(A.nrows, vp(A.rowptr), vp(A.colind), vp(A.values), wcap, out, vp(perm)))
        return list(out), perm

    def species(nsites, n, hop):
        return oracle.hubbard_csr(nsites, n, 0, hop, np.zeros(nsites), np.linspace(-1, 1, nsites))  # T plus a diagonal that is skipped

    # (sites, particles, hoppings, wcap) -> (high sites, segments, items)
    cases = [((12, 6, chain(12, -1.0, True), 256), (2, 4, 4)), ((12, 5, square(3, 4, -1.0), 320), (2, 4, 4)), ((16, 8, square(4, 4, -1.0), 4096), (2, 4, 4)),
             ((20, 7, square(4, 5, -1.0), 8128), (5, 32, 14)), ((20, 6, square(4, 5, -1.0), 8128), (4, 16, 7))]
    for (nsites, n, hop, wcap), (s, nseg, nitems) in cases:
        A = species(nsites, n, hop)
        o, perm = plan(A, wcap)
        assert o[0] == 1 and (o[1], o[2]) == (nsites, n), o
        assert (o[3], o[4], o[5]) == (s, nseg, nitems) and o[7] <= wcap, o
        assert sorted(perm.tolist()) == list(range(A.nrows))  # a permutation of the positions
        # segments by length: the first one is the longest class, in the basis order inside
        longest = max(comb(nsites - s, n - p) for p in range(min(s, n) + 1) if n - p <= nsites - s)
        assert np.all(np.diff(perm[:longest]) == 1)
        if nsites == 20:  # everything that is shared by class stays far below one XCD's 4 MiB L2; what is per segment is a few KB
            assert o[8] < 2 << 20 and o[9] < 16 << 10, o
    # not the hopping matrix of one species: refused
    A = species(12, 6, chain(12, -1.0, True))
    k = A.rowptr[400] + (0 if A.colind[A.rowptr[400]] != 400 else 1)
    B = oracle.Csr(A.rowptr.copy(), A.colind.copy(), A.values.copy())
    B.values[k] *= 1.5  # one entry of one row differs from its bond's amplitude
    assert plan(B, 256)[0][0] == 0
    B = oracle.Csr(A.rowptr.copy(), A.colind.copy(), A.values.copy())
    B.colind[k] = (B.colind[k] + 7) % A.nrows if abs(int(B.colind[k]) - 400) > 8 else B.colind[k] + 9  # an entry that is not one hop
    assert plan(B, 256)[0][0] == 0
    # more than two hopping magnitudes among the low sites: the kernel's two value groups do not hold them
    tt = chain(12, -1.0, False) + 0.5 * (np.diag(np.ones(10), 2) + np.diag(np.ones(10), -2))
    assert plan(species(12, 5, tt), 256)[0][0] == 0
    # Heisenberg.h:278-307 for S = 1/2: S+S- moves an up spin without a sign.  On an open chain nothing sits between neighbours, so the
    # off-diagonal part IS a hopping matrix (hard-core bosons = fermions there) and is taken; on a ladder the leg bonds skip a site,
    # the fermion sign the form implies is not in the matrix, and it is refused
    H = oracle.heis_csr(12, 1, 6, chain(12, 1.0), chain(12, 1.0))
    o, _ = plan(H, 256)
    assert o[0] == 1 and (o[1], o[2]) == (12, 6)
    H = oracle.heis_csr(12, 1, 6, square(2, 6, 1.0, False), square(2, 6, 1.0, False))
    assert plan(H, 256)[0][0] == 0


def test_tj_hole_major_plan_reproduces_the_oracle_matrix():
    """lpp_tj_plan_stats (host part of k_tj_apply, csrc/lpp_tj.h, lpp_tj_host.cpp): the one-orbital t-J model planned in its hole-major form --
    hole configurations x spin patterns of the occupied sites, per configuration the bonds whose S+S- flips an antiparallel pair (value
    0.5 J (-1)^(q-p), TjMultiOrb.h:697-783) and the moves of an electron onto a neighbouring hole (a rotation of the pattern's bits between the
    two sites, sign of the same-species electrons between, :649-695).  The library expands the plan again exactly as the kernel walks it and
    compares it with the oracle's restatement of TjMultiOrb::setupHamiltonian in the REFERENCE's basis order (sorted (down << L) | up words):
    every off-diagonal entry, column and value bits.  A CSR of another model is refused."""
    import ctypes as C
    import oracle
    from helpers import chain, square
    from math import comb
    Lib = _capi.lib()

    def vp(a):
        return None if a is None else a.ctypes.data_as(C.c_void_p)

    def plan(L, nup, ndown, hop, jpm, A):
        hop = np.asarray(hop)
        hr = np.ascontiguousarray(hop.real, np.float64)
        hi = np.ascontiguousarray(hop.imag, np.float64) if np.iscomplexobj(hop) else None
        jp = np.ascontiguousarray(jpm, np.float64)
        out = (C.c_int64 * 10)()
        if A is None:
            _capi.check(Lib.lpp_tj_plan_stats(L, nup, ndown, vp(hr), vp(hi), vp(jp), 0, None, None, None, 0, out))
        else:
            _capi.check(Lib.lpp_tj_plan_stats(L, nup, ndown, vp(hr), vp(hi), vp(jp), A.nrows, vp(A.rowptr), vp(A.colind), vp(A.values), int(A.is_complex), out))
        return list(out)

    torus_j = square(3, 4, 0.4, pbc=True) * (1 + 0.25 * np.triu(np.ones((12, 12)), 3) + 0.25 * np.tril(np.ones((12, 12)), -3))
    cases = [  # (L, nup, ndown, hop, jpm, jzz, w, potentialV, complex engine)
        (12, 5, 5, chain(12, -1.0), chain(12, 0.4), chain(12, 0.4), chain(12, -0.1), None, True),  # two holes, open chain
        (12, 5, 4, square(3, 4, -1.0, pbc=True), torus_j, square(3, 4, 0.3, pbc=True), square(3, 4, -0.1, pbc=True), np.linspace(-0.2, 0.3, 24), False),  # torus, three holes, unequal bonds
        (10, 5, 4, chain(10, -1.0, True) * np.exp(0.3j), chain(10, 0.5, True), chain(10, 0.5, True), chain(10, -0.125, True), None, True),  # complex hoppings, one hole
        (9, 3, 4, square(3, 3, -1.0, pbc=True), square(3, 3, 0.4, pbc=True), square(3, 3, 0.3, pbc=True), square(3, 3, -0.1, pbc=True), None, False),  # odd lattice, two holes
        (12, 6, 6, chain(12, -1.0, True), chain(12, 0.4, True), chain(12, 0.4, True), chain(12, -0.1, True), None, False),  # no holes: one block
        (12, 4, 4, square(3, 4, -1.0, pbc=True), square(3, 4, 0.4, pbc=True), square(3, 4, 0.4, pbc=True), square(3, 4, -0.1, pbc=True), None, True),  # four holes
    ]
    for L, nup, ndown, hop, jpm, jzz, w, pv, cplx in cases:
        A = oracle.tj_csr(L, nup, ndown, hop, jpm, jzz, w, pv, force_complex=cplx)
        o = plan(L, nup, ndown, hop, jpm, A)
        nholes = L - nup - ndown
        assert o[0] == 1, (L, nup, ndown, o)
        assert (o[1], o[2]) == (comb(L, nholes), comb(nup + ndown, nup)) and o[1] * o[2] == A.nrows
        assert o[3] == min(nup + ndown, 12) or o[3] < nup + ndown  # low positions of a segment: as many as fit a window of 1024 patterns
        assert o[7] <= o[5] and o[8] <= 96 and o[9] <= 64
        # off-diagonal entries of the CSR = flips of the antiparallel settings of every bond + one entry per move and pattern
        assert A.nnz - A.nrows == o[5] * 2 * comb(nup + ndown - 2, nup - 1) + o[6] * o[2]
    # BASELINE config 4 (4x5 torus, 9 up 9 down): 190 configurations x 48620 patterns in 61 items, segments of the low 12 positions; statistics only
    o = plan(20, 9, 9, square(5, 4, -1.0, pbc=True), square(5, 4, 0.4, pbc=True), None)
    assert o[:7] == [1, 190, 48620, 12, 61, 6120, 1440] and (o[8], o[9]) == (33, 8), o
    # not this model: another J, a changed entry
    L, nup, ndown, hop, jpm, jzz, w, pv, cplx = cases[0]
    A = oracle.tj_csr(L, nup, ndown, hop, jpm, jzz, w, pv, force_complex=cplx)
    assert plan(L, nup, ndown, hop, 2 * jpm, A)[0] == 0
    vals = A.values.copy()
    k = int(A.rowptr[100]) + (0 if A.colind[A.rowptr[100]] != 100 else 1)
    vals[k] = -vals[k]
    assert plan(L, nup, ndown, hop, jpm, oracle.Csr(A.rowptr, A.colind, vals))[0] == 0
