"""CPU tests that pin the oracle (oracle/lpp_oracle.c).

The reference tree holds no expected outputs for this path ("parity unpinned"), so the
pins are: (1) golden vectors produced by the one reference translation unit that compiles
stand-alone (tests/golden/heis_inf_temp.json, generator committed next to it), (2) the structural
formulas that ARE in the reference tree (basis order, perfectIndex, literal linear-scan index ==
fast index), (3) closed-form energies (input0.inp -> -2 sqrt 5, two-site Hubbard, 4-site Heisenberg
ring) and (4) an independent dense ED built from Jordan-Wigner / Pauli operators with numpy only.
"""
import itertools
import json
import os
from math import comb

import numpy as np
import pytest

import oracle
from helpers import chain, rel, square

HERE = os.path.dirname(os.path.abspath(__file__))


# ------------------------------------------------------------------ structural pins
def test_comb_table_and_onespin_order():
    for L, n in [(4, 2), (6, 3), (8, 4), (10, 0), (10, 10), (12, 5), (16, 8)]:
        b = oracle.onespin_basis(L, n)
        assert len(b) == comb(L, n)
        # ascending integer order, fixed popcount (BasisOneSpin.h:46-61)
        assert np.all(np.diff(b.astype(np.int64)) > 0) or len(b) == 1
        assert all(bin(int(w)).count("1") == n for w in b[:200])
        # perfectIndex (BasisOneSpin.h:73-81) is the position in that list
        idx = [oracle.onespin_rank(int(w)) for w in b[:: max(1, len(b) // 300)]]
        assert idx == list(range(0, len(b), max(1, len(b) // 300)))
    # the enumeration is exactly "all words of that popcount, sorted"
    ref = sorted(w for w in range(1 << 8) if bin(w).count("1") == 3)
    assert list(oracle.onespin_basis(8, 3)) == ref


def test_hubbard_product_basis_indexing():
    L, nup, ndown = 6, 2, 3
    up, dn = oracle.hubbard_basis_words(L, nup, ndown)
    n1 = comb(L, nup)
    b1, b2 = oracle.onespin_basis(L, nup), oracle.onespin_basis(L, ndown)
    # state i <-> (basis1[i % N_up], basis2[i / N_up])  (BasisHubbardLanczos.h:77-84)
    i = np.arange(len(up))
    assert np.array_equal(up, b1[i % n1]) and np.array_equal(dn, b2[i // n1])
    for k in range(0, len(up), 7):
        assert oracle.lib().lppo_hubbard_perfect_index(L, nup, ndown, int(up[k]), int(dn[k])) == k


def test_golden_heisenberg_sector_enumeration_and_diagonal():
    """Reference program output (count, sum of the Jzz=1 chain diagonal over the sector) == oracle."""
    gold = json.load(open(os.path.join(HERE, "golden", "heis_inf_temp.json")))
    for c in gold["cases"]:
        L, twiceS, per = c["L"], c["twiceS"], c["periodic"]
        sz = twiceS * L // 2
        basis = oracle.heis_basis(L, twiceS, sz)
        assert len(basis) == c["count"], c
        zero = np.zeros((L, L))
        A = oracle.heis_csr(L, twiceS, sz, zero, chain(L, 1.0, bool(per)))
        # with Jpm = 0 the matrix is its stored diagonal (one explicit entry per row)
        assert A.nnz == A.nrows and np.array_equal(A.colind, np.arange(A.nrows))
        assert abs(A.values.sum() - c["sum"]) < 1e-9 * max(1, abs(c["sum"])), c


def test_golden_reference_binary_if_present():
    """In the build container the reference program itself is re-run (oracle/_ref); elsewhere skipped."""
    exe = os.path.join(HERE, "..", "oracle", "_ref", "heis_inf_temp")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref not built here")
    import subprocess
    out = subprocess.check_output([exe, "12", "1", "1"], text=True).splitlines()[-1].split()
    assert out[1:] == ["-252", "924"]


def test_heisenberg_literal_index_equals_fast_index():
    L = 10
    jpm, jzz = chain(L, 1.0, True), chain(L, 0.5, True)
    a = oracle.heis_csr(L, 1, 5, jpm, jzz, literal_index=True)  # O(N) scan, BasisHeisenberg.h:73-80
    b = oracle.heis_csr(L, 1, 5, jpm, jzz, literal_index=False)
    assert np.array_equal(a.rowptr, b.rowptr) and np.array_equal(a.colind, b.colind) and np.array_equal(a.values, b.values)
    # S=1 chain (2 bits per site)
    a = oracle.heis_csr(5, 2, 5, chain(5, 1.0), chain(5, 1.0), literal_index=True)
    b = oracle.heis_csr(5, 2, 5, chain(5, 1.0), chain(5, 1.0), literal_index=False)
    assert np.array_equal(a.colind, b.colind) and np.array_equal(a.values, b.values)
    # S=1/2 basis == one-spin basis (same ascending order), so rank == combinatorial rank
    assert np.array_equal(oracle.heis_basis(12, 1, 6), oracle.onespin_basis(12, 6))


def test_tj_basis_sorted_and_indices_agree():
    L, nup, ndown = 8, 3, 2
    basis = oracle.tj_basis(L, nup, ndown)
    assert len(basis) == comb(L, ndown) * comb(L - ndown, nup)
    assert np.all(np.diff(basis.astype(np.int64)) > 0)
    mask = (1 << L) - 1
    assert all((int(w) & mask) & (int(w) >> L) == 0 for w in basis)
    lib = oracle.lib()
    cfree = comb(L - ndown, nup)
    for k in range(0, len(basis), 5):
        w = int(basis[k])
        up, dn = w & mask, w >> L
        assert lib.lppo_tj_perfect_index_literal(basis, len(basis), L, up, dn) == k  # BasisTjMultiOrbLanczos.h:70-107
        assert lib.lppo_find_bisect(basis, len(basis), w) == k
        # closed form used by the device assembler: rank(down)*C(L-nd,nu) + rank(up compressed to the free sites)
        free = [s for s in range(L) if not (dn >> s) & 1]
        upc = sum(((up >> s) & 1) << j for j, s in enumerate(free))
        assert oracle.onespin_rank(dn) * cfree + oracle.onespin_rank(upc) == k
    a = oracle.tj_csr(L, nup, ndown, chain(L, -1.0), chain(L, 0.4), chain(L, 0.4), chain(L, -0.1), literal_index=True)
    b = oracle.tj_csr(L, nup, ndown, chain(L, -1.0), chain(L, 0.4), chain(L, 0.4), chain(L, -0.1), literal_index=False)
    assert np.array_equal(a.colind, b.colind) and np.array_equal(a.values, b.values)


def test_matrices_are_hermitian_with_explicit_sorted_rows():
    mats = [
        oracle.hubbard_csr(6, 3, 3, square(2, 3, -1.0, False), np.linspace(0, 5, 6), np.linspace(-1, 1, 12)),
        oracle.heis_csr(8, 1, 4, chain(8, 1.0, True), chain(8, 0.3, True), field=np.linspace(-1, 1, 8)),
        oracle.tj_csr(6, 2, 2, chain(6, -1.0), chain(6, 0.4), chain(6, 0.4), chain(6, -0.1), force_complex=True),
    ]
    hc = chain(5, -1.0).astype(complex)
    hc[1, 2] = -0.5j
    hc[2, 1] = 0.5j
    mats.append(oracle.hubbard_csr(5, 2, 2, hc, np.full(5, 3.0)))
    for A in mats:
        D = A.to_scipy().toarray()
        assert np.abs(D - D.conj().T).max() < 1e-14
        for i in range(A.nrows):
            cols = A.colind[A.rowptr[i]:A.rowptr[i + 1]]
            assert np.all(np.diff(cols) > 0)  # sorted, duplicates merged (SparseRow::finalize)
            assert i in cols  # the diagonal is always stored (HubbardHelper.h:93)


# ------------------------------------------------------------------ independent dense ED (numpy only)
def _jw_ops(n):
    """Jordan-Wigner annihilation operators for n modes (mode 0 = least significant)."""
    I, Z = np.eye(2), np.diag([1.0, -1.0])
    a = np.array([[0.0, 1.0], [0.0, 0.0]])
    ops = []
    for k in range(n):
        mats = [Z] * k + [a] + [I] * (n - k - 1)
        m = mats[-1]
        for x in reversed(mats[:-1]):
            m = np.kron(m, x)
        ops.append(m)
    return ops


def _sector(dense, nmodes, pred):
    idx = [s for s in range(1 << nmodes) if pred(s)]
    return dense[np.ix_(idx, idx)]


def _hubbard_dense(L, hop, U, V):
    c = _jw_ops(2 * L)  # modes: up 0..L-1, down L..2L-1
    H = np.zeros((1 << (2 * L),) * 2, dtype=complex)
    for s in range(2):
        for i in range(L):
            for j in range(L):
                if hop[i, j] != 0:
                    H += hop[i, j] * c[j + s * L].conj().T @ c[i + s * L]
    n = [x.conj().T @ x for x in c]
    for i in range(L):
        H += U[i] * n[i] @ n[i + L] + V[i] * (n[i] + n[i + L])
    return H


@pytest.mark.parametrize("L,nup,ndown", [(4, 2, 2), (4, 1, 3), (5, 2, 2), (6, 3, 2)])
def test_hubbard_spectrum_vs_jordan_wigner(L, nup, ndown):
    rng = np.random.default_rng(L * 100 + nup)
    hop = square(2, L // 2, -1.0, False) if L % 2 == 0 else chain(L, -1.0, True)
    hop = hop * (1 + 0.3 * rng.random((L, L)))
    hop = (hop + hop.T) / 2
    U, V = rng.uniform(0, 6, L), rng.uniform(-1, 1, L)
    A = oracle.hubbard_csr(L, nup, ndown, hop, U, np.concatenate([V, V]))
    H = _hubbard_dense(L, hop, U, V)
    up_mask = (1 << L) - 1
    Hs = _sector(H, 2 * L, lambda s: bin(s & up_mask).count("1") == nup and bin(s >> L).count("1") == ndown)
    e1 = np.linalg.eigvalsh(A.to_scipy().toarray())
    e2 = np.linalg.eigvalsh(Hs)
    assert np.abs(e1 - e2).max() < 1e-10


@pytest.mark.parametrize("L,nup,ndown", [(4, 2, 2), (4, 1, 2), (5, 2, 2), (5, 3, 2)])
def test_super_hubbard_extended_spectrum_vs_jordan_wigner(L, nup, ndown):
    """Model=SuperHubbardExtended (HubbardHelper.h:158-177 diagonal, :282-343 spin-flip terms): the oracle's restatement against
    t + U + V + (1/2) sum V_ij n_i n_j + (1/2) sum_ij J_ij S_i . S_j built from Jordan-Wigner operators, S+_i = c+_{i up} c_{i down}."""
    rng = np.random.default_rng(3 + L + nup)
    hop = chain(L, -1.0, True) * (1 + 0.2 * rng.random((L, L)))
    hop = (hop + hop.T) / 2
    J = chain(L, 0.8, True) * (1 + 0.3 * rng.random((L, L)))
    J = (J + J.T) / 2
    if L == 5:
        J[0, 2] = J[2, 0] = 0.45  # not nearest neighbours: the sign of the electrons in between matters
    nj, U, V = chain(L, 0.5, True), rng.uniform(1, 5, L), rng.uniform(-1, 1, L)
    A = oracle.hubbard_csr(L, nup, ndown, hop, U, np.concatenate([V, V]), ninj=nj, jcoup=J)
    c = _jw_ops(2 * L)
    H = _hubbard_dense(L, hop, U, V)
    n = [x.conj().T @ x for x in c]
    Sp = [c[i].conj().T @ c[i + L] for i in range(L)]
    Sz = [0.5 * (n[i] - n[i + L]) for i in range(L)]
    for i in range(L):
        for j in range(L):
            H = H + 0.5 * nj[i, j] * (n[i] + n[i + L]) @ (n[j] + n[j + L])
            if i != j and J[i, j] != 0:
                H = H + 0.5 * J[i, j] * (Sz[i] @ Sz[j] + 0.5 * (Sp[i] @ Sp[j].conj().T + Sp[j] @ Sp[i].conj().T))
    m = (1 << L) - 1
    Hs = _sector(H, 2 * L, lambda s: bin(s & m).count("1") == nup and bin(s >> L).count("1") == ndown)
    assert np.abs(np.linalg.eigvalsh(A.to_scipy().toarray()) - np.linalg.eigvalsh(Hs)).max() < 1e-10


def test_hubbard_complex_hopping_spectrum():
    L = 4
    hop = chain(L, -1.0, True).astype(complex)
    phase = np.exp(0.37j)
    for i in range(L):
        j = (i + 1) % L
        hop[i, j] = -phase
        hop[j, i] = -np.conj(phase)
    U, V = np.full(L, 3.0), np.zeros(L)
    A = oracle.hubbard_csr(L, 2, 2, hop, U)
    assert A.is_complex
    H = _hubbard_dense(L, hop, U, V)
    m = (1 << L) - 1
    Hs = _sector(H, 2 * L, lambda s: bin(s & m).count("1") == 2 and bin(s >> L).count("1") == 2)
    # the reference's element is H[ket,bra] = h(i,j)*sign (HubbardHelper.h:224): the transpose of sum h_ij c+_j c_i
    assert np.abs(np.linalg.eigvalsh(A.to_scipy().toarray()) - np.linalg.eigvalsh(Hs)).max() < 1e-10


def test_heisenberg_spectrum_vs_pauli():
    L = 8
    jpm, jzz = chain(L, 1.0, True), chain(L, 0.6, True)
    field = np.linspace(-0.3, 0.4, L)
    sp = np.array([[0.0, 0.0], [1.0, 0.0]])  # raises bit 0 -> 1 (|1> = up)
    sz = np.diag([-0.5, 0.5])
    I = np.eye(2)

    def op(o, k):
        mats = [I] * k + [o] + [I] * (L - k - 1)
        m = mats[-1]
        for x in reversed(mats[:-1]):
            m = np.kron(m, x)
        return m
    H = np.zeros((1 << L, 1 << L))
    for i in range(L):
        H += field[i] * op(sz, i)
        for j in range(L):
            if i < j:
                H += jzz[i, j] * op(sz, i) @ op(sz, j)
            if i != j and jpm[i, j] != 0:
                H += 0.5 * jpm[i, j] * op(sp, i) @ op(sp.T, j)
    for nupbits in (3, 4):
        A = oracle.heis_csr(L, 1, nupbits, jpm, jzz, field=field)
        Hs = _sector(H, L, lambda s: bin(s).count("1") == nupbits)
        assert np.abs(np.linalg.eigvalsh(A.to_scipy().toarray()) - np.linalg.eigvalsh(Hs)).max() < 1e-10


def test_tj_spectrum_vs_projected_fermions():
    """t-J = P [ sum t c+c + J(S.S) + W n n ] P on the no-double-occupancy subspace."""
    L, nup, ndown = 5, 2, 2
    t, J, W = chain(L, -1.0), chain(L, 0.4), chain(L, -0.1)
    c = _jw_ops(2 * L)
    n = [x.conj().T @ x for x in c]
    dim = 1 << (2 * L)
    H = np.zeros((dim, dim))
    for s in range(2):
        for i in range(L):
            for j in range(L):
                if t[i, j] != 0:
                    H += t[i, j] * c[j + s * L].T @ c[i + s * L]
    for i in range(L):
        for j in range(i + 1, L):
            if J[i, j] != 0:
                szi, szj = 0.5 * (n[i] - n[i + L]), 0.5 * (n[j] - n[j + L])
                spi, smi = c[i].T @ c[i + L], c[i + L].T @ c[i]
                spj, smj = c[j].T @ c[j + L], c[j + L].T @ c[j]
                H += J[i, j] * (szi @ szj + 0.5 * (spi @ smj + smi @ spj))
            if W[i, j] != 0:
                H += W[i, j] * (n[i] + n[i + L]) @ (n[j] + n[j + L])
    m = (1 << L) - 1
    Hs = _sector(H, 2 * L, lambda s: bin(s & m).count("1") == nup and bin(s >> L).count("1") == ndown and (s & m) & (s >> L) == 0)
    A = oracle.tj_csr(L, nup, ndown, t, J, J, W)
    assert np.abs(np.linalg.eigvalsh(A.to_scipy().toarray()) - np.linalg.eigvalsh(Hs)).max() < 1e-10


# ------------------------------------------------------------------ closed forms
def test_closed_form_energies():
    # TestSuite/inputs/input0.inp: L=4 OBC, t=-1 connectors, U=0, 2 up 2 down: free fermions, E0 = -2 sqrt 5
    A = oracle.hubbard_csr(4, 2, 2, chain(4, -1.0), np.zeros(4))
    assert A.nrows == 36
    e, _, _ = oracle.lanczos_solve(A, oracle.fill_random(36, 1234))
    assert abs(e[0] + 2 * np.sqrt(5)) < 1e-12
    # two-site Hubbard, one up one down: E0 = (U - sqrt(U^2 + 16 t^2)) / 2
    for U in (0.0, 2.0, 8.0):
        A = oracle.hubbard_csr(2, 1, 1, chain(2, -1.0), np.full(2, U))
        e0 = np.linalg.eigvalsh(A.to_scipy().toarray())[0]
        assert abs(e0 - (U - np.sqrt(U * U + 16)) / 2) < 1e-12
    # 4-site Heisenberg ring, J=1: E0 = -2
    A = oracle.heis_csr(4, 1, 2, chain(4, 1.0, True), chain(4, 1.0, True))
    assert abs(np.linalg.eigvalsh(A.to_scipy().toarray())[0] + 2.0) < 1e-12


# ------------------------------------------------------------------ the numerical core of the oracle
def test_tridiag_eig_vs_numpy():
    rng = np.random.default_rng(5)
    for n in (1, 2, 3, 10, 57):
        d, e = rng.normal(size=n), rng.normal(size=max(n - 1, 0))
        T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
        w, z = oracle.tridiag_eig(d, e, vectors=True)
        assert np.abs(w - np.linalg.eigvalsh(T)).max() < 1e-12
        assert np.abs(T @ z - z * w).max() < 1e-11


def test_lanczos_oracle_vs_dense_and_otf():
    L = 8
    hop, U = chain(L, -1.0, True), np.full(L, 4.0)
    A = oracle.hubbard_csr(L, 4, 4, hop, U)
    dense = np.linalg.eigvalsh(A.to_scipy().toarray())
    init = oracle.fill_random(A.nrows, 1234)
    e, z, steps = oracle.lanczos_solve(A, init)
    assert abs(e[0] - dense[0]) < 1e-10 * abs(dense[0])
    e3, z3, _ = oracle.lanczos_solve(A, init, nstates=3, max_steps=150, eps=1e-13, reortho=True)
    assert np.abs(z3 @ z3.T - np.eye(3)).max() < 1e-8 and abs(e3[0] - dense[0]) < 1e-10
    # on-the-fly product (HubbardHelper.h:105-134) == stored product, x += H y semantics
    x0, y = oracle.fill_random(A.nrows, 3), oracle.fill_random(A.nrows, 4)
    xs = oracle.spmv_acc(A, x0.copy(), y)
    xo = x0.copy()
    oracle.hubbard_otf_mvp(L, 4, 4, hop, U, np.zeros(L), xo, y, 0, 0, 2)
    assert np.abs(xs - xo).max() < 1e-13
    assert np.abs(xs - (x0 + A.to_scipy() @ y)).max() < 1e-13
    # threaded SpMV == serial
    assert np.array_equal(oracle.spmv_acc(A, x0.copy(), y, nthreads=4), xs)


def test_config1_sizes():
    """BASELINE config 1 (Hubbard chain L=12, 6 up 6 down): N and nnz as tabulated in BASELINE.md."""
    A = oracle.hubbard_csr(12, 6, 6, chain(12, -1.0), np.full(12, 4.0))
    assert (A.nrows, A.nnz) == (853776, 11099088)


def test_tabulated_otf_product_is_bit_identical_to_the_literal_one():
    """lppo_hubbard_otf_apply (tables of one-species hops) == lppo_hubbard_otf_mvp (literal HubbardHelper.h:105-134)
    bit for bit -- same elements, same summation order -- on whole vectors and on row windows; the Lanczos loop over
    it gives the tridiagonal matrix of the stored-CSR run.  This is what carries the config-2 golden fixture."""
    rng = np.random.default_rng(11)
    for (L, nu, nd, hop) in [(8, 4, 4, square(2, 4, -1.0)), (8, 3, 5, chain(8, -1.3, True)), (10, 5, 4, square(2, 5, -0.7)),
                             (6, 0, 3, chain(6, -1.0)), (5, 5, 2, chain(5, -1.0, True))]:
        U, V = rng.standard_normal(L), rng.standard_normal(L)
        n = oracle.lib().lppo_hubbard_size(L, nu, nd)
        y, x0 = oracle.fill_random(n, 3), oracle.fill_random(n, 4)
        xa = x0.copy()
        oracle.hubbard_otf_mvp(L, nu, nd, hop, U, V, xa, y, 0, 0, 3)
        H = oracle.HubbardOtf(L, nu, nd, hop, U, V, nthreads=3)
        assert H.nrows == n
        assert np.array_equal(H.apply(x0.copy(), y), xa)
        r0, r1 = n // 3, max(n // 3 + 1, (2 * n) // 3 + 1)
        xb, xc = x0.copy(), x0.copy()
        oracle.hubbard_otf_mvp(L, nu, nd, hop, U, V, xb, y, r0, r1, 2)
        assert np.array_equal(H.apply(xc, y, r0, r1), xb)
        if n > 100:
            A = oracle.hubbard_csr(L, nu, nd, hop, U, V)
            init = oracle.fill_random(n, 1234)
            s1, a1, b1, _, h1 = oracle.lanczos_decomposition(A, init)
            s2, a2, b2, h2 = H.lanczos(init)
            assert s1 == s2 and rel(a2, a1) < 1e-10 and rel(b2, b1) < 1e-10 and abs(h1[-1] - h2[-1]) < 1e-11 * abs(h1[-1])


def test_threaded_assembly_equals_one_chunk():
    """The row-chunked (OpenMP) assemblers give the same CSR whatever the chunking: compare against a matrix small
    enough for a single chunk built row by row through the dense ED cross-checks above, and a multi-chunk one against
    its own transpose (hermiticity, isHermitian assert of Heisenberg.h:113) and the literal-index variant."""
    A = oracle.heis_csr(16, 1, 8, chain(16, 1.0, True), chain(16, 0.8, True))  # 12870 rows: one chunk
    B = oracle.heis_csr(20, 1, 10, chain(20, 1.0, True), chain(20, 0.8, True))  # 184756 rows: nine chunks
    assert B.nrows == 184756 and np.all(np.diff(B.rowptr) > 0)
    M = B.to_scipy()
    assert abs(M - M.T).max() == 0
    for rr in range(0, B.nrows, 997):  # rows are sorted by column and hold no duplicates (SparseRow::finalize)
        c = B.colind[B.rowptr[rr]:B.rowptr[rr + 1]]
        assert np.all(np.diff(c) > 0)
    assert A.nrows == 12870
    T1 = oracle.tj_csr(12, 4, 4, chain(12, -1.0), chain(12, 0.4), chain(12, 0.4), chain(12, -0.1))
    T2 = oracle.tj_csr(12, 4, 4, chain(12, -1.0), chain(12, 0.4), chain(12, 0.4), chain(12, -0.1), literal_index=True)
    assert np.array_equal(T1.rowptr, T2.rowptr) and np.array_equal(T1.colind, T2.colind) and np.array_equal(T1.values, T2.values)


def test_config2_golden_fixture_is_self_consistent():
    """tests/golden/c2_hubbard4x4_U4.json (made by make_c2_hubbard4x4_U4.py): E0 is the lowest eigenvalue of the
    tridiagonal matrix of its own coefficients, the run stopped by the reference's rule, and the value is the
    literature energy of the 4x4 periodic cluster at U = 4t (-13.6219 t)."""
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "c2_hubbard4x4_U4.json")))
    a, b, h = np.array(g["a"]), np.array(g["b"]), np.array(g["e0_history"])
    assert g["rows"] == 165636900 and len(a) == len(b) == len(h) == g["steps"] < g["max_steps"]
    w = oracle.tridiag_eig(a, b[:-1])
    assert abs(w[0] - g["e0"]) < 1e-12 * abs(g["e0"]) and h[-1] == g["e0"]
    assert abs(h[-1] - h[-2]) < g["eps"] and abs(h[-2] - h[-3]) >= g["eps"]
    assert abs(g["e0"] + 13.62185) < 2e-4


def test_hubbard_time_dependent_potential_is_a_site_potential():
    """HubbardHelper.h:180-183: tmp = potentialV[i]; tmp += potentialT[i]*timeFactor; s += tmp*ne -- the literal restatement equals
    the matrix with the site potential V + T*timeFactor (formed with the same two operations), differs from the one without it,
    and its diagonal moves by exactly sum_i T_i*timeFactor*n_i."""
    from helpers import chain
    L, nup, ndown = 6, 3, 2
    hop, U = chain(L, -1.0, True), np.linspace(1.0, 3.5, L)
    V = np.linspace(-0.5, 0.5, L)
    T, tf = np.array([0.3, -0.1, 0.7, 0.0, 0.2, -0.6]), 0.37
    A0 = oracle.hubbard_csr(L, nup, ndown, hop, U, V)
    A1 = oracle.hubbard_csr(L, nup, ndown, hop, U, V, potentialT=T, timeFactor=tf)
    A2 = oracle.hubbard_csr(L, nup, ndown, hop, U, V + T * tf)
    assert np.array_equal(A1.rowptr, A2.rowptr) and np.array_equal(A1.colind, A2.colind)
    assert np.array_equal(A1.values.view(np.uint64), A2.values.view(np.uint64))
    assert not np.array_equal(A1.values, A0.values)
    d0 = A0.to_scipy().diagonal()
    d1 = A1.to_scipy().diagonal()
    assert np.isclose((d1 - d0).sum(), (T * tf).sum() * (nup + ndown) / L * A0.nrows, rtol=1e-12)  # every site is occupied by (nup+ndown)/L electrons on average
