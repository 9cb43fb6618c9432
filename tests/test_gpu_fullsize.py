"""-m gpu: size-independent parity checks at BASELINE's full sizes (no CPU oracle can follow there).

Free fermions (U=0): the many-body ground-state energy is twice the sum of the lowest single-particle
levels of the hopping matrix -- an exact closed form the Lanczos run must reproduce to 1e-10,
at 1.66e8 states through the stored CSR (config 2 shape, 5.8e9 non-zeros assembled on the device) and
at 2.36e9 states (beyond 2^31 rows; ~1 TB as a CSR) through the matrix-free engine.
"""
import numpy as np
import pytest

from helpers import square
from lanczosplusplus_amd import LanczosEngine

pytestmark = pytest.mark.gpu


def _exact(hop, n):
    lev = np.sort(np.linalg.eigvalsh(hop))
    return 2 * lev[:n].sum()


def test_config2_shape_stored_csr_free_fermions():
    L = 16
    hop = square(4, 4, -1.0, pbc=True)
    exact = _exact(hop, 8)
    with LanczosEngine(max_steps=300, eps=1e-11, save_vectors=0) as e:
        e.assemble_hubbard(L, 8, 8, hop, np.zeros(L))
        st0 = e.stats()
        assert (st0["nrows"], st0["nnz"]) == (165636900, 5819376420)
        eg, _, st = e.lanczos(1, want_vectors=False)
    assert abs(eg[0] - exact) <= 1e-10 * abs(exact), (eg[0], exact, st["steps"])


def test_2e9_states_matrix_free_free_fermions():
    L = 18
    hop = square(3, 6, -1.0, pbc=True)
    exact = _exact(hop, 9)
    with LanczosEngine(max_steps=300, eps=1e-11, save_vectors=0) as e:
        e.setup_hubbard_onthefly(L, 9, 9, hop, np.zeros(L))
        assert e.rows() == 48620 * 48620 > 2 ** 31
        eg, _, st = e.lanczos(1, want_vectors=False)
    assert abs(eg[0] - exact) <= 1e-10 * abs(exact), (eg[0], exact, st["steps"])


def test_config2_shape_matrix_free_equals_stored_energy_with_U(monkeypatch):
    """Same 4x4 matrix, U=4: the stored engine (product-basis kernels) and the matrix-free engine's fused block-order kernel
    (LPP_ONTHEFLY_KRON=1: two independent implementations of the product) agree to 1e-10."""
    monkeypatch.setenv("LPP_ONTHEFLY_KRON", "1")
    L = 16
    hop, U = square(4, 4, -1.0, pbc=True), np.full(L, 4.0)
    with LanczosEngine(max_steps=300, eps=1e-11, save_vectors=0) as e:
        e.setup_hubbard_onthefly(L, 8, 8, hop, U)
        e1, _, s1 = e.lanczos(1, want_vectors=False)
        e.assemble_hubbard(L, 8, 8, hop, U)
        e2, _, s2 = e.lanczos(1, want_vectors=False)
    assert abs(e1[0] - e2[0]) <= 1e-10 * abs(e2[0]) and s1["steps"] == s2["steps"]
    # literature cross-check: the 4x4 periodic Hubbard cluster at U=4t, half filling, has E0 = -13.6219 t
    # (E0/N = -0.8514, exact diagonalisation results quoted since Fano, Ortolani & Parola 1990)
    assert abs(e2[0] - (-13.62185)) < 2e-4


def test_config2_shape_products_agree_and_are_hermitian(monkeypatch):
    """x += H y at config 2's full size: the stored engine (compressed layout: shared offsets, block template, diagonal
    codes) and the matrix-free engine give the same vector, the product is linear, and <u|H v> = <H u|v>."""
    L = 16
    hop, U = square(4, 4, -1.0, pbc=True), np.full(L, 4.0)
    n = 12870 * 12870
    rng = np.random.default_rng(42)
    u = rng.standard_normal(n)
    v = rng.standard_normal(n)
    with LanczosEngine(save_vectors=0) as e:
        e.assemble_hubbard(L, 8, 8, hop, U)
        lay = e.layout()
        assert lay["block_template"] == 2 and lay["diagonal_codes"] == 1 and lay["local16"] == 1
        assert lay["resident_bytes"] < 6e9  # 71 GB as a plain CSR
        hv = e.matrixVectorProduct(np.zeros(n), v)
        hu = e.matrixVectorProduct(np.zeros(n), u)
        acc = e.matrixVectorProduct(u.copy(), v)           # accumulate form
        lin = e.matrixVectorProduct(np.zeros(n), 2.0 * u - 0.5 * v)
    assert np.max(np.abs(acc - (u + hv))) <= 1e-12 * np.max(np.abs(hv))
    assert np.max(np.abs(lin - (2.0 * hu - 0.5 * hv))) <= 1e-12 * np.max(np.abs(hv))
    assert abs(np.dot(u, hv) - np.dot(hu, v)) <= 1e-11 * abs(np.dot(u, hv))
    monkeypatch.setenv("LPP_ONTHEFLY_KRON", "1")  # the fused block-order kernel: an implementation of its own
    with LanczosEngine(save_vectors=0) as e:
        e.setup_hubbard_onthefly(L, 8, 8, hop, U)
        hv2 = e.matrixVectorProduct(np.zeros(n), v)
    assert np.max(np.abs(hv2 - hv)) <= 1e-13 * np.max(np.abs(hv))


def test_3e8_states_stored_csr_free_fermions(monkeypatch):
    """Twice config 2: the 3x6 cluster with 6 up / 6 down electrons -- 344,622,096 states, 1.2e10 non-zeros, 144 GB as a plain
    CSR while it is being assembled on the device, ~8 GB once it is in the compressed layout -- through the STORED engine, general
    layout (LPP_PRODUCT_LAYOUT=0: by itself this sector takes the product-basis form in two pieces per row, next test)."""
    monkeypatch.setenv("LPP_PRODUCT_LAYOUT", "0")
    L = 18
    hop = square(3, 6, -1.0, pbc=True)
    exact = _exact(hop, 6)
    with LanczosEngine(max_steps=300, eps=1e-11, save_vectors=0) as e:
        e.assemble_hubbard(L, 6, 6, hop, np.zeros(L))
        st0, lay = e.stats(), e.layout()
        assert (st0["nrows"], st0["nnz"]) == (18564 * 18564, 12021229584)
        assert lay["block_template"] == 2 and lay["resident_bytes"] < 12e9
        eg, _, st = e.lanczos(1, want_vectors=False)
    assert abs(eg[0] - exact) <= 1e-10 * abs(exact), (eg[0], exact, st["steps"])


def test_3e8_states_product_basis_form_in_pieces_free_fermions():
    """The same sector by itself: a species' row of 18,564 positions fits an LDS window alone but not with its diagonal codes and list
    heads beside it, so it is cut into pieces (k_pb_up_big2); 0.4 GB resident, exact energy."""
    L = 18
    hop = square(3, 6, -1.0, pbc=True)
    exact = _exact(hop, 6)
    with LanczosEngine(max_steps=300, eps=1e-11, save_vectors=0) as e:
        e.assemble_hubbard(L, 6, 6, hop, np.zeros(L))
        lay = e.layout()
        assert lay["kernel"] == 4 and lay["pieces"] > 1 and lay["resident_bytes"] < 1e9 and lay["nnz"] == 12021229584
        eg, _, st = e.lanczos(1, want_vectors=False)
    assert abs(eg[0] - exact) <= 1e-10 * abs(exact), (eg[0], exact, st["steps"])


def test_config5_lattice_sector_matrix_free_free_fermions():
    """BASELINE config 5's lattice (4x5, periodic), the (6,6) sector of SURVEY 8(e): 1,502,337,600 states (0.67 TB as a CSR),
    matrix-free on one GPU; exact free-fermion energy."""
    L = 20
    hop = square(4, 5, -1.0, pbc=True)
    exact = _exact(hop, 6)
    with LanczosEngine(max_steps=300, eps=1e-11, save_vectors=0) as e:
        e.setup_hubbard_onthefly(L, 6, 6, hop, np.zeros(L))
        assert e.rows() == 38760 * 38760 == 1502337600
        lay = e.layout()  # rows of 38760 positions in pieces, couplings over parts of the 38760 blocks (lpp_pbig_kernels.h)
        assert lay["kernel"] == 4 and lay["pieces"] > 1
        eg, _, st = e.lanczos(1, want_vectors=False)
    assert abs(eg[0] - exact) <= 1e-10 * abs(exact), (eg[0], exact, st["steps"])


def test_config5_lattice_products_agree_beyond_the_lds_window(monkeypatch):
    """x += H y where a block's row no longer fits one LDS window (4x5 lattice, 6 up / 5 down: N_up = 38760 in pieces, 15504 blocks,
    6.0e8 states): the product-basis kernels for long rows (lpp_pbig_kernels.h) against the fused block-order kernel
    (LPP_ONTHEFLY_KRON=1, an implementation of its own), plus linearity, the accumulate form and <u|H v> = <H u|v>."""
    L = 20
    hop, U = square(4, 5, -1.0, pbc=True), np.full(L, 4.0)
    n = 38760 * 15504
    rng = np.random.default_rng(43)
    u = rng.standard_normal(n)
    v = rng.standard_normal(n)
    with LanczosEngine(save_vectors=0) as e:
        e.setup_hubbard_onthefly(L, 6, 5, hop, U)
        lay = e.layout()
        assert lay["kernel"] == 4 and lay["pieces"] > 1 and lay["segments"] == 0  # below 65536 positions: the per-position template (k_pb_up_big2)
        hv = e.matrixVectorProduct(np.zeros(n), v)
        hu = e.matrixVectorProduct(np.zeros(n), u)
        acc = e.matrixVectorProduct(u.copy(), v)
        lin = e.matrixVectorProduct(np.zeros(n), 2.0 * u - 0.5 * v)
    monkeypatch.setenv("LPP_PB_SEG", "1")  # the same rows decomposed by the high sites of the basis word (k_pb_up_seg; what longer rows take by themselves)
    with LanczosEngine(save_vectors=0) as e:
        e.setup_hubbard_onthefly(L, 6, 5, hop, U)
        assert e.layout()["segments"] == 32
        hv3 = e.matrixVectorProduct(np.zeros(n), v)
    monkeypatch.delenv("LPP_PB_SEG")
    assert np.max(np.abs(hv3 - hv)) <= 1e-13 * np.max(np.abs(hv))
    assert np.max(np.abs(acc - (u + hv))) <= 1e-12 * np.max(np.abs(hv))
    assert np.max(np.abs(lin - (2.0 * hu - 0.5 * hv))) <= 1e-12 * np.max(np.abs(hv))
    assert abs(np.dot(u, hv) - np.dot(hu, v)) <= 1e-11 * abs(np.dot(u, hv))
    monkeypatch.setenv("LPP_ONTHEFLY_KRON", "1")
    with LanczosEngine(save_vectors=0) as e:
        e.setup_hubbard_onthefly(L, 6, 5, hop, U)
        hv2 = e.matrixVectorProduct(np.zeros(n), v)
    assert np.max(np.abs(hv2 - hv)) <= 1e-13 * np.max(np.abs(hv))


def test_complex_hoppings_take_the_product_basis_layout_free_fermions():
    """4x4 lattice with a Peierls phase on every bond, 6 up 6 down (6.4e7 complex states, 1 GB per vector): large enough for the
    product-basis layout to be chosen by itself (realified in-block matrix + complex couplings, 0.14 GB instead of 8.6 GB); the exact
    free-fermion energy of the Hermitian hopping matrix to 1e-10, and x += H y against the general layout."""
    L = 16
    hop = square(4, 4, -1.0, pbc=True) * np.where(np.triu(np.ones((L, L)), 1) > 0, np.exp(0.2j), np.exp(-0.2j))
    assert np.allclose(hop, hop.conj().T)
    exact = _exact(hop, 6)
    rng = np.random.default_rng(3)
    n = 8008 * 8008
    y = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    with LanczosEngine(dtype="c128", max_steps=300, eps=1e-11, save_vectors=0) as e:
        e.assemble_hubbard(L, 6, 6, hop, np.zeros(L))
        lay = e.layout()
        assert lay["kernel"] == 4 and lay["resident_bytes"] < 0.2e9 and e.stats()["nrows"] == n
        xp = e.matrixVectorProduct(np.zeros(n, dtype=complex), y)
        eg, _, st = e.lanczos(1, want_vectors=False)
        assert abs(eg[0] - exact) <= 1e-10 * abs(exact), (eg[0], exact, st["steps"])
    import os

    os.environ["LPP_PB_COMPLEX"] = "0"
    try:
        with LanczosEngine(dtype="c128") as e:
            e.assemble_hubbard(L, 6, 6, hop, np.zeros(L))
            assert e.layout()["kernel"] != 4
            xg = e.matrixVectorProduct(np.zeros(n, dtype=complex), y)
    finally:
        del os.environ["LPP_PB_COMPLEX"]
    assert np.linalg.norm(xp - xg) <= 1e-13 * np.linalg.norm(xg)


def test_complex_hoppings_beyond_one_lds_window_free_fermions():
    """4x4 lattice with a Peierls phase on every bond, 8 up 7 down (1.47e8 complex states, 2.36 GB per vector): 12870 complex positions per
    row do not fit one LDS window, so the realified in-block matrix is held in pieces (k_pb_up_big with four value groups) and the
    couplings stay complex -- 0.31 GB resident instead of the general layout's 19.5 GB.  Exact free-fermion energy to 1e-10; the product
    is Hermitian (<x|Hy> = conj <y|Hx>) to rounding."""
    L = 16
    hop = square(4, 4, -1.0, pbc=True) * np.where(np.triu(np.ones((L, L)), 1) > 0, np.exp(0.2j), np.exp(-0.2j))
    lev = np.sort(np.linalg.eigvalsh(hop))
    exact = lev[:8].sum() + lev[:7].sum()
    n = 12870 * 11440
    rng = np.random.default_rng(11)
    with LanczosEngine(dtype="c128", max_steps=300, eps=1e-11, save_vectors=0) as e:
        e.assemble_hubbard(L, 8, 7, hop, np.zeros(L))
        lay = e.layout()
        assert lay["kernel"] == 4 and lay["pieces"] > 1 and lay["segments"] == 0 and lay["resident_bytes"] < 0.5e9 and e.stats()["nrows"] == n
        x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        y = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        hy = e.matrixVectorProduct(np.zeros(n, dtype=complex), y)
        hx = e.matrixVectorProduct(np.zeros(n, dtype=complex), x)
        a, b = np.vdot(x, hy), np.vdot(y, hx)
        assert abs(a - np.conj(b)) <= 1e-12 * abs(a), (a, b)
        del hy, hx
        eg, _, st = e.lanczos(1, want_vectors=False)
        assert abs(eg[0] - exact) <= 1e-10 * abs(exact), (eg[0], exact, st["steps"])


def test_config2_handed_over_as_a_host_csr_full_size():
    """The reference's hand-over at BASELINE config 2's size: DefaultSymmetry.h:54-57 -> InternalProductStored.h:116 delivers a HOST CSR.
    165,636,900 rows / 5,819,376,420 non-zeros = 71 GB of int64 row pointers + int32 columns + f64 values go through lpp_engine_set_csr
    without a hint: PCIe upload, basis-block detection, T / C / D read off the matrix, every row verified against them bit by bit
    (pb_from_csr), then the same layout, the same ground-state energy and the same stopping step as the device-assembled matrix, and
    lpp_engine_get_csr hands the uploaded arrays back.  (The host arrays come from reading a device-assembled matrix back -- itself
    checked bit for bit against the oracle at the sizes the oracle reaches -- because the oracle's host assembly takes minutes here.)"""
    import json
    import os
    L = 16
    hop, U = square(4, 4, -1.0, pbc=True), np.full(L, 4.0)
    with LanczosEngine(max_steps=200) as d:
        d.assemble_hubbard(L, 8, 8, hop, U)
        lay_d = d.layout()
        rp, ci, va = d.get_csr()
        e_d, _, st_d = d.lanczos(1, want_vectors=False)
    assert len(rp) - 1 == 165636900 and len(ci) == 5819376420 and rp[-1] == len(ci)
    assert rp.nbytes + ci.nbytes + va.nbytes > 71e9
    with LanczosEngine(max_steps=200) as e:
        e.set_csr(rp, ci, va)
        lay = e.layout()
        assert lay["kernel"] == 4 and lay == lay_d, (lay, lay_d)
        eg, _, st = e.lanczos(1, want_vectors=False)
        assert st["steps"] == st_d["steps"] and abs(eg[0] - e_d[0]) <= 1e-12 * abs(e_d[0])
        fix = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "c2_hubbard4x4_U4.json")))
        assert abs(eg[0] - fix["e0"]) <= 1e-10 * abs(fix["e0"])  # the CPU-oracle run of this configuration
        # lpp_engine_get_csr from the layout the upload ended in: the bits that went in (row pointers whole, three windows of 2e7 entries)
        r2, c2, v2 = e.get_csr()
        assert np.array_equal(r2, rp) and len(c2) == len(ci) and len(v2) == len(va)
        for lo in (0, len(ci) // 2 - 10 ** 7, len(ci) - 2 * 10 ** 7):
            w = slice(lo, lo + 2 * 10 ** 7)
            assert np.array_equal(c2[w], ci[w]) and np.array_equal(v2[w].view(np.uint64), va[w].view(np.uint64))


def test_plain_format_matrix_with_line_aligned_blocks_takes_the_split_panel_order(monkeypatch):
    """Plain 12-byte format (no value dictionary, no shared offsets, no product-basis layout) of the 4x4 cluster with 7 + 7 electrons:
    N_up = 11440 = 16 x 715, so a panel of 16 positions is one 128-byte line of every source block and the entries that leave the row
    blocks are taken panel-major by themselves (round 4; 13.2 against 13.9 ms per product).  130,873,600 rows: exact free-fermion
    energy, the same through the one-kernel form."""
    for k in ("LPP_PRODUCT_LAYOUT", "LPP_COMPRESS_VALUES", "LPP_SHARED_OFFSETS"):
        monkeypatch.setenv(k, "0")
    L = 16
    hop = square(4, 4, -1.0, pbc=True)
    lev = np.sort(np.linalg.eigvalsh(hop))
    exact = 2 * lev[:7].sum()
    es = []
    for sp in (None, "0"):
        if sp is not None:
            monkeypatch.setenv("LPP_SPLIT_PANEL", sp)
        with LanczosEngine(max_steps=300, eps=1e-11, save_vectors=0, compress_values=0) as e:
            e.assemble_hubbard(L, 7, 7, hop, np.zeros(L))
            lay = e.layout()
            assert lay["kernel"] == 3 and lay["coded"] == 0 and (lay["split_panel"] >= 1) == (sp is None), lay
            assert e.stats()["nrows"] == 11440 * 11440
            eg, _, st = e.lanczos(1, want_vectors=False)
        assert abs(eg[0] - exact) <= 1e-10 * abs(exact), (eg[0], exact, st["steps"])
        es.append(eg[0])
    assert abs(es[0] - es[1]) <= 1e-12 * abs(exact)


def test_xx_chain_L32_without_a_csr_free_fermions():
    """A spin chain beyond what a CSR allows (round 5): L = 32, Sz = 0 -- 601,080,390 states, 9.6e9 entries, 115 GB as a plain CSR.  The chain
    layout is planned from the couplings alone (pb_chain: no matrix is assembled, the diagonal comes straight from the states, the layout
    is checked against the assembler's row walk by one product), so the engine holds its vectors and a few tables.  With J_zz = 0 the model
    is the XX chain in a staggered field: by Jordan-Wigner free fermions with hopping J/2 and on-site energies h_i (Heisenberg.h:251-307:
    S+S- moves an up spin between neighbours, the field term is h_i (n_i - 1/2)) -- an exact energy at full size."""
    from helpers import chain
    L, m, J, h = 32, 16, 1.0, 0.3
    field = h * (-1.0) ** np.arange(L)
    single = chain(L, 0.5 * J) + np.diag(field)
    exact = np.sort(np.linalg.eigvalsh(single))[:m].sum() - 0.5 * field.sum()
    with LanczosEngine(max_steps=400, eps=1e-11, save_vectors=0) as e:
        e.assemble_heisenberg(L, m, chain(L, J), np.zeros((L, L)), field)
        assert e.rows() == 601080390
        lay = e.layout()  # 17 high sites: 2^17 - 2 segments of <= C(15, 7) positions; up to 16 high-high hops per segment
        assert lay["kernel"] == 4 and lay["segments"] == 131070 and lay["nnz"] == 601080390 + 2 * 31 * 155117520, lay  # the diagonal + two antiparallel settings of every bond
        assert lay["resident_bytes"] < 1.0e9, lay  # one diagonal code per row + tables (the boundary's permutation is not part of the matrix)
        eg, _, st = e.lanczos(1, want_vectors=False)
    assert abs(eg[0] - exact) <= 1e-10 * abs(exact), (eg[0], exact, st["steps"])


def test_tj_four_holes_hole_major_form_against_the_general_layout(monkeypatch):
    """The t-J form without a stored matrix at a shape its small parity cases do not reach: the 4x5 torus with FOUR holes (8 up, 8 down) --
    4845 hole configurations x 12870 spin patterns = 62,355,150 states, complex engine.  No CPU oracle follows there in test time; the same
    device-assembled model in the general layout (a 1.9e9-entry CSR) does: coefficients of the first 30 steps to 1e-9, E0 to 1e-10, same
    stopping step, and x += H y of one random vector element by element."""
    from helpers import rel
    from lanczosplusplus_amd import tridiag_lowest
    L, nup, ndown, t, J = 20, 8, 8, -1.0, 0.4
    lat = lambda v: square(5, 4, v, pbc=True)
    rng = np.random.default_rng(11)
    n = 4845 * 12870
    y = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex128)
    res = {}
    for form in ("1", "0"):
        monkeypatch.setenv("LPP_TJ_LAYOUT", form)
        with LanczosEngine(dtype="c128", max_steps=300, save_vectors=0) as e:
            e.assemble_tj(L, nup, ndown, lat(t), lat(J), lat(J), lat(-J / 4))
            assert e.rows() == n
            lay = e.layout()
            assert (lay["kernel"] == 5) == (form == "1") and lay["nnz"] > 0
            x = e.matrixVectorProduct(np.zeros(n, np.complex128), y)
            a, b, st = e.decomposition()
        res[form] = (lay["nnz"], x, a, b)
    (z1, x1, a1, b1), (z0, x0, a0, b0) = res["1"], res["0"]
    assert z1 == z0  # the entry count of the CSR the hole-major form stands for is the assembled one's
    assert rel(x1, x0) < 1e-13
    assert abs(len(a1) - len(a0)) <= 1 and rel(a1[:30], a0[:30]) < 1e-9 and rel(b1[:30], b0[:30]) < 1e-9
    e1, e0 = tridiag_lowest(a1, b1[:-1], 1)[0], tridiag_lowest(a0, b0[:-1], 1)[0]
    assert abs(e1 - e0) <= 1e-10 * abs(e0), (e1, e0)


@pytest.mark.parametrize("nup,ndown", [(10, 3), (3, 10)])
def test_config5_as_written_species_shapes(nup, ndown, monkeypatch):
    """BASELINE config 5 as literally written (4x5, 10 up 10 down: 184756 x 184756 = 3.4e10 states, 273 GB per vector) does not fit one GPU;
    the shapes its two kernels see do, on sectors of 2.1e8 states: (10,3) -- rows of 184756 positions, the in-block kernel in the segmented
    form; (3,10) -- 184756 blocks, the coupling kernel with 24-bit block numbers walking every workgroup's 5774 blocks in 7 rounds per panel
    (round 5; the product layout is forced onto rows of 1140 positions).  Exact free-fermion energy, and with U = 4 the same energy and
    stopping step as the fused block-order kernel that such sectors took before."""
    if ndown == 10:
        monkeypatch.setenv("LPP_PRODUCT_LAYOUT", "1")
    L = 20
    hop = square(4, 5, -1.0, pbc=True)
    lev = np.sort(np.linalg.eigvalsh(hop))
    exact = lev[:nup].sum() + lev[:ndown].sum()
    got = {}
    for U, kron in ((0.0, "0"), (4.0, "0"), (4.0, "1")):
        monkeypatch.setenv("LPP_ONTHEFLY_KRON", kron)
        with LanczosEngine(max_steps=400, eps=1e-11, save_vectors=0) as e:
            e.setup_hubbard_onthefly(L, nup, ndown, hop, np.full(L, U))
            assert e.rows() == 184756 * 1140
            if kron == "0":
                lay = e.layout()
                assert lay["kernel"] == 4 and lay["coupling_parts"] == 1, lay
                assert (lay["segments"], lay["pieces"], lay["coupling_rounds"]) == ((32, 31, 1) if nup == 10 else (0, 1, 7)), lay
            eg, _, st = e.lanczos(1, want_vectors=False)
            got[(U, kron)] = (eg[0], st["steps"])
    assert abs(got[(0.0, "0")][0] - exact) <= 1e-10 * abs(exact)
    assert abs(got[(4.0, "0")][0] - got[(4.0, "1")][0]) <= 1e-10 * abs(got[(4.0, "1")][0]) and got[(4.0, "0")][1] == got[(4.0, "1")][1]
