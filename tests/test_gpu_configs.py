"""-m gpu: BASELINE.json's configurations at FULL size, each against the CPU oracle (or a closed form where no CPU can follow).

  C2  2-D Hubbard 4x4, 8 up 8 down, U = 4 (1.66e8 states): E0 and the Lanczos coefficients of the stored AND the matrix-free
      engine against tests/golden/c2_hubbard4x4_U4.json, the oracle's on-the-fly Lanczos run made in the build container.
  C3  Heisenberg S=1/2 chain L = 28, Sz = 0 (4.0e7 states, 6.0e8 non-zeros): device assembly bit-exact against the oracle
      (O(log N) index, == the reference's O(N) scan, tests/test_oracle_pins.py) and E0 against the oracle's OpenMP Lanczos.
  C4  t-J 4x5, 9 up 9 down, complex<double> (9.2e6 states, 2.4e8 non-zeros): the same two checks.
  C5  2-D Hubbard 4x5: the (7,6) sector, 3,004,675,200 states (the "~3.4e9" of BASELINE config 5), matrix-free
      on one GPU, exact free-fermion energy; and a 4x5 sector over 4 ranks through the transposition exchange.
The bar: structure/values bit-exact, energies 1e-10 relative (north_star), coefficients 1e-8.
"""
import json
import os

import numpy as np
import pytest

import oracle
from helpers import chain, rel, square
from lanczosplusplus_amd import LanczosEngine, tridiag_lowest

pytestmark = pytest.mark.gpu

E_TOL = 1e-10
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint64)


@pytest.mark.parametrize("engine", ["stored", "onthefly", "onthefly_kron"])
def test_config2_hubbard4x4_U4_against_the_cpu_oracle(engine, monkeypatch):
    """onthefly: where a species' row fits the LDS window the matrix-free engine is the product-basis one (T, C, a code per row);
    onthefly_kron: the fused block-order kernel that serves everything else (LPP_ONTHEFLY_KRON=1), kept under the same fixture."""
    if engine == "onthefly_kron":
        monkeypatch.setenv("LPP_ONTHEFLY_KRON", "1")
    g = json.load(open(os.path.join(GOLD, "c2_hubbard4x4_U4.json")))
    L = g["L"]
    hop, U = square(4, 4, -1.0, pbc=True), np.full(L, g["U"])
    with LanczosEngine(max_steps=g["max_steps"], min_steps=g["min_steps"], eps=g["eps"], save_vectors=0, seed=g["seed"]) as e:
        if engine == "stored":
            e.assemble_hubbard(L, g["nup"], g["ndown"], hop, U)
            lay = e.layout()  # a silent fall-back to the general layout would leave every check below green: say which layout ran
            assert lay["kernel"] == 4 and lay["chained_step"] == 1 and lay["pieces"] == 1 and lay["segments"] == 0, lay
        else:
            e.setup_hubbard_onthefly(L, g["nup"], g["ndown"], hop, U)
            if engine == "onthefly":
                assert e.layout()["kernel"] == 4 and e.layout()["resident_bytes"] < 0.3e9
            else:
                with pytest.raises(Exception):
                    e.layout()  # the block-order kernel stores no matrix to describe
        assert e.rows() == g["rows"]
        a, b, st = e.decomposition()  # built-in start vector == the oracle's splitmix64 stream (seed 1234)
    e0 = tridiag_lowest(a, b[:-1], 1)[0]
    assert abs(e0 - g["e0"]) <= E_TOL * abs(g["e0"]), (e0, g["e0"])
    # the reference's stopping rule sits at the rounding floor here (|dE| < 1e-12 at E = -13.6): same step, give or take one
    assert abs(len(a) - g["steps"]) <= 1, (len(a), g["steps"])
    n = 40
    assert rel(a[:n], np.array(g["a"][:n])) < 1e-8 and rel(b[:n], np.array(g["b"][:n])) < 1e-8
    # every intermediate energy of the run, not just the last one
    m = min(len(a), g["steps"])
    hist = np.array([tridiag_lowest(a[:k], b[:k - 1], 1)[0] for k in range(5, m + 1, 5)])
    assert np.abs(hist - np.array(g["e0_history"])[4:m:5]).max() <= 1e-9


def test_config3_heisenberg_L28_full_size():
    L = 28
    jpm, jzz = chain(L, 1.0), chain(L, 1.0)
    A = oracle.heis_csr(L, 1, 14, jpm, jzz)
    assert (A.nrows, A.nnz) == (40116600, 601749000)  # C(28,14) states, OBC
    with LanczosEngine(max_steps=300, save_vectors=0) as e:
        e.assemble_heisenberg(L, 14, jpm, jzz)
        lay = e.layout()  # the chain as ONE block of the segmented form (13 high sites), not the general layout
        assert lay["kernel"] == 4 and lay["segments"] == 8192, lay
        rp, ci, va = e.get_csr()
        assert np.array_equal(rp, A.rowptr) and np.array_equal(ci, A.colind) and np.array_equal(_bits(va), _bits(A.values))
        del rp, ci, va
        ag, bg, st = e.decomposition()
    steps_o, ao, bo, _, hist = oracle.lanczos_decomposition(A, oracle.fill_random(A.nrows, 1234), max_steps=300, nthreads=0)
    eg = tridiag_lowest(ag, bg[:-1], 1)[0]
    assert abs(eg - hist[-1]) <= E_TOL * abs(hist[-1]), (eg, hist[-1])
    assert abs(len(ag) - steps_o) <= 1
    assert rel(ag[:40], ao[:40]) < 1e-8 and rel(bg[:40], bo[:40]) < 1e-8
    # Bethe-ansatz scale check of the oracle itself: E0/L of the open S=1/2 chain approaches 1/4 - ln 2 = -0.4431
    assert -0.4431 < hist[-1] / L < -0.43


def test_config4_tj_4x5_complex_full_size():
    L, nup, ndown, t, J = 20, 9, 9, -1.0, 0.4
    lat = lambda v: square(5, 4, v, pbc=True)
    A = oracle.tj_csr(L, nup, ndown, lat(t), lat(J), lat(J), lat(-J / 4), force_complex=True)
    assert A.nrows == 9237800 and A.is_complex  # C(20,9)*C(11,9)
    with LanczosEngine(dtype="c128", max_steps=300, save_vectors=0) as e:
        e.assemble_tj(L, nup, ndown, lat(t), lat(J), lat(J), lat(-J / 4))
        lay = e.layout()  # round 5: no stored matrix -- 190 hole configurations x 48620 spin patterns, the CSR below is REGENERATED by the assembler
        assert lay["kernel"] == 5 and lay["rows_per_block"] == 48620 and lay["nnz"] == A.nnz and lay["resident_bytes"] < 0.2e9, lay
        rp, ci, va = e.get_csr()
        assert np.array_equal(rp, A.rowptr) and np.array_equal(ci, A.colind) and np.array_equal(_bits(va), _bits(A.values))
        del rp, ci, va
        ag, bg, st = e.decomposition()
    steps_o, ao, bo, _, hist = oracle.lanczos_decomposition(A, oracle.fill_random(A.nrows, 1234, True), max_steps=300, nthreads=0)
    eg = tridiag_lowest(ag, bg[:-1], 1)[0]
    assert abs(eg - hist[-1]) <= E_TOL * abs(hist[-1]), (eg, hist[-1])
    assert abs(len(ag) - steps_o) <= 1
    assert rel(ag[:40], ao[:40]) < 1e-8 and rel(bg[:40], bo[:40]) < 1e-8
    # the reference's hand-over at full size: the host CSR (4.7 GB) through lpp_engine_set_csr with the model's description in front of it --
    # regenerated, compared bit for bit, held without a stored matrix; same coefficients
    with LanczosEngine(dtype="c128", max_steps=300, save_vectors=0) as e:
        e.set_model_tj(L, nup, ndown, lat(t), lat(J), lat(J), lat(-J / 4))
        e.set_csr(A.rowptr, A.colind, A.values)
        lay = e.layout()
        assert lay["kernel"] == 5 and lay["nnz"] == A.nnz and lay["resident_bytes"] < 0.2e9, lay
        ah, bh, _ = e.decomposition()
    assert len(ah) == len(ag) and rel(ah, ag) < 1e-12 and rel(bh, bg) < 1e-12


def test_config5_7up6down_sector_matrix_free_free_fermions():
    """3,004,675,200 states (1.39 TB as a CSR): the sector BASELINE config 5's "~3.4e9 states" stands for."""
    L = 20
    hop = square(4, 5, -1.0, pbc=True)
    lev = np.sort(np.linalg.eigvalsh(hop))
    exact = lev[:7].sum() + lev[:6].sum()
    with LanczosEngine(max_steps=300, eps=1e-11, save_vectors=0) as e:
        e.setup_hubbard_onthefly(L, 7, 6, hop, np.zeros(L))
        assert e.rows() == 77520 * 38760 == 3004675200
        lay = e.layout()  # rows of 77520 positions: the in-block matrix decomposed by the 5 high sites (one slab of the lattice), 32 segments in 14 items
        assert lay["kernel"] == 4 and (lay["segments"], lay["pieces"]) == (32, 14), lay
        assert lay["coupling_rounds"] == 4  # 1212 blocks per workgroup of the coupling kernel, walked in four pieces per panel (round 5)
        eg, _, st = e.lanczos(1, want_vectors=False)
    assert abs(eg[0] - exact) <= E_TOL * abs(exact), (eg[0], exact, st["steps"])


def test_7up7down_sector_more_blocks_than_one_lds_image_free_fermions():
    """6,009,350,400 states of BASELINE config 5's lattice: 77520 blocks of 77520 positions, 48 GB per vector.  A workgroup of the coupling
    kernel owns 2423 blocks, more than one LDS image of their coupling lists holds and more than 16-bit places number: until round 5 such
    sectors fell back to the fused block-order kernel (337 ms per step); now the product-basis form walks every workgroup's range in rounds
    (k_pb_down, PbDownArgs::rounds; 186 ms per step).  Exact free-fermion energy."""
    L = 20
    hop = square(4, 5, -1.0, pbc=True)
    lev = np.sort(np.linalg.eigvalsh(hop))
    exact = 2 * lev[:7].sum()
    with LanczosEngine(max_steps=300, eps=1e-11, save_vectors=0) as e:
        e.setup_hubbard_onthefly(L, 7, 7, hop, np.zeros(L))
        assert e.rows() == 77520 * 77520 == 6009350400
        lay = e.layout()
        assert lay["kernel"] == 4 and (lay["segments"], lay["pieces"]) == (32, 14) and lay["coupling_parts"] == 1 and lay["coupling_rounds"] == 8, lay
        eg, _, st = e.lanczos(1, want_vectors=False)
    assert abs(eg[0] - exact) <= E_TOL * abs(exact), (eg[0], exact, st["steps"])


@pytest.mark.parametrize("workload", ["hubbard_chain_L12_half_filling_U4", "heisenberg_chain_L24_sz0_obc", "tj_chain_L12_5up5down_complex", "hubbard_chain_L14_complex_U4",
                                      "hubbard_4x4_6up6down_complex_U4"])  # the last one: complex hoppings in the product-basis layout, chained step
def test_bench_line_of_the_other_model_families(workload):
    """`bench.py --workload W` prints its one JSON line for every model family (a Hubbard-only assumption in the attempt loop
    once broke the Heisenberg and t-J lines while the default line stayed green)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", workload, "--steps", "5", "--warmup", "1", "--no-cpu-baseline",
                          "--no-generic-csr", "--no-reortho-leg"], capture_output=True, text=True, timeout=400)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["config"]["workload"] == workload and line["steps"] == 5 and line["value"] > 0
    assert line["roofline"]["achieved"] > 0 and 0 < line["roofline"]["frac"] <= 1.0
    if workload == "hubbard_4x4_6up6down_complex_U4":
        assert line["config"]["layout"]["kernel"] == "product" and line["config"]["layout"]["chained_step"]
