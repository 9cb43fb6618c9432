"""The C++ host shim (lanczosplusplus_amd/host): host assembly through the reference-named model classes is
bit-identical to the oracle for every input file; the `lanczos` driver prints the reference's "Energy=" line."""
import os
import re
import subprocess

import numpy as np
import pytest

import oracle
from lanczosplusplus_amd import geometry

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "lanczosplusplus_amd", "host")
GOLD = os.path.join(ROOT, "tests", "golden")


def _oracle_csr(path):
    inp = geometry.parse_input(open(path).read())
    L = int(inp["TotalNumberOfSites"])
    terms = geometry.terms_from_input(inp)
    model = inp["Model"]
    if model in ("HubbardOneBand", "HubbardOneBandExtended", "SuperHubbardExtended", "KaneMeleHubbard"):
        # ModelSelector.h:76-80; HubbardHelper.h:39-66: Coulomb coupling = term 1, spin coupling = term 2, Kane-Mele hoppings = term 0 + term 1
        hop = terms[0] + terms[1] if model == "KaneMeleHubbard" else terms[0]
        return oracle.hubbard_csr(L, int(inp["TargetElectronsUp"]), int(inp["TargetElectronsDown"]), hop,
                                  inp["hubbardU"], inp["potentialV"], potentialT=inp.get("PotentialT"), timeFactor=float(inp.get("timeFactor", 0.0)),
                                  ninj=(terms[1] if model in ("HubbardOneBandExtended", "SuperHubbardExtended") else None),
                                  jcoup=(terms[2] if model == "SuperHubbardExtended" else None))
    if model == "Heisenberg":
        return oracle.heis_csr(L, int(inp["HeisenbergTwiceS"]), int(inp["TargetSzPlusConst"]), terms[0], terms[1],
                               field=inp.get("MagneticField"), aniso=inp.get("AnisotropyD"))
    if model == "TjMultiOrb":
        return oracle.tj_csr(L, int(inp["TargetElectronsUp"]), int(inp["TargetElectronsDown"]), terms[0], terms[1], terms[2],
                             terms[3], potentialV=inp.get("potentialV"),
                             force_complex="useComplex" in inp.get("SolverOptions", ""))
    raise ValueError(model)


def _read_dump(path):
    with open(path, "rb") as f:
        n, nnz, cplx = (int(x) for x in f.readline().split())
        rp = np.frombuffer(f.read(8 * (n + 1)), np.int64)
        ci = np.frombuffer(f.read(4 * nnz), np.int32)
        va = np.frombuffer(f.read(8 * nnz * (2 if cplx else 1)), np.float64)
    return rp, ci, va.view(np.complex128) if cplx else va


@pytest.mark.parametrize("name", ["input0.inp", "hubbard_ladder_2x4.inp", "heisenberg_chain_L12.inp",
                                  "tj_chain_L8_complex.inp", "hubbard_chain_L12.inp", "hubbard_extended_2x4.inp",
                                  "heisenberg_spin1_L8.inp", "heisenberg_spin32_L6.inp", "super_hubbard_2x4.inp",
                                  "kane_mele_hubbard_chain_L8.inp", "hubbard_ladder_2x4_potentialT.inp"])
def test_host_assembly_bit_exact(name, tmp_path):
    exe = os.path.join(HOST, "dump_csr")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    out = str(tmp_path / "csr.bin")
    subprocess.check_call([exe, os.path.join(GOLD, name), out])
    rp, ci, va = _read_dump(out)
    A = _oracle_csr(os.path.join(GOLD, name))
    assert np.array_equal(rp, A.rowptr) and np.array_equal(ci, A.colind)
    assert va.dtype == A.values.dtype
    assert np.array_equal(va.view(np.uint64), A.values.view(np.uint64))  # bit patterns, signed zeros included


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["input0.inp", "hubbard_ladder_2x4.inp", "hubbard_ladder_2x4_onthefly.inp", "heisenberg_chain_L12.inp",
                                  "tj_chain_L8_complex.inp", "hubbard_chain_L12.inp", "hubbard_extended_2x4.inp",
                                  "hubbard_extended_2x4_onthefly.inp", "heisenberg_spin1_L8.inp", "heisenberg_spin32_L6.inp",
                                  "super_hubbard_2x4.inp", "super_hubbard_2x4_onthefly.inp", "kane_mele_hubbard_chain_L8.inp",
                                  "hubbard_ladder_2x4_potentialT.inp"])
def test_lanczos_driver_prints_reference_energy_line(name):
    # hubbard_chain_L12.inp is BASELINE config 1 (853,776 states): host assembly, upload with the N_up = 924 row-block
    # hint, i.e. the LDS-window kernel with the block template on an UPLOADED matrix
    exe = os.path.join(HOST, "lanczos")
    assert os.path.exists(exe)
    res = subprocess.run([exe, "-f", os.path.join(GOLD, name), "-p", "12"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    m = re.search(r"^Energy=(\S+)$", res.stdout, re.M)
    assert m, res.stdout
    e = float(m.group(1))
    A = _oracle_csr(os.path.join(GOLD, name))
    D = A.to_scipy().toarray() if A.nrows <= 5000 else None
    # dense check where the matrix is Hermitian (the reference's S >= 3/2 Heisenberg matrix is not: Heisenberg.h:296-303)
    hermitian = D is not None and np.array_equal(D, D.conj().T)
    e0 = np.linalg.eigvalsh(D)[0] if hermitian else None
    eo, _, _ = oracle.lanczos_solve(A, oracle.fill_random(A.nrows, 1234, A.is_complex), want_vectors=False)
    assert abs(e - eo[0]) <= (1e-10 if (D is None or hermitian) else 1e-7) * abs(eo[0])
    if e0 is not None:
        assert abs(e - e0) <= 1e-10 * abs(e0)
    if name == "input0.inp":
        assert abs(e + 2 * np.sqrt(5)) < 1e-10
    assert re.search(r"^E\[0\]=\S+ norm=\S+$", res.stdout, re.M)


def _with_lines(name, tmp_path, extra):
    """copy of a golden input with extra `Label=value` lines appended"""
    txt = open(os.path.join(GOLD, name)).read()
    out = tmp_path / ("mod_" + name)
    out.write_text(txt.rstrip("\n") + "\n" + "\n".join(extra) + "\n")
    return str(out)


def _boundary(path, mode):
    exe = os.path.join(HOST, "test_boundary")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    res = subprocess.run([exe, "-f", path, "-m", mode, "-p", "14"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    return res


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["hubbard_ladder_2x4.inp", "tj_chain_L8_complex.inp"])
def test_engine_loops_over_symmetry_sectors(name):
    """Engine.h:616-652 with a symmetry class of three sectors (H+3, H, empty): specialSymmetrySector(p) must make sector p's
    matrix the resident one (InternalProductStored.h:124); the lowest sector wins and rs.transform embeds its vector."""
    path = os.path.join(GOLD, name)
    A = _oracle_csr(path)
    e0 = np.linalg.eigvalsh(A.to_scipy().toarray())[0]
    res = _boundary(path, "sectors")
    e = float(re.search(r"^Energy=(\S+)$", res.stdout, re.M).group(1))
    assert abs(e - e0) <= 1e-10 * abs(e0), (e, e0, "an engine stuck on sector 0 reports E0 + 3")
    m = re.search(r"^Sector=(\d+) Length=(\d+) NormInside=(\S+) NormOutside=(\S+)$", res.stdout, re.M)
    assert m and int(m.group(1)) == 1 and int(m.group(2)) == 2 * A.nrows
    assert abs(float(m.group(3)) - 1) < 1e-8 and float(m.group(4)) == 0
    # both non-empty sectors were solved: two E[0]= lines would be wrong (printEnergiesAndNorms runs once), one is right
    assert len(re.findall(r"^E\[0\]=", res.stdout, re.M)) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["input0.inp", "tj_chain_L8_complex.inp"])
def test_lanczos_failure_falls_back_to_full_diagonalisation(name, tmp_path):
    """Engine.h:625-639: when the solver throws (here: Excited=2 but LanczosSteps=2, fewer steps than states) the engine
    catches and calls hamiltonian.fullDiag (DefaultSymmetry.h:80-93); the levels are then the dense spectrum."""
    path = _with_lines(name, tmp_path, ["LanczosSteps=2", "Excited=2"])
    A = _oracle_csr(os.path.join(GOLD, name))
    dense = np.linalg.eigvalsh(A.to_scipy().toarray())
    res = _boundary(path, "fulldiag")
    assert "trying exact diagonalization" in res.stderr
    assert re.search(r"^UsedFullDiag=1$", res.stdout, re.M)
    lev = [float(re.search(r"^Level%d=(\S+)$" % k, res.stdout, re.M).group(1)) for k in range(3)]
    assert np.abs(np.array(lev) - dense[:3]).max() <= 1e-10 * np.abs(dense[:3]).max()
    norms = [float(x) for x in re.findall(r"^E\[\d\]=\S+ norm=(\S+)$", res.stdout, re.M)]
    assert len(norms) == 3 and all(abs(v - 1) < 1e-10 for v in norms)
    # without the failure the same binary does not touch fullDiag
    res = _boundary(os.path.join(GOLD, name), "fulldiag")
    assert re.search(r"^UsedFullDiag=0$", res.stdout, re.M)


@pytest.mark.gpu
def test_reference_constructors_honour_the_solver_parameters(tmp_path):
    """The reference's (model, rs) constructor + LanczosSolver(hamiltonian, params): LanczosSteps=20 must bound the
    decomposition (the engine is created before the parameters are known: lpp_engine_set_solver), eps is honoured."""
    name = "hubbard_ladder_2x4.inp"
    A = _oracle_csr(os.path.join(GOLD, name))
    init = oracle.fill_random(A.nrows, 1234)
    for steps, eps in ((20, 0.0), (200, 1e-12)):
        path = _with_lines(name, tmp_path, ["LanczosSteps=%d" % steps, "LanczosEps=%g" % eps])
        res = _boundary(path, "decomp")
        n = int(re.search(r"^Steps=(\d+)$", res.stdout, re.M).group(1))
        ab = np.array([[float(x) for x in ln.split()[2:4]] for ln in res.stdout.splitlines() if ln.startswith("ab ")])
        so, ao, bo, _, _ = oracle.lanczos_decomposition(A, init, max_steps=steps, eps=eps)
        assert n == so == len(ab) and (n == 20 if eps == 0.0 else n < 200)
        assert np.abs(ab[:, 0] - ao).max() < 1e-8 and np.abs(ab[:, 1] - bo).max() < 1e-8


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["hubbard_ladder_2x4.inp", "hubbard_extended_2x4_onthefly.inp", "hubbard_ladder_2x4_complex.inp", "heisenberg_chain_L12.inp",
                                  "tj_chain_L8_complex.inp"])
def test_partitioned_driver_over_rccl_at_world_size_one(name):
    """lanczos -P: one process per GPU through liblpp_comm_rccl.so (no Python in the loop).  A one-GPU box hosts one rank (RCCL refuses two
    ranks on one device: scripts/experiments/r05_rccl_same_gpu.py): the communicator is created (ncclCommInitRank), the rank's rows are
    assembled on the device, the energy line is the oracle's.  The third input says SolverOptions=useComplex (lanczos.cpp:194-226): complex
    vectors and a c128 communicator (round 5).  The last two are not of the Hubbard family: the model assembles its CSR on the host
    (DefaultSymmetry.h:54-57) and the rank's rows go through lpp_engine_set_csr_partition (Heisenberg real, t-J complex)."""
    exe = os.path.join(HOST, "lanczos")
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    res = subprocess.run([exe, "-f", os.path.join(GOLD, name), "-p", "12", "-P"], capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stderr
    e = float(re.search(r"^Energy=(\S+)$", res.stdout, re.M).group(1))
    A = _oracle_csr(os.path.join(GOLD, name))
    e0 = np.linalg.eigvalsh(A.to_scipy().toarray())[0]
    assert abs(e - e0) <= 1e-10 * abs(e0)
    assert "ranks=1" in res.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("name,env,marker", [
    ("tj_chain_L12_complex.inp", {"LPP_TJ_LAYOUT": "1"}, "t-J hole-major form"),
    ("heisenberg_chain_L16.inp", {"LPP_PRODUCT_LAYOUT": "1", "LPP_PB_PIECE_ROWS": "256"}, "chain layout against the assembler's row walk"),
])
def test_driver_describes_the_model_and_the_engine_takes_its_structured_form(name, env, marker):
    """The reference's hand-over is a CSR (DefaultSymmetry.h:54-57 -> InternalProductStored.h:116).  Round 5: the shim also tells the engine which
    model that CSR belongs to (lpp_engine_set_model_*); the engine regenerates the matrix from the description, finds it identical to the
    handed-over one bit for bit, and holds the model without a stored matrix (t-J: hole-major form; spin chain: one block of the segmented
    form).  Same `lanczos -f` command line, same energy line; the forms are forced onto these small inputs, LPP_VERBOSE says which one ran."""
    exe = os.path.join(HOST, "lanczos")
    res = subprocess.run([exe, "-f", os.path.join(GOLD, name), "-p", "12"], capture_output=True, text=True, timeout=300, env=dict(os.environ, LPP_VERBOSE="1", **env))
    assert res.returncode == 0, res.stderr
    assert "the model description regenerates the handed-over CSR" in res.stderr and marker in res.stderr, res.stderr[-3000:]
    e = float(re.search(r"^Energy=(\S+)$", res.stdout, re.M).group(1))
    A = _oracle_csr(os.path.join(GOLD, name))
    eo, _, _ = oracle.lanczos_solve(A, oracle.fill_random(A.nrows, 1234, A.is_complex), want_vectors=False)
    assert abs(e - eo[0]) <= 1e-10 * abs(eo[0])
    # without the switch that forces the form onto a small input the same command takes the general layout: same line
    res0 = subprocess.run([exe, "-f", os.path.join(GOLD, name), "-p", "12"], capture_output=True, text=True, timeout=300, env=dict(os.environ, LPP_VERBOSE="1"))
    assert res0.returncode == 0 and marker not in res0.stderr
    assert abs(float(re.search(r"^Energy=(\S+)$", res0.stdout, re.M).group(1)) - e) <= 1e-10 * abs(e)
