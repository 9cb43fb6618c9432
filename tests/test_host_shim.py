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
    if model == "HubbardOneBand":
        return oracle.hubbard_csr(L, int(inp["TargetElectronsUp"]), int(inp["TargetElectronsDown"]), terms[0],
                                  inp["hubbardU"], inp["potentialV"])
    if model == "Heisenberg":
        return oracle.heis_csr(L, int(inp["HeisenbergTwiceS"]), int(inp["TargetSzPlusConst"]), terms[0], terms[1],
                               field=inp.get("MagneticField"))
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
                                  "tj_chain_L8_complex.inp", "hubbard_chain_L12.inp"])
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
                                  "tj_chain_L8_complex.inp", "hubbard_chain_L12.inp"])
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
    e0 = np.linalg.eigvalsh(A.to_scipy().toarray())[0] if A.nrows <= 5000 else None
    eo, _, _ = oracle.lanczos_solve(A, oracle.fill_random(A.nrows, 1234, A.is_complex), want_vectors=False)
    assert abs(e - eo[0]) <= 1e-10 * abs(eo[0])
    if e0 is not None:
        assert abs(e - e0) <= 1e-10 * abs(e0)
    if name == "input0.inp":
        assert abs(e + 2 * np.sqrt(5)) < 1e-10
    assert re.search(r"^E\[0\]=\S+ norm=\S+$", res.stdout, re.M)
