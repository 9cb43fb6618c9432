"""The reference's hand-over at BASELINE config 2's size (DefaultSymmetry.h:54-57 -> InternalProductStored.h:116 delivers a host CSR):
the 4x4 half-filled Hubbard matrix -- 165,636,900 rows, 5,819,376,420 non-zeros, 71 GB as int64 row pointers + int32 columns + f64 values --
goes through lpp_engine_set_csr from HOST arrays: PCIe upload, basis-block detection, T / C / D read off the matrix, every row verified
against them, then the solve.  The host arrays are made by reading a device-assembled matrix back (lpp_engine_get_csr; the oracle's
single-threaded host assembly would take ~6 minutes), so what is timed is exactly the boundary.  Prints one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from bench import square_lattice
from lanczosplusplus_amd import LanczosEngine

L, nu, nd = (16, 8, 8) if len(sys.argv) < 2 else tuple(int(v) for v in sys.argv[1:4])
hop, U = square_lattice(4, 4, -1.0) if L == 16 else square_lattice(3, 4, -1.0), np.full(L, 4.0)
out = {}
with LanczosEngine(max_steps=200) as d:
    t0 = time.time()
    d.assemble_hubbard(L, nu, nd, hop, U, np.zeros(2 * L))
    d.sync()
    out["device_assembly_s"] = time.time() - t0
    lay_d = d.layout()
    t0 = time.time()
    rp, ci, va = d.get_csr()
    out["readback_s"] = time.time() - t0
    e_d, _, st_d = d.lanczos(1, want_vectors=False)
nbytes = rp.nbytes + ci.nbytes + va.nbytes
out.update(rows=int(len(rp) - 1), nnz=int(len(ci)), host_csr_GB=nbytes / 1e9)
with LanczosEngine(max_steps=200) as e:
    os.environ["LPP_VERBOSE"] = "1"
    t0 = time.time()
    e.set_csr(rp, ci, va)  # no hint: the basis block is detected
    e.sync()
    out["set_csr_s"] = time.time() - t0
    os.environ.pop("LPP_VERBOSE")
    lay = e.layout()
    t0 = time.time()
    eg, _, st = e.lanczos(1, want_vectors=False)
    out["solve_s"] = time.time() - t0
    ms = e.bench_spmv(2, 10)
out.update(set_csr_GBps=nbytes / 1e9 / out["set_csr_s"], layout_equals_device_assembled=bool(lay == lay_d), kernel=lay["kernel"], resident_GB=lay["resident_bytes"] / 1e9,
           steps=st["steps"], e0=float(eg[0]), e0_device_assembled=float(e_d[0]), steps_device_assembled=st_d["steps"], spmv_ms=ms)
print(json.dumps(out))
