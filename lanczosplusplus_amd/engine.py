"""Host-side mirror of the reference's plug-in surface for the stored-CSR Lanczos path.

`LanczosEngine` plays the roles of `InternalProductStored` (rows(), matrixVectorProduct(x, y):
x += H y; reference src/Engine/InternalProductStored.h:104-132) and of the PsimagLite
`LanczosSolver` the reference's Engine drives (computeAllStatesBelow / decomposition;
src/Engine/Engine.h:626,478).  It is a thin ctypes layer over the C ABI (include/lpp_engine.h);
all arithmetic runs in liblpp_engine.so on the GPU.  numpy arrays are the host buffers.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import LPP_C128, LPP_F64, Comm, Config, Stats, check


def _vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _mat(a, L):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(L, L))


class LanczosEngine:
    """One engine = one GPU (one process per GPU in the multi-GPU path)."""

    def __init__(self, dtype="f64", device=0, max_steps=200, min_steps=4, eps=1e-12, reortho=False,
                 save_vectors=-1, check_lag=2, spmv_kernel=0, time_kernels=False, seed=1234, stream=None,
                 compress_values=-1):
        self._lib = _capi.lib()
        self._h = C.c_void_p()
        self.is_complex = dtype in ("c128", "complex128", np.complex128, LPP_C128)
        self.np_dtype = np.complex128 if self.is_complex else np.float64
        cfg = Config()
        self._lib.lpp_config_default(C.byref(cfg))
        cfg.device = device
        cfg.dtype = LPP_C128 if self.is_complex else LPP_F64
        cfg.max_steps = max_steps
        cfg.min_steps = min_steps
        cfg.eps = eps
        cfg.reortho = int(bool(reortho))
        cfg.save_vectors = int(save_vectors)
        cfg.check_lag = check_lag
        cfg.spmv_kernel = spmv_kernel
        cfg.time_kernels = int(bool(time_kernels))
        cfg.seed = seed
        cfg.stream = stream
        cfg.compress_values = int(compress_values)
        self.max_steps = max_steps
        self._comm_keepalive = None
        check(self._lib.lpp_engine_create(C.byref(self._h), C.byref(cfg)))

    def stream_ptr(self):
        """the HIP stream the engine enqueues on (lpp_engine_stream), as a ctypes void pointer"""
        return C.c_void_p(self._lib.lpp_engine_stream(self._h))

    # ---- lifetime -------------------------------------------------------------------------
    @property
    def closed(self):
        return not (getattr(self, "_h", None) is not None and self._h)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.lpp_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_solver(self, max_steps=200, min_steps=4, eps=1e-12, reortho=False, save_vectors=-1):
        """ParametersForSolver after creation (the reference builds LanczosSolver after the InternalProduct, Engine.h:608-610)."""
        check(self._lib.lpp_engine_set_solver(self._h, int(max_steps), int(min_steps), float(eps), int(bool(reortho)), int(save_vectors)))
        self.max_steps = int(max_steps)

    # ---- the stored Hamiltonian --------------------------------------------------------------
    def set_row_block(self, rows_per_block):
        """Layout hint for the next set_csr: rows_per_block = N_up of the Hubbard product basis (0 = unknown)."""
        check(self._lib.lpp_engine_set_row_block(self._h, int(rows_per_block)))

    def set_model_tj(self, L, nup, ndown, hop, jpm, jzz, w, potentialV=None):
        """describe the model behind the NEXT set_csr (lpp_engine_set_model_tj): a layout hint, verified against the CSR bit for bit"""
        hop = np.asarray(hop).reshape(L, L)
        hr = _mat(hop.real, L)
        hi = _mat(hop.imag, L) if np.iscomplexobj(hop) else None
        pv = None if potentialV is None else np.ascontiguousarray(potentialV, np.float64)
        check(self._lib.lpp_engine_set_model_tj(self._h, L, nup, ndown, _vp(hr), _vp(hi), _vp(_mat(jpm, L)), _vp(_mat(jzz, L)), _vp(_mat(w, L)),
                                                _vp(pv), 0 if pv is None else len(pv)))

    def set_model_heisenberg(self, L, szPlusConst, jpm, jzz, field=None):
        f = None if field is None else np.ascontiguousarray(field, np.float64)
        check(self._lib.lpp_engine_set_model_heisenberg(self._h, L, szPlusConst, _vp(_mat(jpm, L)), _vp(_mat(jzz, L)), _vp(f), 0 if f is None else len(f)))

    def set_csr(self, rowptr, colind, values):
        rowptr = np.ascontiguousarray(rowptr, np.int64)
        colind = np.ascontiguousarray(colind, np.int32)
        values = np.ascontiguousarray(values, self.np_dtype)
        n = len(rowptr) - 1
        if n < 0 or len(colind) != rowptr[-1] or len(values) != rowptr[-1]:
            raise ValueError("inconsistent CSR arrays")
        check(self._lib.lpp_engine_set_csr(self._h, n, _vp(rowptr), _vp(colind), _vp(values)))

    def set_csr_device(self, nrows, d_rowptr, d_colind, d_values):
        """CSR already resident on the engine's GPU: raw device addresses (e.g. torch tensor .data_ptr())."""
        check(self._lib.lpp_engine_set_csr_device(self._h, int(nrows), C.c_void_p(d_rowptr), C.c_void_p(d_colind), C.c_void_p(d_values)))

    def set_csr_partition(self, comm, global_rows, shard_starts, rowptr, colind, values):
        shard_starts = np.ascontiguousarray(shard_starts, np.int64)
        rowptr = np.ascontiguousarray(rowptr, np.int64)
        colind = np.ascontiguousarray(colind, np.int32)
        values = np.ascontiguousarray(values, self.np_dtype)
        self._comm_keepalive = comm
        check(self._lib.lpp_engine_set_csr_partition(self._h, C.byref(comm.struct), global_rows, _vp(shard_starts),
                                                     _vp(rowptr), _vp(colind), _vp(values)))

    def assemble_hubbard(self, L, nup, ndown, hop, U, V=None, comm=None, ninj=None, jcoup=None):
        """ninj: L x L Coulomb coupling of Model=HubbardOneBandExtended (the reference's second geometry term), or None;
        jcoup: L x L spin coupling (third term), with ninj Model=SuperHubbardExtended.  Model=KaneMeleHubbard: pass hop = term 0 + term 1."""
        hop = np.asarray(hop).reshape(L, L)
        hr = _mat(hop.real, L)
        hi = _mat(hop.imag, L) if np.iscomplexobj(hop) else None
        U = np.ascontiguousarray(U, np.float64)
        V = np.zeros(L) if V is None else np.ascontiguousarray(np.asarray(V, np.float64)[:L])
        nj = None if ninj is None else _mat(ninj, L)
        self._comm_keepalive = comm
        cs = C.byref(comm.struct) if comm is not None else None
        jc = None if jcoup is None else _mat(jcoup, L)
        check(self._lib.lpp_engine_assemble_hubbard_super(self._h, cs, L, nup, ndown, _vp(hr), _vp(hi), _vp(U), _vp(V), _vp(nj), _vp(jc)))

    def setup_hubbard_onthefly(self, L, nup, ndown, hop, U, V=None, comm=None, ninj=None, jcoup=None):
        """Matrix-free Hubbard product (InternalProductOnTheFly semantics): nothing but H_up and H_down is stored.
        jcoup (Model=SuperHubbardExtended): the spin-flip terms move both species; the product then re-derives every row from the
        term list (lpp_engine_setup_hubbard_onthefly_super)."""
        hop = np.asarray(hop).reshape(L, L)
        hr = _mat(hop.real, L)
        hi = _mat(hop.imag, L) if np.iscomplexobj(hop) else None
        U = np.ascontiguousarray(U, np.float64)
        V = np.zeros(L) if V is None else np.ascontiguousarray(np.asarray(V, np.float64)[:L])
        nj = None if ninj is None else _mat(ninj, L)
        self._comm_keepalive = comm
        cs = C.byref(comm.struct) if comm is not None else None
        jc = None if jcoup is None else _mat(jcoup, L)
        check(self._lib.lpp_engine_setup_hubbard_onthefly_super(self._h, cs, L, nup, ndown, _vp(hr), _vp(hi), _vp(U), _vp(V), _vp(nj), _vp(jc)))

    def assemble_heisenberg(self, L, szPlusConst, jpm, jzz, field=None, twiceS=1, anisotropy=None):
        """Heisenberg.h:80-114 on the device; twiceS > 1 or an anisotropy take the any-spin assembler (digit basis)."""
        f = None if field is None else np.ascontiguousarray(field, np.float64)
        if twiceS == 1 and anisotropy is None:
            check(self._lib.lpp_engine_assemble_heisenberg(self._h, L, szPlusConst, _vp(_mat(jpm, L)), _vp(_mat(jzz, L)),
                                                           _vp(f), 0 if f is None else len(f)))
            return
        a = None if anisotropy is None else np.ascontiguousarray(anisotropy, np.float64)
        check(self._lib.lpp_engine_assemble_heisenberg_spin(self._h, L, twiceS, szPlusConst, _vp(_mat(jpm, L)), _vp(_mat(jzz, L)),
                                                            _vp(f), 0 if f is None else len(f), _vp(a), 0 if a is None else len(a)))

    def assemble_tj(self, L, nup, ndown, hop, jpm, jzz, w, potentialV=None):
        hop = np.asarray(hop).reshape(L, L)
        hr = _mat(hop.real, L)
        hi = _mat(hop.imag, L) if np.iscomplexobj(hop) else None
        pv = None if potentialV is None else np.ascontiguousarray(potentialV, np.float64)
        check(self._lib.lpp_engine_assemble_tj(self._h, L, nup, ndown, _vp(hr), _vp(hi), _vp(_mat(jpm, L)),
                                               _vp(_mat(jzz, L)), _vp(_mat(w, L)), _vp(pv),
                                               0 if pv is None else len(pv)))

    def get_csr(self, which=0):
        n, nnz = C.c_int64(), C.c_int64()
        check(self._lib.lpp_engine_get_csr(self._h, which, C.byref(n), C.byref(nnz), None, None, None))
        rowptr = np.zeros(n.value + 1, np.int64)
        colind = np.zeros(nnz.value, np.int32)
        values = np.zeros(nnz.value, self.np_dtype)
        check(self._lib.lpp_engine_get_csr(self._h, which, None, None, _vp(rowptr), _vp(colind), _vp(values)))
        return rowptr, colind, values

    def rows(self):
        return self.stats()["nrows"]

    # ---- A1: x += H y -------------------------------------------------------------------------
    def matrixVectorProduct(self, x, y):
        """x += H y on host arrays (InternalProductStored::matrixVectorProduct semantics)."""
        if x.dtype != self.np_dtype or y.dtype != self.np_dtype or not x.flags.c_contiguous or not y.flags.c_contiguous:
            raise ValueError("x and y must be contiguous %s arrays" % self.np_dtype.__name__)
        n = self.rows()
        if len(x) != n or len(y) != n:
            raise ValueError("vector length %d/%d does not match rows() = %d" % (len(x), len(y), n))
        check(self._lib.lpp_engine_spmv_acc(self._h, _vp(x), _vp(y)))
        return x

    spmv_acc = matrixVectorProduct

    # ---- A2/A3: the solve ---------------------------------------------------------------------
    def _init_ptr(self, init):
        if init is None:
            return None, None
        init = np.ascontiguousarray(init, self.np_dtype)
        if len(init) != self.rows():
            raise ValueError("initial vector length does not match rows()")
        return init, _vp(init)

    def computeAllStatesBelow(self, nstates=1, init=None, want_vectors=True):
        """Lowest `nstates` Ritz values (and vectors): LanczosSolver::computeAllStatesBelow."""
        keep, ip = self._init_ptr(init)
        eigs = np.zeros(nstates, np.float64)
        zs = np.zeros((nstates, self.rows()), self.np_dtype) if want_vectors else None
        st = Stats()
        check(self._lib.lpp_engine_lanczos(self._h, ip, nstates, _vp(eigs), _vp(zs), C.byref(st)))
        return eigs, zs, st.as_dict()

    lanczos = computeAllStatesBelow

    def decomposition(self, init=None):
        keep, ip = self._init_ptr(init)
        a = np.zeros(self.max_steps + 2)
        b = np.zeros(self.max_steps + 2)
        n = C.c_int32()
        st = Stats()
        check(self._lib.lpp_engine_decomposition(self._h, ip, C.byref(n), _vp(a), _vp(b), C.byref(st)))
        return a[:n.value].copy(), b[:n.value].copy(), st.as_dict()

    # ---- incremental interface ------------------------------------------------------------------
    def begin(self, init=None):
        keep, ip = self._init_ptr(init)
        check(self._lib.lpp_engine_lanczos_begin(self._h, ip))

    def step(self, nsteps=1):
        check(self._lib.lpp_engine_lanczos_step(self._h, nsteps))

    def sync(self):
        check(self._lib.lpp_engine_sync(self._h))

    def coeffs(self):
        n = C.c_int32()
        a = np.zeros(self.max_steps + 2)
        b = np.zeros(self.max_steps + 2)
        check(self._lib.lpp_engine_lanczos_coeffs(self._h, C.byref(n), _vp(a), _vp(b)))
        return a[:n.value].copy(), b[:n.value].copy()

    def stats(self):
        st = Stats()
        check(self._lib.lpp_engine_get_stats(self._h, C.byref(st)))
        return st.as_dict()

    def layout(self, which=0):
        """HBM layout of the stored matrix (lpp_layout as a dict)."""
        lay = _capi.Layout()
        check(self._lib.lpp_engine_get_layout(self._h, which, C.byref(lay)))
        return lay.as_dict()

    def bench_spmv(self, warmup=3, iters=20):
        ms = C.c_double()
        check(self._lib.lpp_engine_bench_spmv(self._h, warmup, iters, C.byref(ms)))
        return ms.value


# ---- host-only helpers (no GPU) -------------------------------------------------------------------
def partition_rows(nrows, nranks, block=1):
    starts = np.zeros(nranks + 1, np.int64)
    check(_capi.lib().lpp_partition_rows(nrows, nranks, block, _vp(starts)))
    return starts


def split_csr(rank, nranks, shard_starts, shard_stride, rowptr, colind, values):
    """Split a row block (global columns) into local-column and remote-column CSRs."""
    L = _capi.lib()
    shard_starts = np.ascontiguousarray(shard_starts, np.int64)
    rowptr = np.ascontiguousarray(rowptr, np.int64)
    colind = np.ascontiguousarray(colind, np.int32)
    values = np.ascontiguousarray(values)
    local_rows = len(rowptr) - 1
    esz = values.dtype.itemsize
    nl, nr = C.c_int64(), C.c_int64()
    check(L.lpp_split_csr(rank, nranks, _vp(shard_starts), shard_stride, local_rows, _vp(rowptr), _vp(colind),
                          _vp(values), esz, C.byref(nl), C.byref(nr), None, None, None, None, None, None))
    rpl, rpr = np.zeros(local_rows + 1, np.int64), np.zeros(local_rows + 1, np.int64)
    cl, cr = np.zeros(nl.value, np.int32), np.zeros(nr.value, np.int32)
    vl, vr = np.zeros(nl.value, values.dtype), np.zeros(nr.value, values.dtype)
    check(L.lpp_split_csr(rank, nranks, _vp(shard_starts), shard_stride, local_rows, _vp(rowptr), _vp(colind),
                          _vp(values), esz, C.byref(nl), C.byref(nr), _vp(rpl), _vp(cl), _vp(vl), _vp(rpr), _vp(cr),
                          _vp(vr)))
    return (rpl, cl, vl), (rpr, cr, vr)


def tridiag_lowest(d, e, k=1, vectors=False):
    d = np.ascontiguousarray(d, np.float64)
    n = len(d)
    e2 = np.zeros(max(n, 1), np.float64)
    e2[:max(n - 1, 0)] = np.asarray(e, np.float64)[:max(n - 1, 0)]
    w = np.zeros(k)
    z = np.zeros((n, k)) if vectors else None
    check(_capi.lib().lpp_tridiag_lowest(n, _vp(d), _vp(e2), k, _vp(w), _vp(z)))
    return (w, z) if vectors else w
