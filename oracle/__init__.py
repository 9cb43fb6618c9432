"""ctypes binding of the CPU oracle (oracle/lpp_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (lanczosplusplus_amd) never imports this.
"parity unpinned" caveats: see the header of lpp_oracle.c.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblpp_oracle.so")


def build(force=False):
    """Compile the C restatement (and oracle/_ref when /root/reference is present)."""
    src = os.path.join(_HERE, "lpp_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liblpp_oracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None

_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


class LanczosParams(C.Structure):
    _fields_ = [("max_steps", C.c_int), ("min_steps", C.c_int), ("eps", C.c_double), ("reortho", C.c_int)]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        build()
    L = C.CDLL(_SO)
    L.lppo_comb.restype = C.c_uint64
    L.lppo_comb.argtypes = [C.c_int, C.c_int]
    L.lppo_onespin_size.restype = C.c_int64
    L.lppo_onespin_size.argtypes = [C.c_int, C.c_int]
    L.lppo_onespin_fill.argtypes = [C.c_int, C.c_int, _u64p]
    L.lppo_onespin_rank.restype = C.c_int64
    L.lppo_onespin_rank.argtypes = [C.c_uint64]
    L.lppo_hubbard_size.restype = C.c_int64
    L.lppo_hubbard_size.argtypes = [C.c_int] * 3
    L.lppo_hubbard_basis_words.argtypes = [C.c_int] * 3 + [_u64p, _u64p]
    L.lppo_hubbard_perfect_index.restype = C.c_int64
    L.lppo_hubbard_perfect_index.argtypes = [C.c_int] * 3 + [C.c_uint64] * 2
    L.lppo_hubbard_setup.restype = C.c_void_p
    L.lppo_hubbard_setup.argtypes = [C.c_int] * 3 + [C.c_void_p] * 5 + [C.c_int]
    L.lppo_hubbard_setup_super.restype = C.c_void_p
    L.lppo_hubbard_setup_super.argtypes = [C.c_int] * 3 + [C.c_void_p] * 6 + [C.c_int]
    L.lppo_hubbard_setup_time.restype = C.c_void_p
    L.lppo_hubbard_setup_time.argtypes = [C.c_int] * 3 + [C.c_void_p] * 6 + [C.c_int, C.c_void_p, C.c_double]
    L.lppo_hubbard_otf_mvp.argtypes = [C.c_int] * 3 + [_f64p, _f64p, _f64p, _f64p, _f64p, C.c_int64, C.c_int64, C.c_int]
    L.lppo_hubbard_otf_new.restype = C.c_void_p
    L.lppo_hubbard_otf_new.argtypes = [C.c_int] * 3 + [_f64p, _f64p, _f64p, C.c_int]
    L.lppo_hubbard_otf_free.argtypes = [C.c_void_p]
    L.lppo_hubbard_otf_rows.restype = C.c_int64
    L.lppo_hubbard_otf_rows.argtypes = [C.c_void_p]
    L.lppo_hubbard_otf_apply.argtypes = [C.c_void_p, _f64p, _f64p, C.c_int64, C.c_int64]
    L.lppo_hubbard_otf_lanczos.restype = C.c_int
    L.lppo_hubbard_otf_lanczos.argtypes = [C.c_void_p, _f64p, C.POINTER(LanczosParams), _f64p, _f64p, _f64p]
    L.lppo_heis_bits.restype = C.c_int
    L.lppo_heis_bits.argtypes = [C.c_int]
    L.lppo_heis_basis.restype = C.c_int64
    L.lppo_heis_basis.argtypes = [C.c_int] * 3 + [C.c_void_p]
    L.lppo_find_linear.restype = C.c_int64
    L.lppo_find_linear.argtypes = [_u64p, C.c_int64, C.c_uint64]
    L.lppo_find_bisect.restype = C.c_int64
    L.lppo_find_bisect.argtypes = [_u64p, C.c_int64, C.c_uint64]
    L.lppo_heis_setup.restype = C.c_void_p
    L.lppo_heis_setup.argtypes = [C.c_int] * 3 + [_f64p, _f64p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
    L.lppo_tj_basis.restype = C.c_int64
    L.lppo_tj_basis.argtypes = [C.c_int] * 3 + [C.c_void_p]
    L.lppo_tj_perfect_index_literal.restype = C.c_int64
    L.lppo_tj_perfect_index_literal.argtypes = [_u64p, C.c_int64, C.c_int, C.c_uint64, C.c_uint64]
    L.lppo_tj_setup.restype = C.c_void_p
    L.lppo_tj_setup.argtypes = [C.c_int] * 3 + [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_int]
    for f in ("nrows", "nnz"):
        getattr(L, "lppo_csr_" + f).restype = C.c_int64
        getattr(L, "lppo_csr_" + f).argtypes = [C.c_void_p]
    L.lppo_csr_is_complex.restype = C.c_int
    L.lppo_csr_is_complex.argtypes = [C.c_void_p]
    for f in ("rowptr", "colind", "values"):
        getattr(L, "lppo_csr_" + f).restype = C.c_void_p
        getattr(L, "lppo_csr_" + f).argtypes = [C.c_void_p]
    L.lppo_csr_free.argtypes = [C.c_void_p]
    L.lppo_spmv_acc.argtypes = [C.c_int64, _i64p, _i32p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    L.lppo_fill_random.argtypes = [C.c_void_p, C.c_int64, C.c_uint64]
    L.lppo_tridiag_eig.restype = C.c_int
    L.lppo_tridiag_eig.argtypes = [C.c_int, _f64p, _f64p, _f64p, C.c_void_p]
    L.lppo_lanczos_decomposition.restype = C.c_int
    L.lppo_lanczos_decomposition.argtypes = [C.c_int64, _i64p, _i32p, C.c_void_p, C.c_int, C.c_void_p,
                                             C.POINTER(LanczosParams), _f64p, _f64p, C.c_void_p, C.c_void_p, C.c_int]
    L.lppo_lanczos_solve.restype = C.c_int
    L.lppo_lanczos_solve.argtypes = [C.c_int64, _i64p, _i32p, C.c_void_p, C.c_int, C.c_void_p,
                                     C.POINTER(LanczosParams), C.c_int, _f64p, C.c_void_p, C.c_int]
    _lib = L
    return L


class Csr:
    """Host CSR (int64 rowptr, int32 colind, float64|complex128 values) copied out of the oracle."""

    def __init__(self, rowptr, colind, values):
        self.rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        self.colind = np.ascontiguousarray(colind, dtype=np.int32)
        self.values = np.ascontiguousarray(values)
        self.nrows = len(self.rowptr) - 1
        self.nnz = int(self.rowptr[-1])
        self.is_complex = np.iscomplexobj(self.values)

    def to_scipy(self):
        import scipy.sparse as sp
        return sp.csr_matrix((self.values, self.colind.astype(np.int64), self.rowptr), shape=(self.nrows, self.nrows))


def _take_csr(handle):
    L = lib()
    n, nnz = L.lppo_csr_nrows(handle), L.lppo_csr_nnz(handle)
    cplx = L.lppo_csr_is_complex(handle)
    rp = np.ctypeslib.as_array(C.cast(L.lppo_csr_rowptr(handle), C.POINTER(C.c_int64)), shape=(n + 1,)).copy()
    if nnz > 0:
        ci = np.ctypeslib.as_array(C.cast(L.lppo_csr_colind(handle), C.POINTER(C.c_int32)), shape=(nnz,)).copy()
        nv = nnz * (2 if cplx else 1)
        va = np.ctypeslib.as_array(C.cast(L.lppo_csr_values(handle), C.POINTER(C.c_double)), shape=(nv,)).copy()
    else:
        ci = np.zeros(0, np.int32)
        va = np.zeros(0, np.float64)
    if cplx:
        va = va.view(np.complex128)
    L.lppo_csr_free(handle)
    return Csr(rp, ci, va)


def _mat(a, L):
    if a is None:
        return None
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(L, L))
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def onespin_basis(nsite, npart):
    n = lib().lppo_onespin_size(nsite, npart)
    out = np.zeros(n, np.uint64)
    lib().lppo_onespin_fill(nsite, npart, out)
    return out


def onespin_rank(state):
    return lib().lppo_onespin_rank(int(state))


def hubbard_basis_words(L, nup, ndown):
    n = lib().lppo_hubbard_size(L, nup, ndown)
    up, dn = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
    lib().lppo_hubbard_basis_words(L, nup, ndown, up, dn)
    return up, dn


def hubbard_csr(L, nup, ndown, hop, U, V=None, ninj=None, jcoup=None, potentialT=None, timeFactor=0.0):
    """HubbardHelper::setupHamiltonian.  ninj: Coulomb coupling (HubbardOneBandExtended); jcoup: spin coupling, with ninj
    Model=SuperHubbardExtended; Model=KaneMeleHubbard is hop = term 0 + term 1 (HubbardHelper.h:63-66), summed by the caller;
    potentialT, timeFactor: the time-dependent potential of HubbardHelper.h:181-182."""
    hop = np.asarray(hop).reshape(L, L)
    cplx = np.iscomplexobj(hop) and np.any(hop.imag != 0)
    hr = _mat(hop.real, L)
    hi = _mat(hop.imag, L) if cplx else None
    U = np.ascontiguousarray(U, np.float64)
    V = np.zeros(L) if V is None else np.ascontiguousarray(np.asarray(V, np.float64)[:L])
    nj = _mat(ninj, L)
    jc = _mat(jcoup, L)
    pt = None if potentialT is None else np.ascontiguousarray(np.asarray(potentialT, np.float64)[:L])
    h = lib().lppo_hubbard_setup_time(L, nup, ndown, _ptr(hr), _ptr(hi), _ptr(U), _ptr(V), _ptr(nj), _ptr(jc), int(cplx), _ptr(pt), float(timeFactor))
    return _take_csr(h)


def hubbard_otf_mvp(L, nup, ndown, hop, U, V, x, y, row0=0, row1=0, nthreads=0):
    hr = _mat(np.asarray(hop).real, L)
    lib().lppo_hubbard_otf_mvp(L, nup, ndown, hr, np.ascontiguousarray(U, np.float64),
                               np.ascontiguousarray(np.asarray(V, np.float64)[:L]), x, y, row0, row1, nthreads)


class HubbardOtf:
    """Tabulated form of hubbard_otf_mvp (bit-identical results, see lpp_oracle.c) + the Lanczos loop over it:
    the reference's SolverOptions=InternalProductOnTheFly run, fast enough for BASELINE config 2 on a few cores."""

    def __init__(self, L, nup, ndown, hop, U, V=None, nthreads=0):
        hr = _mat(np.asarray(hop).real, L)
        U = np.ascontiguousarray(U, np.float64)
        V = np.zeros(L) if V is None else np.ascontiguousarray(np.asarray(V, np.float64)[:L])
        self._h = lib().lppo_hubbard_otf_new(L, nup, ndown, hr, U, V, nthreads)
        self.nrows = lib().lppo_hubbard_otf_rows(self._h)

    def close(self):
        if self._h:
            lib().lppo_hubbard_otf_free(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def apply(self, x, y, row0=0, row1=0):
        """x += H y on rows [row0, row1) (0, 0 = all)."""
        lib().lppo_hubbard_otf_apply(self._h, x, y, row0, row1)
        return x

    def lanczos(self, init, max_steps=200, min_steps=4, eps=1e-12):
        prm = LanczosParams(max_steps, min_steps, eps, 0)
        ms = min(max_steps, self.nrows)
        a, b, hist = np.zeros(ms + 1), np.zeros(ms + 1), np.zeros(ms + 1)
        steps = lib().lppo_hubbard_otf_lanczos(self._h, init, C.byref(prm), a, b, hist)
        return steps, a[:steps], b[:steps], hist[:steps]


def heis_basis(L, twiceS, szPlusConst):
    n = lib().lppo_heis_basis(L, twiceS, szPlusConst, None)
    out = np.zeros(n, np.uint64)
    if n:
        lib().lppo_heis_basis(L, twiceS, szPlusConst, out.ctypes.data_as(C.c_void_p))
    return out


def heis_csr(L, twiceS, szPlusConst, jpm, jzz, field=None, aniso=None, literal_index=False):
    f = None if field is None else np.ascontiguousarray(field, np.float64)
    a = None if aniso is None else np.ascontiguousarray(aniso, np.float64)
    h = lib().lppo_heis_setup(L, twiceS, szPlusConst, _mat(jpm, L), _mat(jzz, L), _ptr(f), 0 if f is None else len(f),
                              _ptr(a), 0 if a is None else len(a), int(literal_index))
    return _take_csr(h)


def tj_basis(L, nup, ndown):
    n = lib().lppo_tj_basis(L, nup, ndown, None)
    out = np.zeros(n, np.uint64)
    if n:
        lib().lppo_tj_basis(L, nup, ndown, out.ctypes.data_as(C.c_void_p))
    return out


def tj_csr(L, nup, ndown, hop, jpm, jzz, w, potentialV=None, force_complex=False, literal_index=False):
    hop = np.asarray(hop).reshape(L, L)
    cplx = force_complex or (np.iscomplexobj(hop) and np.any(hop.imag != 0))
    hr = _mat(hop.real, L)
    hi = _mat(hop.imag, L) if np.iscomplexobj(hop) else None
    pv = None if potentialV is None else np.ascontiguousarray(potentialV, np.float64)
    assert pv is None or len(pv) == 2 * L
    h = lib().lppo_tj_setup(L, nup, ndown, _ptr(hr), _ptr(hi), _ptr(_mat(jpm, L)), _ptr(_mat(jzz, L)), _ptr(_mat(w, L)),
                            _ptr(pv), 0 if pv is None else len(pv), int(cplx), int(literal_index))
    return _take_csr(h)


def spmv_acc(csr, x, y, nthreads=1):
    """x += A y in place (x, y numpy arrays of csr dtype)."""
    assert x.dtype == csr.values.dtype and y.dtype == csr.values.dtype
    lib().lppo_spmv_acc(csr.nrows, csr.rowptr, csr.colind, csr.values.ctypes.data_as(C.c_void_p), int(csr.is_complex),
                        x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), nthreads)
    return x


def fill_random(n, seed, is_complex=False):
    v = np.zeros(n, np.complex128 if is_complex else np.float64)
    lib().lppo_fill_random(v.ctypes.data_as(C.c_void_p), n * (2 if is_complex else 1), seed)
    return v


def tridiag_eig(d, e, vectors=False):
    n = len(d)
    d = np.ascontiguousarray(d, np.float64)
    e2 = np.zeros(max(n, 1), np.float64)
    e2[:max(n - 1, 0)] = np.asarray(e, np.float64)[:max(n - 1, 0)]
    w = np.zeros(n)
    z = np.zeros((n, n)) if vectors else None
    rc = lib().lppo_tridiag_eig(n, d, e2, w, _ptr(z))
    if rc != 0:
        raise RuntimeError("tridiag_eig failed")
    return (w, z) if vectors else w


def lanczos_decomposition(csr, init, max_steps=200, min_steps=4, eps=1e-12, reortho=False, keep_vectors=False,
                          nthreads=1):
    prm = LanczosParams(max_steps, min_steps, eps, int(reortho))
    ms = min(max_steps, csr.nrows)
    a, b = np.zeros(ms + 1), np.zeros(ms + 1)
    V = np.zeros((ms, csr.nrows), csr.values.dtype) if (keep_vectors or reortho) else None
    hist = np.zeros(ms + 1)
    steps = lib().lppo_lanczos_decomposition(csr.nrows, csr.rowptr, csr.colind, csr.values.ctypes.data_as(C.c_void_p),
                                             int(csr.is_complex), init.ctypes.data_as(C.c_void_p), C.byref(prm), a, b,
                                             _ptr(V), hist.ctypes.data_as(C.c_void_p), nthreads)
    return steps, a[:steps], b[:steps], (V[:steps] if V is not None else None), hist[:steps]


def lanczos_solve(csr, init, nstates=1, max_steps=200, min_steps=4, eps=1e-12, reortho=False, want_vectors=True,
                  nthreads=1):
    prm = LanczosParams(max_steps, min_steps, eps, int(reortho))
    eigs = np.zeros(nstates)
    zs = np.zeros((nstates, csr.nrows), csr.values.dtype) if want_vectors else None
    steps = lib().lppo_lanczos_solve(csr.nrows, csr.rowptr, csr.colind, csr.values.ctypes.data_as(C.c_void_p),
                                     int(csr.is_complex), init.ctypes.data_as(C.c_void_p), C.byref(prm), nstates, eigs,
                                     _ptr(zs), nthreads)
    if steps < 0:
        raise RuntimeError("oracle lanczos failed")
    return eigs, zs, steps
