"""ctypes binding of include/lpp_engine.h (liblpp_engine.so, built in-tree under csrc/).

There is no fallback: if the shared library is missing the import of any compute entry point
raises, and every compute call fails when no HIP device is present.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# LPP_ENGINE_LIB: another build of the same library (scripts/experiments build their variants into a scratch directory, never over this one)
LIB_PATH = os.environ.get("LPP_ENGINE_LIB") or os.path.join(_HERE, "csrc", "liblpp_engine.so")

LPP_ABI_VERSION = 5
LPP_OK, LPP_ERR_INVALID, LPP_ERR_HIP, LPP_ERR_NOMEM, LPP_ERR_NOCONV, LPP_ERR_STATE, LPP_ERR_COMM = range(7)
LPP_F64, LPP_C128 = 0, 1
LPP_SPMV_AUTO, LPP_SPMV_ROWGROUP, LPP_SPMV_SLICED, LPP_SPMV_WINDOW = 0, 1, 2, 3

STATUS_NAMES = {0: "LPP_OK", 1: "LPP_ERR_INVALID", 2: "LPP_ERR_HIP", 3: "LPP_ERR_NOMEM", 4: "LPP_ERR_NOCONV",
                5: "LPP_ERR_STATE", 6: "LPP_ERR_COMM"}


class LppError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("%s: %s" % (STATUS_NAMES.get(status, status), message))
        self.status = status


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("device", C.c_int32), ("dtype", C.c_int32), ("max_steps", C.c_int32),
                ("min_steps", C.c_int32), ("reortho", C.c_int32), ("save_vectors", C.c_int32),
                ("check_lag", C.c_int32), ("spmv_kernel", C.c_int32), ("time_kernels", C.c_int32),
                ("eps", C.c_double), ("seed", C.c_uint64), ("stream", C.c_void_p),
                ("compress_values", C.c_int32), ("reserved", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("steps", C.c_int32), ("steps_enqueued", C.c_int32), ("converged", C.c_int32),
                ("vectors_saved", C.c_int32), ("nrows", C.c_int64), ("nnz", C.c_int64),
                ("seconds_total", C.c_double), ("spmv_ms_total", C.c_double), ("spmv_launches", C.c_int64),
                ("spmv_bytes", C.c_double), ("reortho_ms_total", C.c_double), ("reortho_calls", C.c_int64), ("reortho_columns", C.c_double)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class Layout(C.Structure):
    _fields_ = [("kernel", C.c_int32), ("coded", C.c_int32), ("local16", C.c_int32), ("shared_stride", C.c_int32),
                ("block_template", C.c_int32), ("diagonal_codes", C.c_int32),
                ("nnz", C.c_int64), ("per_row_entries", C.c_int64), ("shared_entries", C.c_int64),
                ("rows_per_block", C.c_int64), ("resident_bytes", C.c_int64), ("stream_bytes", C.c_int64),
                ("pieces", C.c_int32), ("coupling_parts", C.c_int32), ("diagonal_plain", C.c_int32), ("chained_step", C.c_int32),
                ("split_panel", C.c_int32), ("rows_by_list_length", C.c_int32), ("segments", C.c_int32), ("coupling_rounds", C.c_int32)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


CB_VOID = C.CFUNCTYPE(C.c_int32, C.c_void_p)
CB_REDUCE = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32, C.c_int32)
CB_XCHG = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32)


class Comm(C.Structure):
    _fields_ = [("rank", C.c_int32), ("nranks", C.c_int32), ("ctx", C.c_void_p), ("send_buf", C.c_void_p),
                ("gath_buf", C.c_void_p), ("red_buf", C.c_void_p), ("shard_stride", C.c_int64),
                ("red_len", C.c_int32), ("allgather_begin", CB_VOID), ("allgather_end", CB_VOID),
                ("allreduce_sum", CB_REDUCE), ("send2_buf", C.c_void_p), ("recv2_buf", C.c_void_p),
                ("xchg_chunk", C.c_int64), ("exchange_begin", CB_XCHG), ("exchange_end", CB_XCHG)]


# every symbol include/lpp_engine.h declares: (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "lpp_last_error": (C.c_char_p, []),
    "lpp_abi_version": (C.c_int32, []),
    "lpp_xchg_chunk": (C.c_int64, [C.c_int64, C.c_int64, C.c_int32]),
    "lpp_config_default": (None, [C.POINTER(Config)]),
    "lpp_engine_create": (C.c_int32, [C.POINTER(_P), C.POINTER(Config)]),
    "lpp_engine_destroy": (C.c_int32, [_P]),
    "lpp_engine_stream": (C.c_void_p, [_P]),
    "lpp_engine_set_solver": (C.c_int32, [_P, C.c_int32, C.c_int32, C.c_double, C.c_int32, C.c_int32]),
    "lpp_engine_set_row_block": (C.c_int32, [_P, C.c_int64]),
    "lpp_engine_set_csr": (C.c_int32, [_P, C.c_int64, _P, _P, _P]),
    "lpp_engine_set_csr_device": (C.c_int32, [_P, C.c_int64, _P, _P, _P]),
    "lpp_engine_set_csr_partition": (C.c_int32, [_P, C.POINTER(Comm), C.c_int64, _P, _P, _P, _P]),
    "lpp_engine_assemble_hubbard": (C.c_int32, [_P, C.POINTER(Comm), C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P]),
    "lpp_engine_setup_hubbard_onthefly": (C.c_int32, [_P, C.POINTER(Comm), C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P]),
    "lpp_engine_assemble_hubbard_ext": (C.c_int32, [_P, C.POINTER(Comm), C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P]),
    "lpp_engine_assemble_hubbard_super": (C.c_int32, [_P, C.POINTER(Comm), C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P]),
    "lpp_engine_setup_hubbard_onthefly_ext": (C.c_int32, [_P, C.POINTER(Comm), C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P]),
    "lpp_engine_setup_hubbard_onthefly_super": (C.c_int32, [_P, C.POINTER(Comm), C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P]),
    "lpp_engine_assemble_heisenberg": (C.c_int32, [_P, C.c_int32, C.c_int32, _P, _P, _P, C.c_int32]),
    "lpp_engine_assemble_heisenberg_spin": (C.c_int32, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, C.c_int32, _P, C.c_int32]),
    "lpp_engine_assemble_tj": (C.c_int32, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P, C.c_int32]),
    "lpp_engine_set_model_tj": (C.c_int32, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P, C.c_int32]),
    "lpp_engine_set_model_heisenberg": (C.c_int32, [_P, C.c_int32, C.c_int32, _P, _P, _P, C.c_int32]),
    "lpp_engine_get_csr": (C.c_int32, [_P, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _P, _P, _P]),
    "lpp_engine_spmv_acc": (C.c_int32, [_P, _P, _P]),
    "lpp_engine_lanczos": (C.c_int32, [_P, _P, C.c_int32, _P, _P, C.POINTER(Stats)]),
    "lpp_engine_decomposition": (C.c_int32, [_P, _P, C.POINTER(C.c_int32), _P, _P, C.POINTER(Stats)]),
    "lpp_engine_lanczos_begin": (C.c_int32, [_P, _P]),
    "lpp_engine_lanczos_step": (C.c_int32, [_P, C.c_int32]),
    "lpp_engine_sync": (C.c_int32, [_P]),
    "lpp_engine_lanczos_coeffs": (C.c_int32, [_P, C.POINTER(C.c_int32), _P, _P]),
    "lpp_engine_get_stats": (C.c_int32, [_P, C.POINTER(Stats)]),
    "lpp_engine_get_layout": (C.c_int32, [_P, C.c_int32, C.POINTER(Layout)]),
    "lpp_engine_bench_spmv": (C.c_int32, [_P, C.c_int32, C.c_int32, C.POINTER(C.c_double)]),
    "lpp_partition_rows": (C.c_int32, [C.c_int64, C.c_int32, C.c_int64, _P]),
    "lpp_split_csr": (C.c_int32, [C.c_int32, C.c_int32, _P, C.c_int64, C.c_int64, _P, _P, _P, C.c_int32,
                                  C.POINTER(C.c_int64), C.POINTER(C.c_int64), _P, _P, _P, _P, _P, _P]),
    "lpp_tridiag_lowest": (C.c_int32, [C.c_int32, _P, _P, C.c_int32, _P, _P]),
    "lpp_pb_pack_template": (C.c_int32, [C.c_int64, C.c_int64, _P, _P, _P, C.POINTER(C.c_int32), _P, C.POINTER(C.c_int32),
                                         C.POINTER(C.c_int64), _P, _P, _P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int32]),
    "lpp_pb_seg_plan_stats": (C.c_int32, [C.c_int64, _P, _P, _P, C.c_int32, _P, _P]),
    "lpp_tj_plan_stats": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, C.c_int64, _P, _P, _P, C.c_int32, _P]),
}

_lib = None


def lib():
    """Load liblpp_engine.so (raises if it has not been built: there is no Python/CPU fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); the engine has no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if L.lpp_abi_version() != LPP_ABI_VERSION:
        raise ImportError("liblpp_engine.so ABI version mismatch")
    _lib = L
    return L


def check(status):
    if status != LPP_OK:
        msg = lib().lpp_last_error()
        raise LppError(status, msg.decode() if msg else "")
