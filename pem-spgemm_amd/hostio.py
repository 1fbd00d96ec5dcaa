"""ctypes view of libpemhost.so (include/pem_host.h): the pemspgemm tool's Matrix-Market
reader and result/CSV writers.  Host-only; no GPU needed."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpemhost.so")
CLI_PATH = os.path.join(_HERE, "pemspgemm")

HOST_SYMBOLS = ["pem_mm_read", "pem_coo_free", "pem_host_last_error", "pem_write_result_files", "pem_csv_append", "pem_write_mtx_csr"]


class _Coo(C.Structure):
    _fields_ = [("rows", C.c_int32), ("cols", C.c_int32), ("nnz", C.c_int64), ("I", C.POINTER(C.c_int32)), ("J", C.POINTER(C.c_int32)),
                ("V", C.POINTER(C.c_double)), ("symmetric", C.c_int32), ("field", C.c_int32)]


class CsvRecord(C.Structure):
    _fields_ = [("matrix", C.c_char_p), ("flop", C.c_uint64), ("c_nnz", C.c_int64), ("compression_ratio", C.c_double),
                ("a_conversion_kernel_ms", C.c_double), ("b_conversion_kernel_ms", C.c_double), ("total_conversion_ms", C.c_double),
                ("step1_ms", C.c_double), ("step2_ms", C.c_double), ("step3_ms", C.c_double), ("spgemm_ms", C.c_double),
                ("kernel_ms", C.c_double), ("malloc_ms", C.c_double), ("gflops", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run __graft_entry__.build()")
        _lib = C.CDLL(LIB_PATH)
        _lib.pem_host_last_error.restype = C.c_char_p
    return _lib


def mm_read(path, threads=0):
    m = _Coo()
    rc = lib().pem_mm_read(path.encode(), int(threads), C.byref(m))
    if rc != 0:
        raise RuntimeError(f"pem_mm_read -> {rc}: {lib().pem_host_last_error().decode()}")
    n = int(m.nnz)

    def arr(p, dt):
        return np.ctypeslib.as_array(p, shape=(n,)).astype(dt, copy=True) if n else np.zeros(0, dt)

    out = dict(rows=m.rows, cols=m.cols, nnz=n, I=arr(m.I, np.int32), J=arr(m.J, np.int32), V=arr(m.V, np.float64),
               symmetric=bool(m.symmetric), field=m.field)
    lib().pem_coo_free(C.byref(m))
    return out


def write_result_files(directory, rows, cols, vals):
    rows = np.ascontiguousarray(rows, np.int32)
    cols = np.ascontiguousarray(cols, np.int32)
    vals = np.ascontiguousarray(vals, np.float64)
    rc = lib().pem_write_result_files(directory.encode(), C.c_int64(len(rows)), rows.ctypes.data_as(C.POINTER(C.c_int32)),
                                      cols.ctypes.data_as(C.POINTER(C.c_int32)), vals.ctypes.data_as(C.POINTER(C.c_double)))
    if rc != 0:
        raise RuntimeError(lib().pem_host_last_error().decode())


def csv_append(path, extra=None, **fields):
    rec = CsvRecord()
    for k, v in fields.items():
        setattr(rec, k, v.encode() if k == "matrix" else v)
    rc = lib().pem_csv_append(path.encode(), C.byref(rec), extra.encode() if extra else None)
    if rc != 0:
        raise RuntimeError(lib().pem_host_last_error().decode())


def write_mtx_csr(path, rows, cols, rowptr, colidx, vals, comment=""):
    rowptr = np.ascontiguousarray(rowptr, np.int32)
    colidx = np.ascontiguousarray(colidx, np.int32)
    vals = np.ascontiguousarray(vals, np.float64)
    rc = lib().pem_write_mtx_csr(path.encode(), int(rows), int(cols), rowptr.ctypes.data_as(C.POINTER(C.c_int32)),
                                 colidx.ctypes.data_as(C.POINTER(C.c_int32)), vals.ctypes.data_as(C.POINTER(C.c_double)), comment.encode())
    if rc != 0:
        raise RuntimeError(lib().pem_host_last_error().decode())
