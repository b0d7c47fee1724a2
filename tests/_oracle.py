"""ctypes binding of the CPU oracle (oracle/libkso_oracle.so).  TEST INFRASTRUCTURE: only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
import ctypes as C
import os
import subprocess

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(REPO, "oracle")
ORACLE_BIN = os.path.join(ORACLE_DIR, "kso_oracle")
REF_BIN = os.path.join(ORACLE_DIR, "_ref", "kmer_scrub_count")


def _lib():
    p = os.path.join(ORACLE_DIR, "libkso_oracle.so")
    if not os.path.exists(p):
        subprocess.run(["make", "-C", ORACLE_DIR, "kso_oracle", "libkso_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    L = C.CDLL(p)
    L.kso_table_new.restype = C.c_void_p
    L.kso_table_new.argtypes = [C.c_uint]
    L.kso_table_free.argtypes = [C.c_void_p]
    L.kso_table_size.argtypes = [C.c_void_p]
    L.kso_table_size.restype = C.c_uint
    L.kso_table_capacity.argtypes = [C.c_void_p]
    L.kso_table_capacity.restype = C.c_uint
    L.kso_build_from_file.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_uint, C.c_uint, C.c_int, C.c_int, C.c_int]
    L.kso_build_from_stream.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, C.c_uint, C.c_uint, C.c_int, C.c_int, C.c_int]
    L.kso_scan_file.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
    L.kso_scan_stream.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, C.c_int]
    L.kso_scan_stream.restype = None
    L.kso_table_rows.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.kso_table_rows.restype = None
    L.kso_decode_file.argtypes = [C.c_char_p, C.POINTER(C.c_size_t), C.POINTER(C.c_long), C.POINTER(C.c_int)]
    L.kso_decode_file.restype = C.c_void_p
    L.kso_free.argtypes = [C.c_void_p]
    L.kso_free.restype = None
    return L


L = _lib()
K = 31


class OracleTable:
    def __init__(self, capacity=8000000, ncols=4):
        self.h = L.kso_table_new(capacity)
        self.ncols = ncols

    def build_file(self, path, default=1, incr=1, idx=0, short_policy=0):
        return L.kso_build_from_file(self.h, os.fsencode(path), K, default, incr, idx, self.ncols, short_policy)

    def build_stream(self, data: bytes, default=1, incr=1, idx=0, short_policy=0):
        return L.kso_build_from_stream(self.h, data, len(data), K, default, incr, idx, self.ncols, short_policy)

    def scan_file(self, path, col):
        n = C.c_uint64(0)
        rc = L.kso_scan_file(self.h, os.fsencode(path), K, col, C.byref(n))
        assert rc == 0, path
        return n.value

    def scan_stream(self, data: bytes, col):
        L.kso_scan_stream(self.h, data, len(data), K, col)

    @property
    def size(self):
        return L.kso_table_size(self.h)

    @property
    def capacity(self):
        return L.kso_table_capacity(self.h)

    def rows(self):
        n = self.size
        keys = C.create_string_buffer(max(n, 1) * (K + 1))
        counts = np.zeros((max(n, 1), self.ncols), dtype=np.uint32)
        L.kso_table_rows(self.h, K, keys, counts.ctypes.data)
        raw = keys.raw
        klist = [raw[i * (K + 1): i * (K + 1) + K].split(b"\0")[0] for i in range(n)]
        return klist, counts[:n]

    def close(self):
        if self.h:
            L.kso_table_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


def decode_file(path):
    n = C.c_size_t(0)
    nrec = C.c_long(0)
    st = C.c_int(0)
    p = L.kso_decode_file(os.fsencode(path), C.byref(n), C.byref(nrec), C.byref(st))
    assert p, path
    data = C.string_at(p, n.value)
    L.kso_free(p)
    return data, nrec.value, st.value


def run_oracle_cli(argv, cwd):
    return subprocess.run([ORACLE_BIN] + argv, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)


def run_reference_cli(argv, cwd):
    """The UNMODIFIED reference binary (only present where oracle/_ref was built)."""
    return subprocess.run([REF_BIN] + argv, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)


def have_reference():
    return os.path.exists(REF_BIN)

KSD_BIN = os.path.join(ORACLE_DIR, "ksd_oracle")
REF_SD_BIN = os.path.join(ORACLE_DIR, "_ref", "strain_detect")


def run_sd_oracle_cli(argv, cwd):
    if not os.path.exists(KSD_BIN):
        subprocess.run(["make", "-C", ORACLE_DIR, "ksd_oracle"], check=True, stdout=subprocess.DEVNULL)
    return subprocess.run([KSD_BIN] + argv, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)


def run_sd_reference_cli(argv, cwd):
    return subprocess.run([REF_SD_BIN] + argv, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
