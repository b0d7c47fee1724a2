"""ctypes binding of include/strainer_kmer.h (see that header for the contract of each call).

Fails loudly when the native library is absent: there is no fallback implementation.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def library_path():
    # SK_LIBRARY: another build of the same library (A/B timing of kernel variants)
    return os.environ.get("SK_LIBRARY") or os.path.join(_HERE, "lib", "libstrainer_kmer.so")


def cli_path(name="kmer_scrub_count"):
    return os.path.join(_HERE, "bin", name)


def _load():
    p = library_path()
    if not os.path.exists(p):
        raise ImportError(
            f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C strainer2_amd/csrc` (there is no non-native fallback)")
    return C.CDLL(p, mode=C.RTLD_GLOBAL)


lib = _load()

SK_K = 31
SK_REF_TABLE_SLOTS = 8000000
SK_KEY_NONE = 0xFFFFFFFFFFFFFFFF
SK_OK = 0
SK_E_NODEVICE = -1
SK_E_OPEN = -5
SK_E_SPLIT = -9
SK_E_PLAN = -10

# every symbol include/strainer_kmer.h declares (tests check that the library exports all of them)
ABI_SYMBOLS = [
    "sk_ctx_create", "sk_ctx_destroy", "sk_last_error", "sk_strerror", "sk_table_load", "sk_table_load_ex",
    "sk_table_load_wide", "sk_table_load_text", "sk_table_build_from_text", "sk_table_export_keys", "sk_table_export_keys_of", "sk_scan_stream", "sk_scan_device", "sk_pinned_alloc", "sk_pinned_free", "sk_scan_pinned", "sk_scan_pinned_packed", "sk_scan_device_packed", "sk_pack_stream", "sk_packed_bytes",
    "sk_ticket_wait", "sk_tally_batch", "sk_sync", "sk_counts_fetch",
    "sk_counts_set", "sk_counts_set_rows", "sk_counts_zero", "sk_counts_device_ptr", "sk_table_rows", "sk_table_cols",
    "sk_counts_allreduce", "sk_comm_init", "sk_comm_init_ex", "sk_rendezvous_exchange", "sk_comm_destroy", "sk_comm_sum_u32", "sk_comm_agree_u64", "sk_comm_max_u64", "sk_comm_world", "sk_scan_timing", "sk_set_option", "sk_dev_alloc", "sk_dev_free",
    "sk_dev_upload", "sk_dev_download",
    "skh_keyset_from_file", "skh_keyset_from_stream", "skh_keyset_free", "skh_keyset_key",
    "skh_keyset_load", "skh_scan_file", "skh_scan_list", "skh_scan_list_uncut", "skh_list_plan_hash", "skh_list_plan_owners", "skh_print_counts",
    "skh_kmer_scrub_count_main", "skh_strain_detect_main", "skh_strain_detect_resident", "skh_decode_file",
    "sk_filter_create", "sk_filter_destroy", "sk_filter_load", "sk_filter_load_counts", "sk_filter_sums",
    "sk_filter_hist", "sk_filter_joint", "sk_filter_above", "skh_scrub_filter_main", "skh_scrub_filter_resident",
    "sk_distinct_count", "sk_first_seen_count", "skh_coverage_depth_main",
    "sk_batch_create", "sk_batch_destroy", "sk_batch_sync", "sk_batch_fill", "sk_batch_fill_packed", "sk_tally_launch", "sk_tally_collect", "sk_tally_collect_sparse",
    "sk_union_create", "sk_union_destroy", "sk_union_tally_launch", "sk_union_tally_collect", "sk_union_last_error", "sk_union_scan_timing", "sk_union_sync",
    "sk_union_members", "sk_union_rows",
]


class _KeysetStruct(C.Structure):
    _fields_ = [("nrows", C.c_uint32), ("nwide", C.c_uint32),
                ("packed", C.POINTER(C.c_uint64)), ("first_count", C.POINTER(C.c_uint32)),
                ("locality", C.POINTER(C.c_uint32)),
                ("wide_keys", C.POINTER(C.c_char)), ("wide_rows", C.POINTER(C.c_uint32)),
                ("final_slots", C.c_uint32), ("short_records", C.c_uint64),
                ("text2", C.POINTER(C.c_uint32)), ("text_bases", C.c_uint32), ("first_pos", C.POINTER(C.c_uint32))]


_SINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.c_uint64)

lib.sk_ctx_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
lib.sk_ctx_destroy.argtypes = [C.c_void_p]
lib.sk_ctx_destroy.restype = None
lib.sk_last_error.argtypes = [C.c_void_p]
lib.sk_last_error.restype = C.c_char_p
lib.sk_strerror.argtypes = [C.c_int]
lib.sk_strerror.restype = C.c_char_p
lib.sk_table_load.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
lib.sk_table_load_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
lib.sk_table_load_wide.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
lib.sk_table_load_text.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
lib.sk_scan_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32]
lib.sk_scan_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32]
lib.sk_pinned_alloc.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_uint64]
lib.sk_pinned_free.argtypes = [C.c_void_p, C.c_void_p]
lib.sk_scan_pinned.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64)]
lib.sk_ticket_wait.argtypes = [C.c_void_p, C.c_uint64]
lib.sk_scan_pinned_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64)]
lib.sk_scan_device_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32]
lib.sk_pack_stream.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_int)]
lib.sk_packed_bytes.argtypes = [C.c_uint64]
lib.sk_packed_bytes.restype = C.c_uint64
lib.sk_tally_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                               C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
lib.sk_sync.argtypes = [C.c_void_p]
lib.sk_counts_fetch.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
lib.sk_counts_set.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
lib.sk_counts_zero.argtypes = [C.c_void_p, C.c_uint32]
lib.sk_counts_device_ptr.argtypes = [C.c_void_p]
lib.sk_counts_device_ptr.restype = C.c_void_p
lib.sk_table_rows.argtypes = [C.c_void_p]
lib.sk_table_rows.restype = C.c_uint32
lib.sk_table_cols.argtypes = [C.c_void_p]
lib.sk_table_cols.restype = C.c_uint32
lib.sk_counts_allreduce.argtypes = [C.c_void_p, C.c_void_p]
lib.sk_rendezvous_exchange.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_void_p, C.c_double]
lib.sk_scan_timing.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]
lib.sk_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_long]
lib.sk_dev_alloc.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_uint64]
lib.sk_dev_free.argtypes = [C.c_void_p, C.c_void_p]
lib.sk_dev_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
lib.sk_dev_download.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
lib.skh_keyset_from_file.argtypes = [C.POINTER(_KeysetStruct), C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32]
lib.skh_keyset_from_stream.argtypes = [C.POINTER(_KeysetStruct), C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32]
lib.skh_keyset_free.argtypes = [C.POINTER(_KeysetStruct)]
lib.skh_keyset_free.restype = None
lib.skh_keyset_key.argtypes = [C.POINTER(_KeysetStruct), C.c_uint32, C.c_char_p]
lib.skh_keyset_key.restype = None
lib.skh_keyset_load.argtypes = [C.c_void_p, C.POINTER(_KeysetStruct), C.c_uint32]
lib.skh_scan_file.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.POINTER(C.c_uint64)]
lib.skh_list_plan_hash.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.POINTER(C.c_uint64)]
lib.skh_list_plan_hash.restype = C.c_int
lib.skh_list_plan_owners.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
lib.skh_list_plan_owners.restype = C.c_int
lib.skh_scan_list.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_uint32, C.c_void_p, C.c_void_p,
                              C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
lib.skh_scan_list_uncut.argtypes = lib.skh_scan_list.argtypes
lib.skh_print_counts.argtypes = [C.c_void_p, C.POINTER(_KeysetStruct), C.c_void_p, C.c_int]
lib.skh_kmer_scrub_count_main.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_void_p, C.c_void_p]
lib.skh_decode_file.argtypes = [C.c_char_p, C.c_uint64, _SINK, C.c_void_p, C.POINTER(C.c_uint64)]
lib.skh_decode_file.restype = C.c_int64

lib.sk_filter_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
lib.sk_filter_destroy.argtypes = [C.c_void_p]
lib.sk_filter_destroy.restype = None
lib.sk_filter_load.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
lib.sk_filter_load_counts.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32]
lib.sk_filter_sums.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_uint64),
                               C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
lib.sk_filter_hist.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_uint32, C.c_void_p]
lib.sk_filter_joint.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_uint64, C.c_void_p]
lib.sk_filter_above.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
lib.skh_scrub_filter_main.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_void_p, C.c_void_p]
lib.skh_scrub_filter_resident.argtypes = [C.c_void_p, C.POINTER(_KeysetStruct), C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p]

lib.sk_distinct_count.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]
lib.skh_coverage_depth_main.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_void_p, C.c_void_p]

lib.sk_batch_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
lib.sk_batch_destroy.argtypes = [C.c_void_p]
lib.sk_batch_destroy.restype = None
lib.sk_batch_fill.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32]
lib.sk_batch_fill_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32]
lib.sk_tally_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64]
lib.sk_tally_collect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
lib.sk_tally_collect_sparse.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint64)]
lib.sk_union_create.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
lib.sk_union_destroy.argtypes = [C.c_void_p]
lib.sk_union_destroy.restype = None
lib.sk_union_scan_timing.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]
lib.sk_union_tally_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
lib.sk_union_tally_collect.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint64)]
lib.sk_union_last_error.argtypes = [C.c_void_p]
lib.sk_union_last_error.restype = C.c_char_p
lib.sk_union_members.argtypes = [C.c_void_p]
lib.sk_union_members.restype = C.c_uint32
lib.sk_union_rows.argtypes = [C.c_void_p]
lib.sk_union_rows.restype = C.c_uint32

_libc = C.CDLL(None)
_libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
_libc.fopen.restype = C.c_void_p
_libc.fclose.argtypes = [C.c_void_p]


class SKError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        msg = lib.sk_strerror(code).decode()
        super().__init__(f"libstrainer_kmer error {code}: {msg}" + (f" ({detail})" if detail else ""))


class Keyset:
    """Strain key set in reference output order (host layer: skh_keyset_*)."""

    def __init__(self):
        self._s = _KeysetStruct()
        self._live = False

    @classmethod
    def from_file(cls, path, initial_slots=SK_REF_TABLE_SLOTS, default_val=1, incr=1):
        ks = cls()
        rc = lib.skh_keyset_from_file(C.byref(ks._s), os.fsencode(path), initial_slots, default_val, incr)
        if rc:
            raise SKError(rc, path)
        ks._live = True
        return ks

    @classmethod
    def from_stream(cls, stream: bytes, initial_slots=SK_REF_TABLE_SLOTS, default_val=1, incr=1):
        ks = cls()
        rc = lib.skh_keyset_from_stream(C.byref(ks._s), stream, len(stream), initial_slots, default_val, incr)
        if rc:
            raise SKError(rc)
        ks._live = True
        return ks

    nrows = property(lambda self: self._s.nrows)
    nwide = property(lambda self: self._s.nwide)
    final_slots = property(lambda self: self._s.final_slots)
    short_records = property(lambda self: self._s.short_records)

    def packed(self):
        return np.ctypeslib.as_array(self._s.packed, shape=(max(self.nrows, 1),))[: self.nrows].copy()

    def first_count(self):
        return np.ctypeslib.as_array(self._s.first_count, shape=(max(self.nrows, 1),))[: self.nrows].copy()

    def key(self, row):
        buf = C.create_string_buffer(32)
        lib.skh_keyset_key(C.byref(self._s), row, buf)
        return buf.value

    def keys(self):
        """All keys as bytes, in row order (vectorised decode of the packed ones)."""
        pk = self.packed()
        out = np.empty((self.nrows, SK_K), dtype=np.uint8)
        lut = np.frombuffer(b"ACGT", dtype=np.uint8)
        for i in range(SK_K):
            out[:, i] = lut[((pk >> np.uint64(2 * (SK_K - 1 - i))) & np.uint64(3)).astype(np.int64)]
        keys = [bytes(r) for r in out]
        for r in np.nonzero(pk == np.uint64(SK_KEY_NONE))[0]:
            keys[int(r)] = self.key(int(r))
        return keys

    def close(self):
        if self._live:
            lib.skh_keyset_free(C.byref(self._s))
            self._live = False

    def __del__(self):
        self.close()


class KmerContext:
    """One device context (device layer: sk_*)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        rc = lib.sk_ctx_create(C.byref(self._h), device)
        if rc:
            self._h = None
            raise SKError(rc, "sk_ctx_create: this library needs an MI355X-class GPU; there is no CPU path")
        self._bufs = []

    def _ck(self, rc):
        if rc:
            raise SKError(rc, lib.sk_last_error(self._h).decode())

    def set_option(self, name, value):
        self._ck(lib.sk_set_option(self._h, name.encode(), value))

    def load_keyset(self, ks: Keyset, ncols=4):
        self._ck(lib.skh_keyset_load(self._h, C.byref(ks._s), ncols))

    def load_table(self, keys: np.ndarray, ncols=4):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        self._ck(lib.sk_table_load(self._h, keys.ctypes.data, len(keys), ncols))

    def scan_stream(self, stream, col):
        if isinstance(stream, np.ndarray):
            stream = np.ascontiguousarray(stream, dtype=np.uint8)
            self._ck(lib.sk_scan_stream(self._h, stream.ctypes.data, stream.size, col))
        else:
            self._ck(lib.sk_scan_stream(self._h, stream, len(stream), col))

    def tally_batch(self, stream: bytes, rec_start, type_col=0, informative_value=2):
        """Per-record tallies (strain_detect): returns (tally[nrec, 2], hits[n, 2] sorted by position)."""
        rec_start = np.ascontiguousarray(rec_start, dtype=np.uint32)
        nrec = len(rec_start)
        tally = np.zeros((nrec, 2), dtype=np.uint32)
        cap = max(len(stream), 16)
        hits = np.zeros((cap, 2), dtype=np.uint32)
        nh = C.c_uint64(0)
        self._ck(lib.sk_tally_batch(self._h, stream, len(stream), rec_start.ctypes.data, nrec, type_col, informative_value,
                                    tally.ctypes.data, hits.ctypes.data, cap, C.byref(nh)))
        hits = hits[: nh.value]
        return tally, hits[np.argsort(hits[:, 0], kind="stable")]

    def pinned_alloc(self, nbytes):
        """A pinned host buffer as a writable numpy uint8 array (free with pinned_free(arr))."""
        p = C.c_void_p()
        self._ck(lib.sk_pinned_alloc(self._h, C.byref(p), nbytes))
        arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(nbytes,))
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p.value
        return arr

    def pinned_free(self, arr):
        self._ck(lib.sk_pinned_free(self._h, self._pinned.pop(arr.ctypes.data)))

    def scan_pinned(self, arr, nbytes, col, offset=0):
        t = C.c_uint64(0)
        self._ck(lib.sk_scan_pinned(self._h, arr.ctypes.data + offset, nbytes, col, C.byref(t)))
        return t.value

    def scan_device_packed(self, dev_ptr, nbytes, col):
        self._ck(lib.sk_scan_device_packed(self._h, dev_ptr, nbytes, col))

    def scan_pinned_packed(self, packed_arr, nbytes, col):
        """a batch packed by pack_stream() into page-locked memory (pinned_alloc): returns the ticket"""
        t = C.c_uint64(0)
        self._ck(lib.sk_scan_pinned_packed(self._h, packed_arr.ctypes.data, nbytes, col, C.byref(t)))
        return t.value

    def ticket_wait(self, ticket):
        self._ck(lib.sk_ticket_wait(self._h, ticket))

    def scan_device(self, dev_ptr, nbytes, col):
        self._ck(lib.sk_scan_device(self._h, dev_ptr, nbytes, col))

    def scan_file(self, path, col):
        bases = C.c_uint64(0)
        self._ck(lib.skh_scan_file(self._h, os.fsencode(path), col, C.byref(bases)))
        return bases.value

    def scan_list(self, list_path, col, skip=None, rank=0, world=1, uncut=False):
        """skh_scan_list; uncut=True: the whole-file plan (skh_scan_list_uncut).  With world > 1 and no in-library communicator
        a cut that does not hold raises SKError(SK_E_SPLIT): use strainer2_amd.dist.scan_list_sharded, which agrees across
        the ranks and scans again uncut."""
        bases = C.c_uint64(0)
        fn = lib.skh_scan_list_uncut if uncut else lib.skh_scan_list
        self._ck(fn(self._h, os.fsencode(list_path), None if skip is None else os.fsencode(skip),
                    col, None, None, rank, world, C.byref(bases)))
        return bases.value

    @staticmethod
    def list_plan_owners(list_path, world, skip=None):
        """per list line the rank that scans it (skh_list_plan_owners; 0xFFFFFFFF skipped, 0xFFFFFFFE cut across ranks)"""
        n = C.c_uint32(0)
        args = (os.fsencode(list_path), None if skip is None else os.fsencode(skip), world)
        rc = lib.skh_list_plan_owners(*args, None, 0, C.byref(n))
        own = np.empty(max(n.value, 1), dtype=np.uint32)
        if rc == SK_OK:
            rc = lib.skh_list_plan_owners(*args, own.ctypes.data, n.value, C.byref(n))
        if rc != SK_OK:
            raise OSError(f"skh_list_plan_owners({list_path}) failed: {rc}")
        return own[: n.value]

    def sync(self):
        self._ck(lib.sk_sync(self._h))

    @staticmethod
    def list_plan_hash(list_path, world, skip=None):
        """hash of the work plan scan_list(list_path, .., world=world) follows on every rank (skh_list_plan_hash)"""
        h = C.c_uint64(0)
        rc = lib.skh_list_plan_hash(os.fsencode(list_path), None if skip is None else os.fsencode(skip), world, C.byref(h))
        if rc != SK_OK:
            raise OSError(f"skh_list_plan_hash({list_path}) failed: {rc}")
        return h.value

    def counts(self, col):
        out = np.empty(self.nrows, dtype=np.uint32)
        if self.nrows:
            self._ck(lib.sk_counts_fetch(self._h, col, out.ctypes.data))
        return out

    def set_counts(self, col, values):
        values = np.ascontiguousarray(values, dtype=np.uint32)
        assert values.size == self.nrows
        self._ck(lib.sk_counts_set(self._h, col, values.ctypes.data))

    def zero_counts(self, col):
        self._ck(lib.sk_counts_zero(self._h, col))

    nrows = property(lambda self: lib.sk_table_rows(self._h))
    ncols = property(lambda self: lib.sk_table_cols(self._h))

    def counts_device_ptr(self):
        return lib.sk_counts_device_ptr(self._h)

    def scan_timing(self, reset=False):
        ms = C.c_double(0)
        n = C.c_uint64(0)
        self._ck(lib.sk_scan_timing(self._h, C.byref(ms), C.byref(n), int(reset)))
        return ms.value, n.value

    def dev_alloc(self, nbytes):
        p = C.c_void_p()
        self._ck(lib.sk_dev_alloc(self._h, C.byref(p), nbytes))
        self._bufs.append(p.value)
        return p.value

    def dev_free(self, ptr):
        self._ck(lib.sk_dev_free(self._h, ptr))
        self._bufs.remove(ptr)

    def dev_upload(self, ptr, arr, offset=0):
        arr = np.ascontiguousarray(arr)
        self._ck(lib.sk_dev_upload(self._h, ptr + offset, arr.ctypes.data, arr.nbytes))

    def dev_download(self, ptr, nbytes):
        out = np.empty(nbytes, dtype=np.uint8)
        self._ck(lib.sk_dev_download(self._h, out.ctypes.data, ptr, nbytes))
        return out

    def distinct_count(self, keys, sample, nsamples):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        sample = np.ascontiguousarray(sample, dtype=np.uint32)
        assert len(keys) == len(sample)
        uniq = np.zeros(nsamples, dtype=np.uint64)
        total = np.zeros(nsamples, dtype=np.uint64)
        self._ck(lib.sk_distinct_count(self._h, keys.ctypes.data, sample.ctypes.data, len(keys), nsamples,
                                       uniq.ctypes.data, total.ctypes.data))
        return uniq, total

    def print_counts(self, ks: Keyset, path, with_drug_column=False):
        fp = _libc.fopen(os.fsencode(path), b"w")
        try:
            self._ck(lib.skh_print_counts(self._h, C.byref(ks._s), fp, int(with_drug_column)))
        finally:
            _libc.fclose(fp)

    def close(self):
        if self._h:
            for p in list(self._bufs):
                lib.sk_dev_free(self._h, p)
            self._bufs = []
            lib.sk_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class KmerUnion:
    """One table for several resident strains (sk_union_*): a batch is tallied against all of them in one launch."""

    def __init__(self, contexts, type_col=0, informative_value=2):
        self._members = list(contexts)                    # (they must outlive the union)
        arr = (C.c_void_p * len(self._members))(*[c._h for c in self._members])
        self._h = C.c_void_p()
        rc = lib.sk_union_create(arr, len(self._members), type_col, informative_value, C.byref(self._h))
        if rc:
            self._h = None
            raise SKError(rc, lib.sk_last_error(self._members[0]._h).decode() if self._members else "no members")
        self._batch = C.c_void_p()
        rc = lib.sk_batch_create(self._members[0]._h, C.byref(self._batch))
        if rc:
            raise SKError(rc, "sk_batch_create")

    @property
    def rows(self):
        return lib.sk_union_rows(self._h)

    def tally_batch(self, stream: bytes, rec_start, hits_cap=None):
        """returns (tally[nrec, members, 2], hits[n, 3] = (member, window-end offset, the member's row) sorted)"""
        rec_start = np.ascontiguousarray(rec_start, dtype=np.uint32)
        nrec, n = len(rec_start), len(self._members)
        rc = lib.sk_batch_fill(self._batch, stream, len(stream), rec_start.ctypes.data, nrec)
        if rc:
            raise SKError(rc, lib.sk_last_error(self._members[0]._h).decode())
        cap = hits_cap if hits_cap is not None else max(len(stream) * 2, 16)
        while True:
            rc = lib.sk_union_tally_launch(self._h, self._batch, cap)
            if rc:
                raise SKError(rc, lib.sk_union_last_error(self._h).decode())
            recs = np.zeros((nrec * n + 1, 3), dtype=np.uint32)
            hits = np.zeros((max(cap, 1), 2), dtype=np.uint32)
            nr, nh = C.c_uint64(0), C.c_uint64(0)
            rc = lib.sk_union_tally_collect(self._h, recs.ctypes.data, nrec * n, C.byref(nr), hits.ctypes.data, C.byref(nh))
            if rc:
                raise SKError(rc, lib.sk_union_last_error(self._h).decode())
            if nh.value <= cap:
                break
            cap = nh.value + 16                           # the log overflowed: once more with room
        tally = np.zeros((nrec * n, 2), dtype=np.uint32)
        recs = recs[: nr.value]
        tally[recs[:, 0]] = recs[:, 1:]
        hits = hits[: nh.value]
        out = np.stack([hits[:, 1] >> 27, hits[:, 0], hits[:, 1] & ((1 << 27) - 1)], axis=1) if len(hits) else np.zeros((0, 3), dtype=np.uint32)
        order = np.lexsort((out[:, 2], out[:, 1], out[:, 0]))
        return tally.reshape(nrec, n, 2), out[order]

    def close(self):
        if getattr(self, "_batch", None):
            lib.sk_batch_destroy(self._batch)
            self._batch = None
        if getattr(self, "_h", None):
            lib.sk_union_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def pack_stream(stream, out=None):
    """the host-side 2-bit pre-pack (sk_pack_stream): returns (packed bytes as a uint8 array -- code words, then masks --, odd)"""
    buf = np.frombuffer(stream, dtype=np.uint8) if isinstance(stream, (bytes, bytearray)) else np.ascontiguousarray(stream, dtype=np.uint8)
    n = int(lib.sk_packed_bytes(buf.size))
    if out is None:
        out = np.empty(max(n, 1), dtype=np.uint8)
    odd = C.c_int(0)
    rc = lib.sk_pack_stream(buf.ctypes.data, buf.size, out.ctypes.data, C.byref(odd))
    if rc:
        raise SKError(rc)
    return out[:n], bool(odd.value)


class ScrubFilter:
    """Device side of the scrub filter (sk_filter_*): count columns resident on the GPU."""

    def __init__(self, ctx: KmerContext):
        self._ctx = ctx
        self._h = C.c_void_p()
        ctx._ck(lib.sk_filter_create(ctx._h, C.byref(self._h)))
        self.n = 0

    def load(self, pan, meta, gone=None):
        pan = np.ascontiguousarray(pan, dtype=np.int64)
        meta = np.ascontiguousarray(meta, dtype=np.int64)
        assert len(pan) == len(meta)
        g = None if gone is None else np.ascontiguousarray(gone, dtype=np.uint8)
        self._ctx._ck(lib.sk_filter_load(self._h, pan.ctypes.data, meta.ctypes.data, None if g is None else g.ctypes.data, len(pan)))
        self.n = len(pan)

    def load_counts(self, pan_col=1, meta_col=2, drug_col=-1):
        self._ctx._ck(lib.sk_filter_load_counts(self._h, pan_col, meta_col, drug_col))
        self.n = lib.sk_table_rows(self._ctx._h)

    def sums(self):
        v = [C.c_int64(), C.c_int64(), C.c_uint64(), C.c_uint64(), C.c_uint64()]
        self._ctx._ck(lib.sk_filter_sums(self._h, *[C.byref(x) for x in v]))
        return tuple(int(x.value) for x in v)

    def hist(self, which, lo, nbins):
        out = np.zeros(nbins + 1, dtype=np.uint64)
        self._ctx._ck(lib.sk_filter_hist(self._h, which, lo, nbins, out.ctypes.data))
        return out

    def joint(self, pan_sum, meta_sum, n_scrub):
        out = np.zeros(self.n, dtype=np.uint8)
        self._ctx._ck(lib.sk_filter_joint(self._h, pan_sum, meta_sum, n_scrub, out.ctypes.data))
        return out

    def above(self, pan_thr, meta_thr):
        out = np.zeros(self.n, dtype=np.uint8)
        self._ctx._ck(lib.sk_filter_above(self._h, pan_thr, meta_thr, out.ctypes.data))
        return out

    def close(self):
        if self._h:
            lib.sk_filter_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


def decode_file(path, chunk_bytes=1 << 26):
    """Host reader only (no GPU): the record stream the scan would be fed, as a list of chunks."""
    chunks = []

    def sink(_user, ptr, n):
        chunks.append(C.string_at(ptr, n))
        return 0

    bases = C.c_uint64(0)
    nrec = lib.skh_decode_file(os.fsencode(path), chunk_bytes, _SINK(sink), None, C.byref(bases))
    if nrec < 0:
        raise SKError(int(nrec), path)
    return chunks, int(nrec), bases.value


def run_cli_inprocess(argv, stdout_path, stderr_path):
    """Call the program's main() in this process (the bin/ program is the same function)."""
    args = [b"kmer_scrub_count"] + [os.fsencode(a) for a in argv]
    arr = (C.c_char_p * (len(args) + 1))(*args, None)
    fo = _libc.fopen(os.fsencode(stdout_path), b"w")
    fe = _libc.fopen(os.fsencode(stderr_path), b"w")
    try:
        return lib.skh_kmer_scrub_count_main(len(args), arr, fo, fe)
    finally:
        _libc.fclose(fo)
        _libc.fclose(fe)
