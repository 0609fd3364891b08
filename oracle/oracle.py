"""ctypes binding of oracle/liboracle.so -- the CPU ORACLE.

TEST INFRASTRUCTURE ONLY.  Allowed importers: tests/, __graft_entry__.smoke(),
bench.py's cpu_baseline leg.  The product package never imports this module.
Each wrapper mirrors one function of oracle/bbk_oracle.h (which cites the
reference file:line it restates).
"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = [os.path.join(_HERE, f) for f in ("bbk_oracle.c", "bbk_oracle.h")]
    stale = not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference") and not os.path.exists(os.path.join(_HERE, "_ref", "libxxh3_ref.so")):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return so


class _Kmer(C.Structure):
    _fields_ = [("w", C.c_uint64 * 4)]


class _Reads(C.Structure):
    _fields_ = [("bases", C.c_char_p), ("offsets", C.POINTER(C.c_uint64)), ("n", C.c_size_t)]


class _ExtIndex(C.Structure):
    _fields_ = [
        ("k", C.c_int), ("nbuckets", C.c_uint),
        ("n_kp1", C.c_size_t), ("kp1", C.POINTER(C.c_uint64)), ("kp1_count", C.POINTER(C.c_uint32)),
        ("n_k", C.c_size_t), ("kmers", C.POINTER(C.c_uint64)), ("masks", C.POINTER(C.c_uint8)),
        ("bucket_start", C.POINTER(C.c_size_t)),
    ]


class _Unitigs(C.Structure):
    _fields_ = [("n", C.c_size_t), ("n_loops", C.c_size_t), ("seq", C.POINTER(C.c_char_p)),
                ("len", C.POINTER(C.c_size_t)), ("kc", C.POINTER(C.c_uint64))]


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_xxh3_64.restype = C.c_uint64
        L.orc_xxh3_64.argtypes = [C.POINTER(C.c_uint64), C.c_int]
        L.orc_bucket.restype = C.c_uint64
        L.orc_bucket.argtypes = [C.POINTER(C.c_uint64), C.c_int, C.c_uint64]
        L.orc_longest_valid.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.orc_kmer_from_ascii.argtypes = [C.c_char_p, C.c_int, C.POINTER(_Kmer)]
        L.orc_kmer_to_ascii.argtypes = [C.POINTER(_Kmer), C.c_int, C.c_char_p]
        L.orc_kmer_shl.argtypes = [C.POINTER(_Kmer), C.c_int, C.c_int]
        L.orc_kmer_rc.argtypes = [C.POINTER(_Kmer), C.c_int, C.POINTER(_Kmer)]
        L.orc_kmer_is_minimal.argtypes = [C.POINTER(_Kmer), C.c_int]
        L.orc_kmer_less_nucl.argtypes = [C.POINTER(_Kmer), C.POINTER(_Kmer), C.c_int]
        L.orc_kmercount.argtypes = [C.POINTER(_Reads), C.c_int, C.c_uint, C.c_int,
                                    C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_size_t),
                                    C.POINTER(C.POINTER(C.c_uint32))]
        L.orc_extindex_build.argtypes = [C.POINTER(_Reads), C.c_int, C.c_uint, C.POINTER(_ExtIndex)]
        L.orc_extindex_free.argtypes = [C.POINTER(_ExtIndex)]
        L.orc_unitigs_extract.argtypes = [C.POINTER(_ExtIndex), C.POINTER(_Unitigs)]
        L.orc_unitigs_free.argtypes = [C.POINTER(_Unitigs)]
        L.orc_gfa_write.argtypes = [C.POINTER(_ExtIndex), C.POINTER(_Unitigs), C.c_int, C.c_void_p,
                                    C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.orc_unitigs_fasta_write.argtypes = [C.POINTER(_Unitigs), C.c_void_p]
        L.orc_fastx_read.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        _LIB = L
    return _LIB


_libc = C.CDLL(None)
_libc.fopen.restype = C.c_void_p
_libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
_libc.fclose.argtypes = [C.c_void_p]
_libc.free.argtypes = [C.c_void_p]


def words(k):
    return (k + 31) // 32


def _mk_reads(reads):
    """reads: list of str/bytes.  Returns (struct, keepalive)."""
    bs = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
    offs = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offs[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    blob = b"".join(bs)
    st = _Reads(blob, offs.ctypes.data_as(C.POINTER(C.c_uint64)), len(bs))
    return st, (blob, offs)


def mk_reads_blob(blob, offsets):
    """blob: bytes of concatenated reads, offsets: np.uint64[n+1]."""
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    st = _Reads(blob, offsets.ctypes.data_as(C.POINTER(C.c_uint64)), len(offsets) - 1)
    return st, (blob, offsets)


def read_fastx(path):
    """FASTA/FASTQ(.gz) file -> (blob bytes, offsets np.uint64[n+1]) with the reference parser's semantics."""
    b, o, n = C.c_void_p(), C.c_void_p(), C.c_size_t()
    if lib().orc_fastx_read(os.fsencode(path), C.byref(b), C.byref(o), C.byref(n)) != 0:
        raise IOError("cannot open %s" % path)
    nn = n.value
    offs = np.ctypeslib.as_array(C.cast(o, C.POINTER(C.c_uint64)), shape=(nn + 1,)).copy()
    blob = C.string_at(b, int(offs[nn]))
    _libc.free(b)
    _libc.free(o)
    return blob, offs


def read_fastx_list(path):
    blob, offs = read_fastx(path)
    return [blob[int(offs[i]):int(offs[i + 1])].decode() for i in range(len(offs) - 1)]


def xxh3_64(words_arr):
    a = np.ascontiguousarray(words_arr, dtype=np.uint64)
    return int(lib().orc_xxh3_64(a.ctypes.data_as(C.POINTER(C.c_uint64)), len(a)))


def bucket(words_arr, nb):
    a = np.ascontiguousarray(words_arr, dtype=np.uint64)
    return int(lib().orc_bucket(a.ctypes.data_as(C.POINTER(C.c_uint64)), len(a), nb))


def longest_valid(s):
    b = s.encode() if isinstance(s, str) else s
    f, t = C.c_size_t(), C.c_size_t()
    lib().orc_longest_valid(b, len(b), C.byref(f), C.byref(t))
    return f.value, t.value


def kmer_from_ascii(s):
    km = _Kmer()
    lib().orc_kmer_from_ascii(s.encode(), len(s), C.byref(km))
    return km


def kmer_words(s):
    km = kmer_from_ascii(s)
    return [int(km.w[i]) for i in range(words(len(s)))]


def kmer_str(km, k):
    buf = C.create_string_buffer(k + 1)
    lib().orc_kmer_to_ascii(C.byref(km), k, buf)
    return buf.value.decode()


def kmer_shl(s, c):
    km = kmer_from_ascii(s)
    lib().orc_kmer_shl(C.byref(km), len(s), "ACGT".index(c))
    return kmer_str(km, len(s))


def kmer_rc(s):
    km, out = kmer_from_ascii(s), _Kmer()
    lib().orc_kmer_rc(C.byref(km), len(s), C.byref(out))
    return kmer_str(out, len(s))


def kmer_is_minimal(s):
    return bool(lib().orc_kmer_is_minimal(C.byref(kmer_from_ascii(s)), len(s)))


def kmercount(reads, k, nbuckets=16, nthreads=1, with_counts=False, blob=None):
    """Returns np.uint64[n, words(k)] in final_kmers order (+ np.uint32[n] counts)."""
    st, keep = blob if blob is not None else _mk_reads(reads)
    out = C.POINTER(C.c_uint64)()
    cnt = C.POINTER(C.c_uint32)()
    n = C.c_size_t()
    rc = lib().orc_kmercount(C.byref(st), k, nbuckets, nthreads, C.byref(out), C.byref(n),
                             C.byref(cnt) if with_counts else None)
    if rc != 0:
        raise RuntimeError("orc_kmercount failed: %d" % rc)
    nw = words(k)
    arr = np.ctypeslib.as_array(out, shape=(n.value * nw,)).copy().reshape(n.value, nw) if n.value else \
        np.zeros((0, nw), dtype=np.uint64)
    _libc.free(out)
    if with_counts:
        c = np.ctypeslib.as_array(cnt, shape=(n.value,)).copy() if n.value else np.zeros(0, dtype=np.uint32)
        _libc.free(cnt)
        return arr, c
    return arr


class ExtIndex:
    """orc_extindex_build result (canonical (k+1)-mers, canonical k-mers + InOutMask bytes)."""

    def __init__(self, reads, k, T=1, blob=None):
        self._st = _ExtIndex()
        st, keep = blob if blob is not None else _mk_reads(reads)
        rc = lib().orc_extindex_build(C.byref(st), k, T, C.byref(self._st))
        if rc != 0:
            raise RuntimeError("orc_extindex_build failed: %d" % rc)
        self.k = k
        s = self._st
        nw, nw1 = words(k), words(k + 1)
        self.n_kp1, self.n_k = s.n_kp1, s.n_k
        self.kp1 = np.ctypeslib.as_array(s.kp1, shape=(s.n_kp1 * nw1,)).copy().reshape(s.n_kp1, nw1) \
            if s.n_kp1 else np.zeros((0, nw1), np.uint64)
        self.kp1_count = np.ctypeslib.as_array(s.kp1_count, shape=(s.n_kp1,)).copy() if s.n_kp1 else \
            np.zeros(0, np.uint32)
        self.kmers = np.ctypeslib.as_array(s.kmers, shape=(s.n_k * nw,)).copy().reshape(s.n_k, nw) \
            if s.n_k else np.zeros((0, nw), np.uint64)
        self.masks = np.ctypeslib.as_array(s.masks, shape=(s.n_k,)).copy() if s.n_k else np.zeros(0, np.uint8)

    def clip_tips(self, length_bound):
        """EarlyTipClipperProcessor::ClipTips on the masks held in C; returns (isolated k-mers, removed links) and
        refreshes self.masks."""
        links = C.c_size_t(0)
        lib().orc_extindex_clip_tips.restype = C.c_size_t
        lib().orc_extindex_clip_tips.argtypes = [C.POINTER(_ExtIndex), C.c_size_t, C.POINTER(C.c_size_t)]
        removed = lib().orc_extindex_clip_tips(C.byref(self._st), length_bound, C.byref(links))
        s = self._st
        self.masks = np.ctypeslib.as_array(s.masks, shape=(s.n_k,)).copy() if s.n_k else np.zeros(0, np.uint8)
        return int(removed), int(links.value)

    def unitigs(self, threads=1):
        """Runs UnbranchingPathExtractor (destroys the masks held in C, like the reference).  threads > 1: the path
        phase over 16 * threads chunks in parallel (debruijn_graph_constructor.hpp:351-375); same result."""
        return Unitigs(self, threads)

    def __del__(self):
        try:
            lib().orc_extindex_free(C.byref(self._st))
        except Exception:
            pass


class Unitigs:
    def __init__(self, ext, threads=1):
        self._ext = ext
        self._st = _Unitigs()
        rc = lib().orc_unitigs_extract_mt(C.byref(ext._st), C.byref(self._st), int(threads))
        if rc != 0:
            raise RuntimeError("orc_unitigs_extract failed: %d" % rc)
        s = self._st
        self.n, self.n_loops = s.n, s.n_loops
        self.seqs = [s.seq[i].decode() for i in range(s.n)]
        self.kc = [int(s.kc[i]) for i in range(s.n)]

    def gfa(self, with_cov=False):
        """Returns (gfa_text, n_vertices, n_links)."""
        with tempfile.NamedTemporaryFile(suffix=".gfa", delete=False) as tf:
            path = tf.name
        try:
            fp = _libc.fopen(path.encode(), b"w")
            nv, nl = C.c_size_t(), C.c_size_t()
            rc = lib().orc_gfa_write(C.byref(self._ext._st), C.byref(self._st), int(with_cov), fp,
                                     C.byref(nv), C.byref(nl))
            _libc.fclose(fp)
            if rc != 0:
                raise RuntimeError("orc_gfa_write failed: %d" % rc)
            with open(path) as f:
                return f.read(), nv.value, nl.value
        finally:
            os.unlink(path)

    def fasta(self):
        with tempfile.NamedTemporaryFile(suffix=".fa", delete=False) as tf:
            path = tf.name
        try:
            fp = _libc.fopen(path.encode(), b"w")
            lib().orc_unitigs_fasta_write(C.byref(self._st), fp)
            _libc.fclose(fp)
            with open(path) as f:
                return f.read()
        finally:
            os.unlink(path)

    def __del__(self):
        try:
            lib().orc_unitigs_free(C.byref(self._st))
        except Exception:
            pass
