"""ctypes binding of libbbk.so (include/bbk.h).  Fails loudly when the HIP library is missing."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

BOTH_STRANDS, CANONICAL, WITH_COUNTS, UNSORTED, REFERENCE_ORDER, WITH_MASKS = 1, 2, 4, 8, 16, 32
ORDER_SORTED, ORDER_REFERENCE_BUCKETS16 = 0, 1

# every symbol include/bbk.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "bbk_last_error", "bbk_version", "bbk_ctx_create", "bbk_ctx_destroy", "bbk_ctx_set_stream",
    "bbk_ctx_synchronize", "bbk_ctx_trim", "bbk_kmerset_verify_order", "bbk_kmerset_get", "bbk_ctx_profile_enable", "bbk_ctx_profile_reset", "bbk_ctx_profile_get",
    "bbk_reads_from_ascii", "bbk_reads_from_packed", "bbk_host_alloc", "bbk_host_free", "bbk_reads_from_device", "bbk_reads_synth", "bbk_reads_synth_meta", "bbk_reads_from_spades_binary",
    "bbk_reads_write_spades_binary", "bbk_reads_count", "bbk_reads_bases",
    "bbk_reads_get_ascii", "bbk_reads_export_ascii", "bbk_reads_free",
    "bbk_count", "bbk_count_begin", "bbk_count_push_reads", "bbk_count_push_ascii", "bbk_count_finish", "bbk_count_abort",
    "bbk_count_pushed_instances", "bbk_extindex_from_device", "bbk_extindex_export_u32", "bbk_extindex_begin", "bbk_extindex_push_reads", "bbk_extindex_finish",
    "bbk_extindex_abort", "bbk_extindex_finish_with_set", "bbk_count_extindex", "bbk_kmerset_from_device", "bbk_kmerset_from_device_ex", "bbk_kmerset_both_strands", "bbk_kmerset_both_strands_ex", "bbk_words", "bbk_kmerset_size", "bbk_kmerset_k", "bbk_kmerset_keys", "bbk_reads_median_filter",
    "bbk_kmerset_instances", "bbk_kmerset_export", "bbk_kmerset_export_by_owner", "bbk_kmerset_free",
    "bbk_kmerset_write_final_kmers",
    "bbk_extindex_build", "bbk_extindex_size", "bbk_extindex_k", "bbk_extindex_export", "bbk_extindex_clip_tips", "bbk_extindex_free",
    "bbk_unitigs_build", "bbk_unitigs_build_ex", "bbk_unitigs_to_reads", "bbk_unitigs_add_coverage", "bbk_unitigs_add_coverage_counts", "bbk_unitigs_export_kc", "bbk_unitigs_count", "bbk_unitigs_loops", "bbk_unitigs_total_bases",
    "bbk_unitigs_vertices", "bbk_unitigs_links", "bbk_unitigs_export", "bbk_unitigs_export_links",
    "bbk_unitigs_write_gfa", "bbk_unitigs_write_fasta", "bbk_unitigs_write_fastg", "bbk_unitigs_write_spades", "bbk_unitigs_free",
    "bbk_group_create", "bbk_group_size", "bbk_group_device", "bbk_group_destroy", "bbk_group_abort", "bbk_group_exchange_kmers",
    "bbk_group_exchange_extindex", "bbk_group_gather_extindex", "bbk_group_gather_kmers", "bbk_ctx_memory_stats", "bbk_ctx_device_info", "bbk_kmerset_bucket_offsets",
]


class BBKError(RuntimeError):
    pass


def lib_path():
    # BBK_LIB selects an experimental build variant (python -m spades_for_blackbird_amd.build --suffix=...)
    return os.environ.get("BBK_LIB") or os.path.join(_HERE, "libbbk.so")


def _share_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64, libbbk.so links the system ones
    (/opt/rocm).  Two HSA runtimes cannot both drive the GPU from one process: whichever initialises second sees "no
    devices" (observed: engine first, then torch -> "No HIP GPUs are available").  Python users of this binding
    normally have torch in the process (bench.py, distributed.py), so when torch is installed its HIP runtime is put
    into the global symbol scope BEFORE libbbk.so is loaded and libbbk's hip* calls bind to it -- one runtime, in
    either order.  Without torch (or for the C++ CLIs) the system runtime is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if not spec or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand) and not os.environ.get("BBK_SYSTEM_HIP"):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load_library():
    """Loads libbbk.so; raises BBKError if it has not been built (no fallback exists)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    p = lib_path()
    if not os.path.exists(p):
        raise BBKError("libbbk.so is missing: build it with `python -m spades_for_blackbird_amd.build` "
                       "(hipcc --offload-arch=gfx950); this engine has no CPU fallback")
    _share_torch_hip_runtime()
    L = C.CDLL(p)
    vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
    L.bbk_last_error.restype = C.c_char_p
    L.bbk_version.restype = C.c_char_p
    L.bbk_ctx_create.argtypes = [i32, C.POINTER(vp)]
    L.bbk_ctx_destroy.argtypes = [vp]
    L.bbk_ctx_set_stream.argtypes = [vp, vp]
    L.bbk_ctx_synchronize.argtypes = [vp]
    L.bbk_ctx_trim.argtypes = [vp]
    L.bbk_kmerset_verify_order.argtypes = [vp, vp, C.POINTER(u64), C.POINTER(u64), vp, C.c_uint]
    L.bbk_kmerset_get.argtypes = [vp, vp, u64, u64, vp, vp]
    L.bbk_ctx_profile_enable.argtypes = [vp, i32]
    L.bbk_ctx_profile_reset.argtypes = [vp]
    L.bbk_ctx_profile_get.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(u64), C.POINTER(C.c_double)]
    L.bbk_reads_from_ascii.argtypes = [vp, C.c_char_p, vp, u64, C.POINTER(vp)]
    L.bbk_reads_from_device.argtypes = [vp, vp, vp, vp, u64, u64, C.POINTER(vp)]
    L.bbk_reads_from_packed.argtypes = [vp, vp, u64, vp, u64, C.POINTER(vp)]
    L.bbk_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.bbk_host_free.argtypes = [vp]
    L.bbk_reads_synth.argtypes = [vp, u64, u32, u64, C.c_double, u64, u64, C.POINTER(vp)]
    L.bbk_reads_synth_meta.argtypes = [vp, u64, u32, u32, u64, u64, C.c_double, C.c_double, u64, C.POINTER(vp), vp, vp]
    L.bbk_reads_from_spades_binary.argtypes = [vp, C.c_char_p, C.POINTER(vp)]
    L.bbk_reads_write_spades_binary.argtypes = [vp, vp, C.c_char_p]
    L.bbk_reads_count.restype = u64
    L.bbk_reads_count.argtypes = [vp]
    L.bbk_reads_bases.restype = u64
    L.bbk_reads_bases.argtypes = [vp]
    L.bbk_reads_get_ascii.argtypes = [vp, vp, u64, C.c_char_p, u32, C.POINTER(u32)]
    L.bbk_reads_export_ascii.argtypes = [vp, vp, vp, vp, u64]
    L.bbk_reads_free.argtypes = [vp]
    L.bbk_count.argtypes = [vp, vp, C.c_uint, C.c_uint, C.POINTER(vp)]
    L.bbk_count_begin.argtypes = [vp, C.c_uint, C.c_uint, C.POINTER(vp)]
    L.bbk_count_push_reads.argtypes = [vp, vp]
    L.bbk_count_push_ascii.argtypes = [vp, C.c_char_p, vp, u64]
    L.bbk_count_finish.argtypes = [vp, C.POINTER(vp)]
    L.bbk_count_abort.argtypes = [vp]
    L.bbk_count_pushed_instances.restype = u64
    L.bbk_count_pushed_instances.argtypes = [vp]
    L.bbk_extindex_begin.argtypes = [vp, C.c_uint, C.POINTER(vp)]
    L.bbk_extindex_push_reads.argtypes = [vp, vp]
    L.bbk_extindex_finish.argtypes = [vp, C.POINTER(vp)]
    L.bbk_extindex_abort.argtypes = [vp]
    L.bbk_extindex_finish_with_set.argtypes = [vp, C.c_uint, C.POINTER(vp), C.POINTER(vp)]
    L.bbk_count_extindex.argtypes = [vp, vp, C.c_uint, C.c_uint, C.POINTER(vp), C.POINTER(vp)]
    L.bbk_kmerset_from_device.argtypes = [vp, vp, vp, u64, C.c_uint, C.POINTER(vp)]
    L.bbk_kmerset_from_device_ex.argtypes = [vp, vp, vp, u64, C.c_uint, C.c_uint, C.POINTER(vp)]
    L.bbk_kmerset_both_strands.argtypes = [vp, vp, C.POINTER(vp)]
    L.bbk_kmerset_both_strands_ex.argtypes = [vp, vp, C.c_uint, C.POINTER(vp)]
    L.bbk_words.restype = C.c_uint
    L.bbk_words.argtypes = [C.c_uint]
    L.bbk_kmerset_size.restype = u64
    L.bbk_kmerset_size.argtypes = [vp]
    L.bbk_kmerset_keys.restype = vp
    L.bbk_kmerset_keys.argtypes = [vp, C.POINTER(C.c_uint)]
    L.bbk_kmerset_k.restype = C.c_uint
    L.bbk_kmerset_k.argtypes = [vp]
    L.bbk_kmerset_instances.restype = u64
    L.bbk_kmerset_instances.argtypes = [vp]
    L.bbk_kmerset_export.argtypes = [vp, vp, C.c_uint, vp, vp]
    L.bbk_kmerset_export_by_owner.argtypes = [vp, vp, C.c_uint, vp, vp, vp]
    L.bbk_kmerset_free.argtypes = [vp]
    L.bbk_kmerset_write_final_kmers.argtypes = [vp, vp, C.c_char_p]
    if hasattr(L, "bbk_extindex_build"):
        L.bbk_extindex_build.argtypes = [vp, vp, C.c_uint, C.POINTER(vp)]
        L.bbk_extindex_size.restype = u64
        L.bbk_extindex_size.argtypes = [vp]
        L.bbk_extindex_k.restype = C.c_uint
        L.bbk_extindex_k.argtypes = [vp]
        L.bbk_extindex_export.argtypes = [vp, vp, vp, vp]
        L.bbk_extindex_export_u32.argtypes = [vp, vp, vp, vp]
        L.bbk_extindex_from_device.argtypes = [vp, vp, vp, u64, C.c_uint, C.POINTER(vp)]
        L.bbk_extindex_clip_tips.argtypes = [vp, vp, C.c_uint32, C.POINTER(u64), C.POINTER(u64)]
        L.bbk_extindex_free.argtypes = [vp]
    if hasattr(L, "bbk_unitigs_build"):
        L.bbk_unitigs_build.argtypes = [vp, vp, C.POINTER(vp)]
        L.bbk_unitigs_build_ex.argtypes = [vp, vp, C.c_uint, C.POINTER(vp)]
        L.bbk_unitigs_to_reads.argtypes = [vp, vp, C.POINTER(vp)]
        for f in ("count", "loops", "total_bases", "vertices", "links"):
            getattr(L, "bbk_unitigs_" + f).restype = u64
            getattr(L, "bbk_unitigs_" + f).argtypes = [vp]
        L.bbk_unitigs_add_coverage.argtypes = [vp, vp, vp]
        L.bbk_unitigs_add_coverage_counts.argtypes = [vp, vp, vp]
        L.bbk_unitigs_export_kc.argtypes = [vp, vp, vp]
        L.bbk_unitigs_export.argtypes = [vp, vp, vp, vp]
        L.bbk_unitigs_export_links.argtypes = [vp, vp, vp]
        L.bbk_unitigs_write_gfa.argtypes = [vp, vp, C.c_char_p]
        L.bbk_unitigs_write_fasta.argtypes = [vp, vp, C.c_char_p]
        L.bbk_unitigs_write_fastg.argtypes = [vp, vp, C.c_char_p]
        L.bbk_unitigs_write_spades.argtypes = [vp, vp, C.c_char_p]
        L.bbk_unitigs_free.argtypes = [vp]
    L.bbk_group_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_uint, C.POINTER(vp)]
    L.bbk_group_size.argtypes = [vp]
    L.bbk_group_device.argtypes = [vp, C.c_int]
    L.bbk_group_destroy.argtypes = [vp]
    L.bbk_group_destroy.restype = None
    L.bbk_group_abort.argtypes = [vp]
    L.bbk_group_abort.restype = None
    L.bbk_group_exchange_kmers.argtypes = [vp, C.c_int, vp, vp, C.c_uint, C.POINTER(vp)]
    L.bbk_group_exchange_extindex.argtypes = [vp, C.c_int, vp, vp, C.POINTER(vp)]
    L.bbk_group_gather_extindex.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.POINTER(vp)]
    L.bbk_group_gather_kmers.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.POINTER(vp)]
    L.bbk_ctx_memory_stats.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(C.c_double), C.POINTER(u64), C.POINTER(u64)]
    L.bbk_ctx_device_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.bbk_kmerset_bucket_offsets.argtypes = [vp, vp, vp]
    _LIB = L
    return L


def _check(rc):
    if rc != 0:
        raise BBKError("bbk error %d: %s" % (rc, load_library().bbk_last_error().decode(errors="replace")))


def _ptr(x):
    """Host numpy array, torch tensor (host or device) or raw int -> void pointer."""
    if x is None:
        return None
    if isinstance(x, int):
        return C.c_void_p(x)
    if isinstance(x, np.ndarray):
        return C.c_void_p(x.ctypes.data)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    raise TypeError("cannot take a pointer of %r" % type(x))


def words(k):
    return (k + 31) // 32


class Context:
    """One per GPU / process (bbk_ctx)."""

    def __init__(self, device=0, stream=None):
        self._L = load_library()
        h = C.c_void_p()
        _check(self._L.bbk_ctx_create(device, C.byref(h)))
        self._h = h
        self.device = device
        if stream is not None:
            self.set_stream(stream)

    def set_stream(self, stream):
        """stream: raw hipStream_t as int, or a torch.cuda.Stream."""
        raw = stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream)
        _check(self._L.bbk_ctx_set_stream(self._h, C.c_void_p(raw)))

    def synchronize(self):
        _check(self._L.bbk_ctx_synchronize(self._h))

    def trim(self):
        """Gives the device memory the allocator holds but does not use back to the driver (free chunks at the end of
        the calling thread's arena are unmapped and released).  The next calls map what they need again: ~30 ms/GiB
        when the driver has to clear the memory first -- trim when another process or library needs the memory, not
        between the steps of a loop."""
        _check(self._L.bbk_ctx_trim(self._h))

    def memory_stats(self):
        """dict(mapped_now, mapped_total, map_seconds, device_free, device_total) of this thread's allocator / device"""
        a, b, f, t = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        sec = C.c_double()
        _check(self._L.bbk_ctx_memory_stats(self._h, C.byref(a), C.byref(b), C.byref(sec), C.byref(f), C.byref(t)))
        return {"mapped_now": a.value, "mapped_total": b.value, "map_seconds": sec.value, "device_free": f.value,
                "device_total": t.value}

    def device_info(self):
        """{'num_cus', 'num_xcds'}: what the engine found at context creation (8 XCDs switch the XCD-local fill fronts on)"""
        a, b = C.c_int(), C.c_int()
        _check(self._L.bbk_ctx_device_info(self._h, C.byref(a), C.byref(b)))
        return {"num_cus": a.value, "num_xcds": b.value}

    def profile(self, on=True):
        _check(self._L.bbk_ctx_profile_enable(self._h, int(on)))

    def profile_reset(self):
        _check(self._L.bbk_ctx_profile_reset(self._h))

    def profile_get(self, family):
        ms, n, b = C.c_double(), C.c_uint64(), C.c_double()
        _check(self._L.bbk_ctx_profile_get(self._h, family.encode(), C.byref(ms), C.byref(n), C.byref(b)))
        return {"ms": ms.value, "launches": n.value, "bytes": b.value}

    # ---- reads -------------------------------------------------------------------------------
    def reads_from_ascii(self, reads):
        bs = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
        offs = np.zeros(len(bs) + 1, dtype=np.uint64)
        if bs:
            offs[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
        return self.reads_from_blob(b"".join(bs), offs)

    def reads_from_blob(self, blob, offsets):
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        h = C.c_void_p()
        _check(self._L.bbk_reads_from_ascii(self._h, blob, _ptr(offsets), len(offsets) - 1, C.byref(h)))
        return Reads(self, h)

    def reads_from_packed(self, words, lens):
        """Host-packed reads (numpy uint64 words, uint32 lengths; every read starts on a word)."""
        words = np.ascontiguousarray(words, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        h = C.c_void_p()
        _check(self._L.bbk_reads_from_packed(self._h, _ptr(words), len(words), _ptr(lens), len(lens), C.byref(h)))
        return Reads(self, h)

    def reads_from_device(self, d_words, d_word_off, d_len, n_reads, n_words):
        h = C.c_void_p()
        _check(self._L.bbk_reads_from_device(self._h, _ptr(d_words), _ptr(d_word_off), _ptr(d_len), n_reads, n_words,
                                             C.byref(h)))
        r = Reads(self, h)
        r._keep = (d_words, d_word_off, d_len)
        return r

    def reads_from_spades_binary(self, seq_path):
        """SPAdes binary read cache (<prefix>.seq)."""
        h = C.c_void_p()
        _check(self._L.bbk_reads_from_spades_binary(self._h, seq_path.encode(), C.byref(h)))
        return Reads(self, h)

    def reads_synth(self, n_reads, read_len=150, genome_len=None, sub_rate=0.005, seed_genome=42, seed_reads=43):
        if genome_len is None:
            genome_len = max(read_len, n_reads * read_len // 50)
        h = C.c_void_p()
        _check(self._L.bbk_reads_synth(self._h, n_reads, read_len, genome_len, sub_rate, seed_genome, seed_reads,
                                       C.byref(h)))
        return Reads(self, h)

    def reads_synth_meta(self, n_reads, read_len=150, n_genomes=200, min_len=500_000, max_len=8_000_000, sigma=2.0,
                         sub_rate=0.005, seed=44, want_community=False):
        """Metagenome-shaped reads (SURVEY 8d / BASELINE configs[4]); want_community: also (genome lengths, abundances)."""
        h = C.c_void_p()
        gl = np.zeros(n_genomes, dtype=np.uint64)
        ab = np.zeros(n_genomes, dtype=np.float64)
        _check(self._L.bbk_reads_synth_meta(self._h, n_reads, read_len, n_genomes, min_len, max_len, sigma, sub_rate, seed,
                                            C.byref(h), _ptr(gl), _ptr(ab)))
        r = Reads(self, h)
        return (r, gl, ab) if want_community else r

    # ---- operators -----------------------------------------------------------------------------
    def count(self, reads, k, flags=BOTH_STRANDS):
        """KMerCounter::Count analogue (reference kmer_index_builder.hpp:195-217)."""
        h = C.c_void_p()
        _check(self._L.bbk_count(self._h, reads._h, k, flags, C.byref(h)))
        return KMerSet(self, h)

    def counter(self, k, flags=BOTH_STRANDS):
        """Streaming count (bbk_count_begin / push / finish): KMerSortingSplitter's bounded rounds + MergeKMers."""
        return Counter(self, k, flags)

    def extbuilder(self, k):
        """Streaming extension-index build (bbk_extindex_begin / push / finish)."""
        return ExtBuilder(self, k)

    def kmerset_from_device(self, d_keys, n, k, d_counts=None, flags=0):
        h = C.c_void_p()
        _check(self._L.bbk_kmerset_from_device_ex(self._h, _ptr(d_keys), _ptr(d_counts), n, k, flags, C.byref(h)))
        return KMerSet(self, h)

    def median_filter(self, reads, counts, threshold):
        """CoverageFilter::CheckMedianMlt for every read: uint8[n] (1 = keep), see bbk_reads_median_filter."""
        keep = np.zeros(len(reads), dtype=np.uint8)
        kept = C.c_uint64(0)
        self._L.bbk_reads_median_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p,
                                                    C.POINTER(C.c_uint64)]
        _check(self._L.bbk_reads_median_filter(self._h, reads._h, counts._h, threshold, _ptr(keep), C.byref(kept)))
        assert int(kept.value) == int(keep.sum())
        return keep

    def extindex(self, reads, k):
        """DeBruijnExtensionIndexBuilder::BuildExtensionIndexFromStream analogue."""
        h = C.c_void_p()
        _check(self._L.bbk_extindex_build(self._h, reads._h, k, C.byref(h)))
        return ExtIndex(self, h)

    def count_extindex(self, reads, k, flags=BOTH_STRANDS):
        """k-mer set (both strands) and extension index from ONE pass over the reads: (KMerSet, ExtIndex)"""
        hs, hx = C.c_void_p(), C.c_void_p()
        _check(self._L.bbk_count_extindex(self._h, reads._h, k, flags, C.byref(hs), C.byref(hx)))
        return KMerSet(self, hs), ExtIndex(self, hx)

    def extindex_from_device(self, d_keys, d_masks_u32, n, k):
        """Index of (canonical k-mer, u32 mask) records in HBM: any order, duplicates OR-ed (a shard after the exchange)."""
        h = C.c_void_p()
        _check(self._L.bbk_extindex_from_device(self._h, _ptr(d_keys), _ptr(d_masks_u32), n, k, C.byref(h)))
        return ExtIndex(self, h)

    def unitigs(self, ext, ref_threads=0):
        """UnbranchingPathExtractor + FastGraphFromSequencesConstructor analogue.  ref_threads > 0: perfect loops are
        collected in the k-mer file order of a reference run with that -t (loop rotation, SplitLoop choice)."""
        h = C.c_void_p()
        _check(self._L.bbk_unitigs_build_ex(self._h, ext._h, ref_threads, C.byref(h)))
        return Unitigs(self, h)

    def close(self):
        if getattr(self, "_h", None):
            self._L.bbk_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _Handle:
    _free = None

    def __init__(self, ctx, h):
        self.ctx, self._h, self._L = ctx, h, ctx._L

    def free(self):
        if getattr(self, "_h", None):
            getattr(self._L, self._free)(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Reads(_Handle):
    _free = "bbk_reads_free"

    def __len__(self):
        return int(self._L.bbk_reads_count(self._h))

    @property
    def bases(self):
        return int(self._L.bbk_reads_bases(self._h))

    def to_ascii(self):
        """(blob bytes, offsets np.uint64[n+1])"""
        n = len(self)
        offs = np.zeros(n + 1, dtype=np.uint64)
        _check(self._L.bbk_reads_export_ascii(self.ctx._h, self._h, None, _ptr(offs), 0))
        total = int(offs[n])
        buf = np.zeros(total + 1, dtype=np.uint8)
        _check(self._L.bbk_reads_export_ascii(self.ctx._h, self._h, _ptr(buf), _ptr(offs), total))
        return buf[:total].tobytes(), offs

    def write_spades_binary(self, prefix):
        _check(self._L.bbk_reads_write_spades_binary(self.ctx._h, self._h, prefix.encode()))

    def to_list(self):
        blob, offs = self.to_ascii()
        return [blob[int(offs[i]):int(offs[i + 1])].decode() for i in range(len(offs) - 1)]


class KMerSet(_Handle):
    _free = "bbk_kmerset_free"

    def __len__(self):
        return int(self._L.bbk_kmerset_size(self._h))

    @property
    def k(self):
        return int(self._L.bbk_kmerset_k(self._h))

    @property
    def instances(self):
        return int(self._L.bbk_kmerset_instances(self._h))

    def device_keys(self):
        """(device pointer of the records, order they are stored in); valid until the set is freed."""
        o = C.c_uint(0)
        p = self._L.bbk_kmerset_keys(self._h, C.byref(o))
        return p, int(o.value)

    def both_strands(self, flags=0):
        """canon U rc(canon); flags=REFERENCE_ORDER stores it in the final_kmers order."""
        h = C.c_void_p()
        _check(self._L.bbk_kmerset_both_strands_ex(self.ctx._h, self._h, flags, C.byref(h)))
        return KMerSet(self.ctx, h)

    def export(self, order=ORDER_SORTED, with_counts=False):
        n, nw = len(self), words(self.k)
        keys = np.zeros((n, nw), dtype=np.uint64)
        cnt = np.zeros(n, dtype=np.uint32) if with_counts else None
        _check(self._L.bbk_kmerset_export(self.ctx._h, self._h, order, _ptr(keys), _ptr(cnt)))
        return (keys, cnt) if with_counts else keys

    def export_to(self, dst_keys, order=ORDER_SORTED, dst_counts=None):
        """dst_*: preallocated torch tensors (device or host) or numpy arrays."""
        _check(self._L.bbk_kmerset_export(self.ctx._h, self._h, order, _ptr(dst_keys), _ptr(dst_counts)))

    def export_by_owner(self, nranks, dst_keys=None, dst_counts=None):
        n, nw = len(self), words(self.k)
        counts = np.zeros(nranks, dtype=np.uint64)
        ret = None
        if dst_keys is None:
            dst_keys = ret = np.zeros((n, nw), dtype=np.uint64)
        _check(self._L.bbk_kmerset_export_by_owner(self.ctx._h, self._h, nranks, _ptr(dst_keys), _ptr(dst_counts),
                                                   _ptr(counts)))
        return (ret, counts) if ret is not None else counts

    def verify_order(self):
        """(ascending runs of the stored order, equal neighbours, start index of the first runs) -- on the device"""
        runs, eq = C.c_uint64(), C.c_uint64()
        starts = np.zeros(33, dtype=np.uint64)
        _check(self._L.bbk_kmerset_verify_order(self.ctx._h, self._h, C.byref(runs), C.byref(eq), _ptr(starts), 33))
        return int(runs.value), int(eq.value), [int(x) for x in starts[:min(int(runs.value), 33)]]

    def get(self, first, count, with_counts=False):
        keys = np.zeros((count, words(self.k)), dtype=np.uint64)
        cnt = np.zeros(count, dtype=np.uint32) if with_counts else None
        _check(self._L.bbk_kmerset_get(self.ctx._h, self._h, first, count, _ptr(keys), _ptr(cnt)))
        return (keys, cnt) if with_counts else keys

    def write_final_kmers(self, path):
        _check(self._L.bbk_kmerset_write_final_kmers(self.ctx._h, self._h, path.encode()))


class Counter:
    """bbk_counter: push batches of reads, finish() returns the KMerSet (and releases the counter)."""

    def __init__(self, ctx, k, flags):
        self.ctx, self._L = ctx, ctx._L
        h = C.c_void_p()
        _check(self._L.bbk_count_begin(ctx._h, k, flags, C.byref(h)))
        self._h = h

    def push(self, reads):
        _check(self._L.bbk_count_push_reads(self._h, reads._h))

    def push_ascii(self, reads):
        bs = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
        offs = np.zeros(len(bs) + 1, dtype=np.uint64)
        if bs:
            offs[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
        _check(self._L.bbk_count_push_ascii(self._h, b"".join(bs), _ptr(offs), len(bs)))

    @property
    def instances(self):
        return int(self._L.bbk_count_pushed_instances(self._h))

    def finish(self):
        h = C.c_void_p()
        c, self._h = self._h, None
        _check(self._L.bbk_count_finish(c, C.byref(h)))
        return KMerSet(self.ctx, h)

    def abort(self):
        if getattr(self, "_h", None):
            self._L.bbk_count_abort(self._h)
            self._h = None

    def __del__(self):
        try:
            self.abort()
        except Exception:
            pass


class ExtBuilder:
    """bbk_extbuilder: push batches of reads, finish() returns the ExtIndex."""

    def __init__(self, ctx, k):
        self.ctx, self._L = ctx, ctx._L
        h = C.c_void_p()
        _check(self._L.bbk_extindex_begin(ctx._h, k, C.byref(h)))
        self._h = h

    def push(self, reads):
        _check(self._L.bbk_extindex_push_reads(self._h, reads._h))

    def finish(self):
        h = C.c_void_p()
        b, self._h = self._h, None
        _check(self._L.bbk_extindex_finish(b, C.byref(h)))
        return ExtIndex(self.ctx, h)

    def finish_with_set(self, flags=BOTH_STRANDS):
        """(KMerSet of both strands, ExtIndex) from the one accumulation (bbk_extindex_finish_with_set)"""
        hs, hx = C.c_void_p(), C.c_void_p()
        b, self._h = self._h, None
        _check(self._L.bbk_extindex_finish_with_set(b, flags, C.byref(hs), C.byref(hx)))
        return KMerSet(self.ctx, hs), ExtIndex(self.ctx, hx)

    def abort(self):
        if getattr(self, "_h", None):
            self._L.bbk_extindex_abort(self._h)
            self._h = None

    def __del__(self):
        try:
            self.abort()
        except Exception:
            pass


class ExtIndex(_Handle):
    _free = "bbk_extindex_free"

    def __len__(self):
        return int(self._L.bbk_extindex_size(self._h))

    @property
    def k(self):
        return int(self._L.bbk_extindex_k(self._h))

    def clip_tips(self, length_bound):
        """EarlyTipClipperProcessor(index, length_bound).ClipTips() in place; returns (isolated k-mers, removed links)."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        _check(self._L.bbk_extindex_clip_tips(self.ctx._h, self._h, length_bound, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def export_to_u32(self, dst_keys, dst_masks_u32):
        """keys + masks widened to u32 (the payload layout of the owner exchange) into preallocated tensors/arrays"""
        _check(self._L.bbk_extindex_export_u32(self.ctx._h, self._h, _ptr(dst_keys), _ptr(dst_masks_u32)))

    def export(self):
        n, nw = len(self), words(self.k)
        keys = np.zeros((n, nw), dtype=np.uint64)
        masks = np.zeros(n, dtype=np.uint8)
        _check(self._L.bbk_extindex_export(self.ctx._h, self._h, _ptr(keys), _ptr(masks)))
        return keys, masks


class Unitigs(_Handle):
    _free = "bbk_unitigs_free"

    def __len__(self):
        return int(self._L.bbk_unitigs_count(self._h))

    @property
    def n_loops(self):
        return int(self._L.bbk_unitigs_loops(self._h))

    @property
    def n_vertices(self):
        return int(self._L.bbk_unitigs_vertices(self._h))

    @property
    def n_links(self):
        return int(self._L.bbk_unitigs_links(self._h))

    @property
    def total_bases(self):
        return int(self._L.bbk_unitigs_total_bases(self._h))

    def sequences(self):
        n = len(self)
        offs = np.zeros(n + 1, dtype=np.uint64)
        buf = np.zeros(self.total_bases + 1, dtype=np.uint8)
        _check(self._L.bbk_unitigs_export(self.ctx._h, self._h, _ptr(buf), _ptr(offs)))
        b = buf.tobytes()
        return [b[int(offs[i]):int(offs[i + 1])].decode() for i in range(n)]

    def to_reads(self):
        """the condensed edges as a device-resident read set (contigs of this K as input of the next,
        stages/construction.cpp:117-119,228-236)"""
        h = C.c_void_p()
        _check(self._L.bbk_unitigs_to_reads(self.ctx._h, self._h, C.byref(h)))
        return Reads(self.ctx, h)

    def add_coverage(self, reads):
        """gbuilder -c: KC / DP of every condensed edge from the reads."""
        _check(self._L.bbk_unitigs_add_coverage(self.ctx._h, self._h, reads._h))

    def add_coverage_counts(self, kp1_counts):
        """same from an already counted ascending canonical (k+1)-mer KMerSet with counts"""
        _check(self._L.bbk_unitigs_add_coverage_counts(self.ctx._h, self._h, kp1_counts._h))

    def kc(self):
        a = np.zeros(len(self), dtype=np.uint64)
        _check(self._L.bbk_unitigs_export_kc(self.ctx._h, self._h, _ptr(a)))
        return a

    def links(self):
        a = np.zeros((self.n_links, 4), dtype=np.uint32)
        _check(self._L.bbk_unitigs_export_links(self.ctx._h, self._h, _ptr(a)))
        return a

    def write_gfa(self, path):
        _check(self._L.bbk_unitigs_write_gfa(self.ctx._h, self._h, path.encode()))

    def write_fastg(self, path):
        _check(self._L.bbk_unitigs_write_fastg(self.ctx._h, self._h, path.encode()))

    def write_fasta(self, path):
        _check(self._L.bbk_unitigs_write_fasta(self.ctx._h, self._h, path.encode()))

    def write_spades(self, basename):
        """<basename>.grseq + <basename>.cvr (the SPAdes binary graph of gbuilder --spades)"""
        _check(self._L.bbk_unitigs_write_spades(self.ctx._h, self._h, basename.encode()))
