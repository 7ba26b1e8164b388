"""ctypes binding of libhylight_mi.so (include/hylight_mi.h) - the product's only route to the
hot path.  There is no CPU fallback: a missing library or a missing GPU raises."""
from __future__ import annotations

import ctypes as C
import json
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhylight_mi.so")


class HlmiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libhylight_mi error {code}: {msg}")
        self.code = code


class AvaOpts(C.Structure):
    _fields_ = [("k", C.c_int), ("w", C.c_int), ("hpc", C.c_int), ("min_chain_score", C.c_int),
                ("max_gap", C.c_int), ("bandwidth", C.c_int), ("min_cnt", C.c_int),
                ("min_mid_occ", C.c_int), ("mid_occ_frac", C.c_double),
                ("match", C.c_int), ("mismatch", C.c_int), ("gap_open", C.c_int), ("gap_ext", C.c_int),
                ("ambi", C.c_int), ("min_dp_score", C.c_int), ("end_bonus", C.c_int), ("pair_once", C.c_int),
                ("gap_open2", C.c_int), ("gap_ext2", C.c_int), ("stub_oh", C.c_int), ("zdrop", C.c_int)]


class VqOverlap(C.Structure):
    _fields_ = [("id1", C.c_uint64), ("id2", C.c_uint64), ("pos1", C.c_uint32), ("pos2", C.c_uint32),
                ("perc1", C.c_uint32), ("perc2", C.c_uint32), ("len1", C.c_uint32), ("len2", C.c_uint32),
                ("ord", C.c_char), ("ori1", C.c_char), ("ori2", C.c_char), ("type1", C.c_char), ("type2", C.c_char),
                ("pad", C.c_char * 3)]


ABI_VERSION = 5          # include/hylight_mi.h: HLMI_ABI_VERSION

# every symbol include/hylight_mi.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "hlmi_abi_version": (C.c_int, []),
    "hlmi_init": (C.c_int, [C.c_int, C.c_int]),
    "hlmi_shutdown": (None, []),
    "hlmi_last_error": (C.c_char_p, []),
    "hlmi_version": (C.c_char_p, []),
    "hlmi_split_reads2": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_int,
                                    C.c_int, C.c_double, C.c_int]),
    "hlmi_split_reads2_shard": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_char_p, C.c_int,
                                          C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int]),
    "hlmi_merge_scored_paf": (C.c_int, [C.POINTER(C.c_char_p), C.c_int, C.c_char_p]),
    "hlmi_filter_chunk": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int,
                                    C.c_int]),
    "hlmi_paf_window_filter": (C.c_int, [C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_char_p, C.c_char_p]),
    "hlmi_filter_ovlp_inline": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_double, C.c_int, C.c_double]),
    "hlmi_minimap22sfo": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_double]),
    "hlmi_filter_non_atcg": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int]),
    "hlmi_gfa2fa": (C.c_int, [C.c_char_p, C.c_char_p]),
    "hlmi_pick_up": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]),
    "hlmi_ava_opts_long": (None, [C.POINTER(AvaOpts)]),
    "hlmi_ava_opts_short": (None, [C.POINTER(AvaOpts)]),
    "hlmi_ava": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(AvaOpts), C.c_char_p]),
    "hlmi_miniasm": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p]),
    "hlmi_sfo2overlaps": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_int]),
    "hlmi_vq_parse_overlaps": (C.c_int, [C.c_char_p, C.c_uint32, C.c_uint32, C.c_int, C.c_uint64, C.POINTER(VqOverlap),
                                         C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "hlmi_vq_transitive_edges": (C.c_int, [C.c_uint32, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                           C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)]),
    "hlmi_vq_overlap_scores": (C.c_int, [C.c_char_p, C.POINTER(VqOverlap), C.c_uint64, C.c_double, C.c_uint32, C.POINTER(C.c_double),
                                         C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "hlmi_job_open": (C.c_void_p, [C.c_char_p, C.c_char_p, C.c_int, C.c_int]),
    "hlmi_job_close": (None, [C.c_void_p]),
    "hlmi_job_num_queries": (C.c_int64, [C.c_void_p]),
    "hlmi_job_num_chunks": (C.c_int64, [C.c_void_p]),
    "hlmi_job_sketch_bound": (C.c_int64, [C.c_void_p, C.c_int64, C.c_int64]),
    "hlmi_job_sketch": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                  C.POINTER(C.c_int64)]),
    "hlmi_job_sketch_own": (C.c_int, [C.c_void_p]),
    "hlmi_job_set_query_sketch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "hlmi_job_run": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_char_p]),
    "hlmi_last_stats_json": (C.c_int, [C.c_char_p, C.c_int64]),
}

_lib = None


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64 (same SONAME as
    /opt/rocm's); if both get loaded, whichever initialises second finds no device.  Loading torch's
    copy first makes the dynamic loader bind libhylight_mi.so's NEEDED libamdhip64.so.7 to it, so the
    library, torch's allocator and RCCL share one runtime whatever the import order."""
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load():
    """Load the shared library (fails loudly when it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `python -m hylight_amd.build` "
                              "(there is no CPU fallback)")
        _preload_hip_runtime()
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.hlmi_abi_version() != ABI_VERSION:          # (AvaOpts above mirrors the header's hlmi_ava_opts of that version)
            raise ImportError(f"{LIB_PATH}: ABI version {lib.hlmi_abi_version()}, this binding is written for {ABI_VERSION}")
        _lib = lib
    return _lib


def _b(s):
    return None if s is None else os.fspath(s).encode()


def _check(rc):
    if rc != 0:
        raise HlmiError(rc, load().hlmi_last_error().decode(errors="replace"))


def init(device=-1, host_threads=0):
    _check(load().hlmi_init(device, host_threads))


def shutdown():
    load().hlmi_shutdown()


def version():
    return load().hlmi_version().decode()


def last_stats():
    buf = C.create_string_buffer(1 << 16)
    _check(load().hlmi_last_stats_json(buf, len(buf)))
    return json.loads(buf.value.decode())


def filter_chunk(paf_in, out_paf, len_over, mc, iden, thre=0.0025, min_o=4, long_mode=True):
    _check(load().hlmi_filter_chunk(_b(paf_in), _b(out_paf), len_over, mc, iden, thre, min_o, int(long_mode)))


def paf_window_filter(variant, in_paf, out_path, min_len=60, min_iden=-1.0, min_o=0, sfo=False):
    _check(load().hlmi_paf_window_filter(variant, min_len, min_iden, min_o, int(sfo), _b(in_paf), _b(out_path)))


def filter_ovlp_inline(in_paf, out_paf, min_ovlp_len, min_identity, o=1000, r=0.8):
    _check(load().hlmi_filter_ovlp_inline(_b(in_paf), _b(out_paf), min_ovlp_len, min_identity, o, r))


def minimap22sfo(in_paf, out_sfo, min_overlap_len=0, min_pident=0.0):
    _check(load().hlmi_minimap22sfo(_b(in_paf), _b(out_sfo), min_overlap_len, min_pident))


def filter_non_atcg(fastx, out_fa, model):
    """utils.filter_non_atcg with an explicit output path; model = "fastq" | "fasta" (script/utils.py:81)."""
    _check(load().hlmi_filter_non_atcg(_b(fastx), _b(out_fa), int(model == "fastq")))
    return out_fa


def gfa2fa(gfa, fa):
    _check(load().hlmi_gfa2fa(_b(gfa), _b(fa)))


def pick_up(ovlap_paf, fastx, out_fastx, mode):
    """HyLight.pick_up with an explicit output path; mode = "fastq" | "fasta" (script/HyLight.py:347)."""
    _check(load().hlmi_pick_up(_b(ovlap_paf), _b(fastx), _b(out_fastx), int(mode == "fastq")))
    return out_fastx


def split_reads2(reads_fa, ref_fa, nsplit, out_dir, out_paf, threads=30, len_over=3000, mc=2, iden=0.95,
                 long=False, rank=0, world=1):
    _check(load().hlmi_split_reads2_shard(_b(reads_fa), _b(ref_fa), nsplit, _b(out_dir), _b(out_paf), threads,
                                          len_over, mc, iden, int(long), rank, world))
    return out_paf


def merge_scored_paf(in_pafs, out_paf):
    arr = (C.c_char_p * len(in_pafs))(*[_b(p) for p in in_pafs])
    _check(load().hlmi_merge_scored_paf(arr, len(in_pafs), _b(out_paf)))


def ava_opts_long():
    o = AvaOpts()
    load().hlmi_ava_opts_long(C.byref(o))
    return o


def ava_opts_short():
    o = AvaOpts()
    load().hlmi_ava_opts_short(C.byref(o))
    return o


def ava(target_fa, query_fa, out_paf, opts=None):
    _check(load().hlmi_ava(_b(target_fa), _b(query_fa), C.byref(opts) if opts is not None else None, _b(out_paf)))


def miniasm(paf, reads_fa, out_path, bub_dist=10000, n_rounds_arg=1, max_ext=1, min_dp=1, outfmt="ug"):
    _check(load().hlmi_miniasm(_b(paf), _b(reads_fa), bub_dist, n_rounds_arg, max_ext, min_dp, _b(outfmt),
                               _b(out_path)))


def sfo2overlaps(in_sfo, out_savage, num_singles, num_pairs=0):
    _check(load().hlmi_sfo2overlaps(_b(in_sfo), _b(out_savage), num_singles, num_pairs))


def vq_parse_overlaps(savage_path, min_len=150, min_perc=0, relax_pe=False, max_overlaps=100000000):
    """edge candidates of a 13-column overlaps file -> (list of dicts, n_nonedge, n_skipped)."""
    n, ne, sk = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    _check(load().hlmi_vq_parse_overlaps(_b(savage_path), min_len, min_perc, int(relax_pe), max_overlaps, None, 0,
                                         C.byref(n), C.byref(ne), C.byref(sk)))
    buf = (VqOverlap * max(n.value, 1))()
    _check(load().hlmi_vq_parse_overlaps(_b(savage_path), min_len, min_perc, int(relax_pe), max_overlaps, buf, n.value,
                                         C.byref(n), C.byref(ne), C.byref(sk)))
    rows = []
    for o in buf[:n.value]:
        rows.append(dict(id1=o.id1, id2=o.id2, pos1=o.pos1, pos2=o.pos2, ord=o.ord.decode(), ori1=o.ori1.decode(),
                         ori2=o.ori2.decode(), perc1=o.perc1, perc2=o.perc2, len1=o.len1, len2=o.len2,
                         type1=o.type1.decode(), type2=o.type2.decode()))
    return rows, ne.value, sk.value


def vq_overlap_scores(fastq_singles, overlaps, mismatch=0.0, min_read_len=0):
    """EdgeCalculator::overlap_score for single-single overlaps (dicts as vq_parse_overlaps returns them) ->
    list of (score, mismatch_rate, pos3)."""
    n = len(overlaps)
    buf = (VqOverlap * max(n, 1))()
    for k, o in enumerate(overlaps):
        b = buf[k]
        b.id1, b.id2, b.pos1, b.pos2 = o["id1"], o["id2"], o["pos1"], o.get("pos2", 0)
        b.perc1, b.perc2, b.len1, b.len2 = o.get("perc1", 0), o.get("perc2", 0), o.get("len1", 0), o.get("len2", 0)
        b.ord, b.ori1, b.ori2 = o.get("ord", "-").encode(), o["ori1"].encode(), o["ori2"].encode()
        b.type1, b.type2 = o.get("type1", "s").encode(), o.get("type2", "s").encode()
    sc, mr, p3 = (C.c_double * max(n, 1))(), (C.c_double * max(n, 1))(), (C.c_int64 * max(n, 1))()
    _check(load().hlmi_vq_overlap_scores(_b(fastq_singles), buf, n, mismatch, min_read_len, sc, mr, p3))
    return [(sc[k], mr[k], p3[k]) for k in range(n)]


def vq_transitive_edges(n_vertices, src, dst, ovlen=None, remove_trans=1):
    """-> (flags per edge as bytes: bit 0 transitive in the last round, bit 1 scheduled by the branch reduction; count)."""
    n = len(src)
    a32 = lambda v: (C.c_uint32 * max(n, 1))(*v)
    flags = (C.c_uint8 * max(n, 1))()
    cnt = C.c_uint64(0)
    _check(load().hlmi_vq_transitive_edges(n_vertices, n, a32(src), a32(dst), a32(ovlen) if ovlen is not None else None,
                                           remove_trans, flags, C.byref(cnt)))
    return list(flags[:n]), cnt.value


DEVICE = "cuda"          # where the buffers that cross the C ABI live (stage.py allocates them with torch)


class Job:
    """Staged stage run for the multi-GPU path (sketch shard -> all-gather -> run)."""

    def __init__(self, reads_fa, ref_fa, nsplit, long_mode=True):
        self._h = load().hlmi_job_open(_b(reads_fa), _b(ref_fa), nsplit, int(long_mode))
        if not self._h:
            raise HlmiError(-1, load().hlmi_last_error().decode(errors="replace"))

    def close(self):
        if self._h and _lib is not None:
            _lib.hlmi_job_close(self._h)
        self._h = None

    def __del__(self):          # at interpreter shutdown module globals may already be gone
        try:
            self.close()
        except Exception:
            pass

    @property
    def num_queries(self):
        return load().hlmi_job_num_queries(self._h)

    @property
    def num_chunks(self):
        return load().hlmi_job_num_chunks(self._h)

    def sketch_bound(self, lo, hi):
        return load().hlmi_job_sketch_bound(self._h, lo, hi)

    def sketch(self, lo, hi, dev_mz_ptr, cap, dev_counts_ptr):
        n = C.c_int64(0)
        _check(load().hlmi_job_sketch(self._h, lo, hi, dev_mz_ptr, cap, dev_counts_ptr, C.byref(n)))
        return n.value

    def sketch_own(self):
        _check(load().hlmi_job_sketch_own(self._h))

    def set_query_sketch(self, dev_mz_ptr, n, dev_counts_ptr):
        _check(load().hlmi_job_set_query_sketch(self._h, dev_mz_ptr, n, dev_counts_ptr))

    def run(self, rank, world, len_over, mc, iden, out_paf):
        _check(load().hlmi_job_run(self._h, rank, world, len_over, mc, iden, _b(out_paf)))
