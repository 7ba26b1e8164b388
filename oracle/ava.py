"""ORACLE loader (test infrastructure): ctypes access to oracle/liboracle_ava.so, the CPU
restatement of the overlapper spec (oracle/ava_oracle.c).  Never imported by hylight_amd/."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle_ava.so")


class Opts(C.Structure):
    _fields_ = [("k", C.c_int), ("w", C.c_int), ("hpc", C.c_int), ("min_chain_score", C.c_int),
                ("max_gap", C.c_int), ("bandwidth", C.c_int), ("min_cnt", C.c_int),
                ("min_mid_occ", C.c_int), ("mid_occ_frac", C.c_double),
                ("match", C.c_int), ("mismatch", C.c_int), ("gap_open", C.c_int), ("gap_ext", C.c_int),
                ("ambi", C.c_int), ("min_dp_score", C.c_int), ("end_bonus", C.c_int), ("pair_once", C.c_int),
                ("gap_open2", C.c_int), ("gap_ext2", C.c_int), ("stub_oh", C.c_int), ("zdrop", C.c_int)]


def opts_long():
    """The constants of script/filter_overlap_slr2.py:51 (ava-pb -Hk19 -m100 -g10000)."""
    return Opts(19, 5, 1, 100, 10000, 2000, 3, 10, 2e-4, 2, 4, 4, 2, 1, 80, 0, 1, 24, 1, -1, 400)       # -O4,24 -E2,1 -z400 (preset defaults)


def opts_short():
    """The constants of script/filter_overlap_slr2.py:55 (--sr -k21 -w11 -s60 -m30 -n2 -A4 -B2 --end-bonus=100; from the
    --sr preset: -g200 -r50 -O12 -E2 -f1000)."""
    return Opts(21, 11, 0, 30, 200, 50, 2, 1000, 0.0, 4, 2, 12, 2, 1, 60, 100, 0, 32, 1, -1, 0)     # --sr: -O12,32 -E2,1; extensions stay within 256 rows (max_gap 200): no z-drop


_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "ava_oracle.c")
        if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "oracle"], stdout=subprocess.DEVNULL)
        l = C.CDLL(_LIB)
        l.oracle_hash64.restype = C.c_uint64
        l.oracle_hash64.argtypes = [C.c_uint64, C.c_uint64]
        l.oracle_sketch.restype = C.c_int64
        l.oracle_sketch.argtypes = [C.c_char_p, C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64]
        l.oracle_ava.restype = C.c_int
        l.oracle_ava.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(Opts), C.c_char_p]
        l.oracle_last_counts.restype = None
        l.oracle_last_counts.argtypes = [C.POINTER(C.c_long), C.POINTER(C.c_long)]
        _lib = l
    return _lib


def hash64(key, mask):
    return lib().oracle_hash64(key, mask)


def sketch(seq: bytes, rid=0, k=19, w=5, hpc=1):
    """(n,2) uint64 array of minimizers (x = hash<<8|span, y = rid<<32|pos<<1|strand)."""
    cap = max(len(seq), 1)
    out = np.zeros((cap, 2), dtype=np.uint64)
    n = lib().oracle_sketch(seq, len(seq), rid, k, w, hpc, out.ctypes.data, cap)
    return out[:n].copy()


def ava(target_fa, query_fa, out_paf, opts=None):
    o = opts or opts_long()
    rc = lib().oracle_ava(os.fspath(target_fa).encode(), os.fspath(query_fa).encode(), C.byref(o),
                          os.fspath(out_paf).encode())
    if rc != 0:
        raise RuntimeError(f"oracle_ava failed: {rc}")
    return out_paf


def last_counts():
    """(pieces reported, of them written as stubs) of the last ava() call of this process."""
    a, b = C.c_long(0), C.c_long(0)
    lib().oracle_last_counts(C.byref(a), C.byref(b))
    return a.value, b.value
