"""ctypes view of oracle/libfxoracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (fastx_toolkit_amd) must never do so.  See oracle/fxoracle.h for the parity status.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

STAGE_CLIP, STAGE_QTRIM, STAGE_QFILTER, STAGE_REVCOMP, STAGE_FTRIM, STAGE_FTRIM_END, STAGE_MASK, STAGE_ARTIFACTS, STAGE_NFILTER = 1, 2, 4, 8, 16, 32, 64, 128, 256
CLIP_DISCARD_NON_CLIPPED, CLIP_DISCARD_CLIPPED, CLIP_KEEP_N, CLIP_ADAPTER_ONLY = 1, 2, 4, 8
NCOUNTERS = 24
C_INPUT, C_KEPT, C_KEPT_BASES, C_CLIP_TOO_SHORT, C_CLIP_ADAPTER_ONLY, C_CLIP_NO_ADAPTER, C_CLIP_ADAPTER_FOUND, \
    C_CLIP_N, C_QTRIM_DROPPED, C_QFILTER_DROPPED, C_FTRIM_DROPPED, C_CLIP_OUT, C_QTRIM_OUT = range(13)


class Params(C.Structure):
    _fields_ = [
        ("stages", C.c_uint32), ("qoffset", C.c_int32),
        ("qt_threshold", C.c_int32), ("qt_min_len", C.c_int32),
        ("qf_min_quality", C.c_int32), ("qf_min_percent", C.c_int32),
        ("adapter", C.c_char * 100), ("clip_min_len", C.c_uint32), ("clip_keep_delta", C.c_int32),
        ("clip_min_adapter_len", C.c_int32), ("clip_flags", C.c_uint32),
        ("ft_first", C.c_int32), ("ft_last", C.c_int32), ("ft_trim_end", C.c_uint32), ("ft_min_len", C.c_uint32),
        ("mask_min_quality", C.c_int32), ("mask_char", C.c_uint32), ("nf_keep_n", C.c_uint32),
    ]


class Batch(C.Structure):
    _fields_ = [("bases", C.c_void_p), ("qual", C.c_void_p), ("len", C.c_void_p),
                ("fixed_len", C.c_uint32), ("stride", C.c_uint32), ("n", C.c_uint64)]


class Out(C.Structure):
    _fields_ = [("res", C.c_void_p), ("out_bases", C.c_void_p), ("out_qual", C.c_void_p),
                ("out_len", C.c_void_p), ("kept_index", C.c_void_p), ("counters", C.c_uint64 * NCOUNTERS)]


class AlignRes(C.Structure):
    _fields_ = [(k, C.c_int64) for k in ("query_size", "query_start", "query_end", "target_size", "target_start",
                                         "target_end", "gaps", "neutral_matches", "matches", "mismatches")] + [("score", C.c_float)]


def make_params(stages=0, qoffset=33, qt_threshold=0, qt_min_len=0, qf_min_quality=0, qf_min_percent=0,
                adapter=b"CCTTAAGG", clip_min_len=5, clip_keep_delta=0, clip_min_adapter_len=0, clip_flags=0,
                ft_first=1, ft_last=0, ft_trim_end=0, ft_min_len=0, mask_min_quality=10, mask_char="N", nf_keep_n=0, cls=Params):
    """Parameter block shared (field for field) by the oracle and include/fxg.h's fxg_params."""
    if isinstance(adapter, str):
        adapter = adapter.encode()
    p = cls()
    p.stages, p.qoffset = stages, qoffset
    p.qt_threshold, p.qt_min_len = qt_threshold, qt_min_len
    p.qf_min_quality, p.qf_min_percent = qf_min_quality, qf_min_percent
    p.adapter = adapter
    p.clip_min_len, p.clip_keep_delta = clip_min_len, clip_keep_delta
    p.clip_min_adapter_len, p.clip_flags = clip_min_adapter_len, clip_flags
    p.ft_first, p.ft_last, p.ft_trim_end, p.ft_min_len = ft_first, ft_last, ft_trim_end, ft_min_len
    p.mask_min_quality = mask_min_quality
    p.mask_char = mask_char if isinstance(mask_char, int) else ord(mask_char)
    p.nf_keep_n = nf_keep_n
    return p


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libfxoracle.so")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(_HERE, "fxoracle.c")):
            subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])
        L = C.CDLL(so)
        L.fxo_splitmix.restype = C.c_uint64
        L.fxo_splitmix.argtypes = [C.c_uint64]
        L.fxo_synth_batch.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32]
        L.fxo_synth_fastq.restype = C.c_size_t
        L.fxo_synth_fastq.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, C.c_void_p]
        L.fxo_run_pipeline.argtypes = [C.POINTER(Batch), C.POINTER(Params), C.POINTER(Out)]
        L.fxo_run_pipeline_h.argtypes = [C.POINTER(Batch), C.POINTER(Params), C.POINTER(Out), C.c_void_p]
        L.fxo_aligner_new.restype = C.c_void_p
        L.fxo_aligner_free.argtypes = [C.c_void_p]
        L.fxo_align.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(AlignRes)]
        L.fxo_adapter_cutoff_index.argtypes = [C.POINTER(AlignRes), C.c_int]
        L.fxo_parse_fastq.restype = C.c_int64
        L.fxo_parse_fastq.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_uint64, C.c_uint32] + [C.c_void_p] * 7
        L.fxo_format_fastq.restype = C.c_size_t
        L.fxo_format_fastq.argtypes = [C.c_char_p] + [C.c_void_p] * 8 + [C.c_uint64, C.c_void_p]
        L.fxo_qstats_new.restype = C.c_void_p
        L.fxo_qstats_free.argtypes = [C.c_void_p]
        L.fxo_qstats_add.argtypes = [C.c_void_p, C.POINTER(Batch), C.c_int]
        L.fxo_qstats_format.restype = C.c_size_t
        L.fxo_qstats_format.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.fxo_qstats_hist.restype = C.c_longlong
        L.fxo_qstats_hist.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _LIB = L
    return _LIB


def ref_binary():
    """Path of the static driver around the real reference libfastx, or None if it was never built."""
    p = os.path.join(_HERE, "_ref", "fxref")
    return p if os.path.exists(p) else None


def _ptr(a):
    return None if a is None else a.ctypes.data


def synth_batch(seed, first, n, L, with_adapter=False, stride=None, want_qual=True):
    stride = stride or L
    bases = np.zeros((n, stride), dtype=np.uint8)
    qual = np.zeros((n, stride), dtype=np.uint8) if want_qual else None
    lib().fxo_synth_batch(seed, first, n, L, int(with_adapter), _ptr(bases), _ptr(qual), stride)
    return bases, qual


def synth_fastq(seed, first, n, L, with_adapter=False):
    sz = lib().fxo_synth_fastq(seed, first, n, L, int(with_adapter), None)
    buf = C.create_string_buffer(sz)
    lib().fxo_synth_fastq(seed, first, n, L, int(with_adapter), buf)
    return buf.raw[:sz]


def aligner_new():
    """One reference aligner = one fastx_clipper process (its query buffer survives from batch to batch, N3)."""
    return lib().fxo_aligner_new()


def aligner_free(a):
    lib().fxo_aligner_free(a)


def run_pipeline(bases, qual, lens, params, fixed_len=None, aligner=None):
    """bases/qual: uint8 [n, stride]; lens: uint16 [n] or None (then fixed_len). Returns a dict of numpy arrays."""
    n, stride = bases.shape
    assert bases.flags.c_contiguous and (qual is None or qual.flags.c_contiguous)
    if lens is not None:
        lens = np.ascontiguousarray(lens, dtype=np.uint16)
    b = Batch(_ptr(bases), _ptr(qual), _ptr(lens), int(fixed_len or stride), stride, n)
    res = np.zeros(n, dtype=np.uint32)
    ob = np.zeros(n * stride + 16, dtype=np.uint8)
    oq = np.zeros(n * stride + 16, dtype=np.uint8)
    ol = np.zeros(n, dtype=np.uint16)
    ki = np.zeros(n, dtype=np.uint32)
    o = Out(_ptr(res), _ptr(ob), _ptr(oq), _ptr(ol), _ptr(ki))
    rc = lib().fxo_run_pipeline_h(C.byref(b), C.byref(params), C.byref(o), aligner)
    if rc != 0:
        raise ValueError("fxo_run_pipeline rc=%d" % rc)
    counters = np.array(list(o.counters), dtype=np.uint64)
    kept, nbytes = int(counters[C_KEPT]), int(counters[C_KEPT_BASES])
    return dict(res=res, out_bases=ob[:nbytes], out_qual=oq[:nbytes] if qual is not None else None,
                out_len=ol[:kept], kept_index=ki[:kept], counters=counters)


def align(query, target, aligner=None):
    own = aligner is None
    a = aligner or lib().fxo_aligner_new()
    r = AlignRes()
    lib().fxo_align(a, query, len(query), target, len(target), C.byref(r))
    if own:
        lib().fxo_aligner_free(a)
    return r


def parse_fastq(text, qoffset=33, stride=None, max_reads=None):
    max_reads = max_reads or (text.count(b"\n") // 4 + 1)
    if stride is None:
        stride = max((len(l) for l in text.split(b"\n")[1::4]), default=1) or 1
    bases = np.zeros((max_reads, stride), dtype=np.uint8)
    qual = np.zeros((max_reads, stride), dtype=np.uint8)
    lens = np.zeros(max_reads, dtype=np.uint16)
    no, nl = np.zeros(max_reads, dtype=np.uint64), np.zeros(max_reads, dtype=np.uint32)
    n2o, n2l = np.zeros(max_reads, dtype=np.uint64), np.zeros(max_reads, dtype=np.uint32)
    n = lib().fxo_parse_fastq(text, len(text), qoffset, max_reads, stride, _ptr(bases), _ptr(qual), _ptr(lens),
                              _ptr(no), _ptr(nl), _ptr(n2o), _ptr(n2l))
    if n < 0:
        raise ValueError("invalid FASTQ at line %d" % -n)
    return dict(n=n, bases=bases[:n], qual=qual[:n], lens=lens[:n], names=(no[:n], nl[:n], n2o[:n], n2l[:n]), stride=stride)


def format_fastq(text, names, out_bases, out_qual, out_len, kept_index):
    no, nl, n2o, n2l = names
    kept = len(kept_index)
    cap = int(out_len.astype(np.int64).sum()) * 2 + int(nl.astype(np.int64).sum() + n2l.astype(np.int64).sum()) + 6 * kept + 16
    dst = C.create_string_buffer(cap)
    ob = np.ascontiguousarray(out_bases)
    oq = np.ascontiguousarray(out_qual)
    ol = np.ascontiguousarray(out_len, dtype=np.uint16)
    ki = np.ascontiguousarray(kept_index, dtype=np.uint32)
    w = lib().fxo_format_fastq(text, _ptr(no), _ptr(nl), _ptr(n2o), _ptr(n2l), _ptr(ob), _ptr(oq), _ptr(ol), _ptr(ki), kept, dst)
    return dst.raw[:w]


QS_MINQ, QS_RANGE = -15, 108


class QStats:
    """fastx_quality_stats restated: add() batches, then text() / hist()."""

    def __init__(self):
        self.h = lib().fxo_qstats_new()

    def close(self):
        if self.h:
            lib().fxo_qstats_free(self.h)
            self.h = None

    def add(self, bases, qual, lens, qoffset=33, fixed_len=None):
        n, stride = bases.shape
        if lens is not None:
            lens = np.ascontiguousarray(lens, dtype=np.uint16)
        b = Batch(_ptr(bases), _ptr(qual), _ptr(lens), int(fixed_len or stride), stride, n)
        lib().fxo_qstats_add(self.h, C.byref(b), qoffset)

    def text(self, new_format=False):
        sz = lib().fxo_qstats_format(self.h, int(new_format), None)
        buf = C.create_string_buffer(sz + 1)
        lib().fxo_qstats_format(self.h, int(new_format), buf)
        return buf.raw[:sz]

    def device_layout(self, ncols, qoffset=33, classes=5, bins=128):
        """hist[col][A,C,G,T,N][quality byte] as include/fxg.h's fxg_run_quality_stats defines it."""
        out = np.zeros((ncols, classes, bins), dtype=np.uint64)
        h = (C.c_int * QS_RANGE)()
        for c in range(ncols):
            for k in range(classes):
                lib().fxo_qstats_hist(self.h, c, k + 1, h)
                for v in range(QS_RANGE):
                    if h[v]:
                        out[c, k, v + QS_MINQ + qoffset] = h[v]
        return out
