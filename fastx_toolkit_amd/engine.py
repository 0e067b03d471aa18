"""Thin Python view of the C-ABI in include/fxg.h.

PyTorch is used only as plumbing: device allocations (tensors), the current HIP stream and
torch.distributed.  All compute happens inside libfxg.so's HIP kernels; there is no CPU or
PyTorch fallback -- if the library or a GPU is missing, construction fails.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

STAGE_CLIP, STAGE_QTRIM, STAGE_QFILTER, STAGE_REVCOMP, STAGE_FTRIM, STAGE_FTRIM_END, STAGE_MASK, STAGE_ARTIFACTS, STAGE_NFILTER = 1, 2, 4, 8, 16, 32, 64, 128, 256
CLIP_DISCARD_NON_CLIPPED, CLIP_DISCARD_CLIPPED, CLIP_KEEP_N, CLIP_ADAPTER_ONLY = 1, 2, 4, 8
NCOUNTERS = 24
(C_INPUT, C_KEPT, C_KEPT_BASES, C_CLIP_TOO_SHORT, C_CLIP_ADAPTER_ONLY, C_CLIP_NO_ADAPTER, C_CLIP_ADAPTER_FOUND,
 C_CLIP_N, C_QTRIM_DROPPED, C_QFILTER_DROPPED, C_FTRIM_DROPPED, C_CLIP_OUT, C_QTRIM_OUT) = range(13)
C_ERRORS = 15

EXPORTS = [
    "fxg_abi_version", "fxg_ctx_create", "fxg_ctx_destroy", "fxg_last_error", "fxg_set_stream", "fxg_sync",
    "fxg_device_info", "fxg_malloc_device", "fxg_free_device", "fxg_malloc_host", "fxg_free_host", "fxg_memcpy_h2d",
    "fxg_memcpy_d2h", "fxg_memset_device", "fxg_timer_start", "fxg_timer_stop", "fxg_run_pipeline",
    "fxg_run_qtrim_qfilter", "fxg_run_clip", "fxg_run_revcomp_trim", "fxg_read_counters", "fxg_scan_recoveries", "fxg_synth_generate",
    "fxg_last_launch_info", "fxg_set_profiling", "fxg_last_kernel_ms", "fxg_profiled_kernel_ms", "fxg_set_clip_history", "fxg_run_quality_stats",
    "fxg_fastq_index", "fxg_fastq_pack", "fxg_fastq_format", "fxg_fasta_weights", "fxg_host_register", "fxg_host_unregister",
    "fxg_shard_range", "fxg_epilogue", "fxg_concat_pwrite", "fxg_concat_peer", "fxg_device_count", "fxg_device_numa_node", "fxg_comm_create", "fxg_comm_destroy", "fxg_epilogue_rccl",
]


class FxgParams(C.Structure):
    _fields_ = [
        ("stages", C.c_uint32), ("qoffset", C.c_int32),
        ("qt_threshold", C.c_int32), ("qt_min_len", C.c_int32),
        ("qf_min_quality", C.c_int32), ("qf_min_percent", C.c_int32),
        ("adapter", C.c_char * 100), ("clip_min_len", C.c_uint32), ("clip_keep_delta", C.c_int32),
        ("clip_min_adapter_len", C.c_int32), ("clip_flags", C.c_uint32),
        ("ft_first", C.c_int32), ("ft_last", C.c_int32), ("ft_trim_end", C.c_uint32), ("ft_min_len", C.c_uint32),
        ("mask_min_quality", C.c_int32), ("mask_char", C.c_uint32), ("nf_keep_n", C.c_uint32),
    ]


class FxgBatch(C.Structure):
    _fields_ = [("bases", C.c_void_p), ("qual", C.c_void_p), ("len", C.c_void_p),
                ("fixed_len", C.c_uint32), ("stride", C.c_uint32), ("n", C.c_uint64)]


class FxgTextInfo(C.Structure):
    _fields_ = [("lines", C.c_uint64), ("records", C.c_uint64), ("consumed", C.c_uint64), ("max_len", C.c_uint32),
                ("min_len", C.c_uint32), ("irregular", C.c_uint32), ("first_bad", C.c_uint32), ("numeric_records", C.c_uint32),
                ("has_cr", C.c_uint32)]


class FxgOut(C.Structure):
    _fields_ = [("res", C.c_void_p), ("out_bases", C.c_void_p), ("out_qual", C.c_void_p), ("out_len", C.c_void_p),
                ("kept_index", C.c_void_p), ("out_off", C.c_void_p), ("counters", C.c_void_p)]


def make_params(stages=0, qoffset=33, qt_threshold=0, qt_min_len=0, qf_min_quality=0, qf_min_percent=0,
                adapter=b"CCTTAAGG", clip_min_len=5, clip_keep_delta=0, clip_min_adapter_len=0, clip_flags=0,
                ft_first=1, ft_last=0, ft_trim_end=0, ft_min_len=0, mask_min_quality=10, mask_char="N", nf_keep_n=0):
    """fxg_params with the reference tools' defaults (fastx_args.c:43, fastx_clipper.cpp:68-69)."""
    if isinstance(adapter, str):
        adapter = adapter.encode()
    p = FxgParams()
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


_LIB = None


def load_library(path=None):
    """dlopen libfxg.so (building it in-tree if needed) and declare the prototypes.  Needs no GPU."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    # One HIP runtime per process: PyTorch ships its own libamdhip64 and the Python harness always ends up with torch imported (device tensors,
    # streams).  If libfxg.so were dlopen()ed FIRST it would bring in the system's runtime and torch a second one afterwards -- two runtimes, and every
    # HIP call behind the C-ABI fails (seen as fxg_ctx_create() = FXG_E_HIP when __graft_entry__.build() had loaded the library before smoke()
    # imported torch).  So torch goes first wherever it exists; a C host (host/, the tools) links the system runtime and never sees torch.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    so = path or os.environ.get("FXG_LIB") or _build.LIBFXG   # FXG_LIB: A/B experiments with alternative builds
    if not os.path.exists(so):
        _build.build_engine()
    if not os.path.exists(so):
        raise RuntimeError("libfxg.so is missing and could not be built; there is no fallback path")
    L = C.CDLL(so)
    vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
    L.fxg_abi_version.restype = C.c_int
    L.fxg_ctx_create.argtypes = [i32, C.POINTER(vp)]
    L.fxg_ctx_destroy.argtypes = [vp]; L.fxg_ctx_destroy.restype = None
    L.fxg_last_error.argtypes = [vp]; L.fxg_last_error.restype = C.c_char_p
    L.fxg_set_stream.argtypes = [vp, vp]
    L.fxg_sync.argtypes = [vp]
    L.fxg_device_info.argtypes = [vp, C.POINTER(i32), C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]
    L.fxg_malloc_device.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.fxg_free_device.argtypes = [vp, vp]
    L.fxg_malloc_host.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.fxg_free_host.argtypes = [vp, vp]
    L.fxg_memcpy_h2d.argtypes = [vp, vp, vp, C.c_size_t]
    L.fxg_memcpy_d2h.argtypes = [vp, vp, vp, C.c_size_t]
    L.fxg_memset_device.argtypes = [vp, vp, i32, C.c_size_t]
    L.fxg_timer_start.argtypes = [vp]
    L.fxg_timer_stop.argtypes = [vp, C.POINTER(C.c_float)]
    L.fxg_run_pipeline.argtypes = [vp, C.POINTER(FxgBatch), C.POINTER(FxgParams), C.POINTER(FxgOut)]
    L.fxg_run_qtrim_qfilter.argtypes = [vp, C.POINTER(FxgBatch), i32, i32, i32, i32, i32, i32, i32, C.POINTER(FxgOut)]
    L.fxg_run_clip.argtypes = [vp, C.POINTER(FxgBatch), C.c_char_p, u32, i32, i32, u32, C.POINTER(FxgOut)]
    L.fxg_run_revcomp_trim.argtypes = [vp, C.POINTER(FxgBatch), i32, i32, i32, C.POINTER(FxgOut)]
    L.fxg_read_counters.argtypes = [vp, vp, C.POINTER(u64 * NCOUNTERS)]
    L.fxg_scan_recoveries.argtypes = [vp]
    L.fxg_synth_generate.argtypes = [vp, u64, u64, u64, u32, i32, vp, vp, u32]
    L.fxg_last_launch_info.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
    L.fxg_fastq_index.argtypes = [vp, vp, u64, i32, i32, vp, u64, vp, vp, C.POINTER(FxgTextInfo)]
    L.fxg_fastq_pack.argtypes = [vp, vp, u64, i32, vp, u64, vp, u64, u32, i32, vp, vp, C.POINTER(u32)]
    L.fxg_fastq_format.argtypes = [vp, vp, i32, vp, u64, vp, u64, vp, u32, i32, vp, vp, vp, vp, u32, i32, i32, vp, C.POINTER(u64)]
    L.fxg_fasta_weights.argtypes = [vp, vp, vp, u64, u64, vp, C.POINTER(u64 * 8)]
    L.fxg_host_register.argtypes = [vp, vp, C.c_size_t]
    L.fxg_host_unregister.argtypes = [vp, vp]
    L.fxg_shard_range.argtypes = [u64, u32, u32, C.POINTER(u64), C.POINTER(u64)]
    L.fxg_epilogue.argtypes = [vp, u32, u32, vp, C.POINTER(u64), C.POINTER(u64)]
    L.fxg_concat_pwrite.argtypes = [i32, vp, u64, u64]
    L.fxg_concat_peer.argtypes = [vp, vp, u64, vp, vp, u64]
    L.fxg_comm_create.argtypes = [vp, C.c_char_p, u32, u32, i32, C.POINTER(vp)]
    L.fxg_comm_destroy.argtypes = [vp]; L.fxg_comm_destroy.restype = None
    L.fxg_epilogue_rccl.argtypes = [vp, vp, vp, vp, C.POINTER(u64), C.POINTER(u64), vp]
    L.fxg_set_profiling.argtypes = [vp, i32]
    L.fxg_set_clip_history.argtypes = [vp, i32]
    L.fxg_run_quality_stats.argtypes = [vp, C.POINTER(FxgBatch), vp, C.c_uint32]
    L.fxg_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.fxg_profiled_kernel_ms.argtypes = [vp, C.POINTER(C.c_float), u32, C.POINTER(u32)]
    if path is None:
        _LIB = L
    return L


class TextIndex:
    """Device line index of a block of text (fxg_fastq_index): line starts, line ends after chomp, per-record flags."""

    def __init__(self, line, cap_lines, flags, lpr):
        self.line, self.cap_lines, self.flags, self.lpr = line, cap_lines, flags, lpr

    @property
    def starts(self):
        return self.line[:self.cap_lines]

    @property
    def ends(self):
        return self.line[self.cap_lines:]


class FxgError(RuntimeError):
    pass


def stream_checksum(torch, res, out_bases, out_qual, nbytes, chunk=1 << 27):
    """sum(byte[i] * (1 + i % 251)) over the packed bases, the packed qualities and the bytes of res[], folded into one 63-bit number.
    The weights make it sensitive to a shifted or permuted stream; int64 wrap-around is deterministic."""
    M = (1 << 63) - 1

    def one(t, n):
        acc = 0
        for lo in range(0, n, chunk):
            hi = min(n, lo + chunk)
            w = (torch.arange(lo, hi, device=t.device, dtype=torch.int64) % 251) + 1
            acc = (acc + int((t[lo:hi].to(torch.int64) * w).sum().item())) & M
        return acc
    cs = one(res.view(torch.uint8), res.numel() * res.element_size())
    for t in (out_bases, out_qual):
        if t is not None:
            cs = (cs * 1000003 + one(t, nbytes)) & M
    return cs


class Result:
    """Device-resident outputs of one pipeline run (torch tensors) plus lazily fetched host counters."""

    def __init__(self, engine, res, out_bases, out_qual, out_len, kept_index, out_off, counters):
        self.engine = engine
        self.res, self.out_bases, self.out_qual = res, out_bases, out_qual
        self.out_len, self.kept_index, self.out_off = out_len, kept_index, out_off
        self.d_counters = counters
        self._counters = None

    @property
    def counters(self):
        if self._counters is None:
            self._counters = self.engine.read_counters(self.d_counters)
        return self._counters

    @property
    def kept(self):
        return int(self.counters[C_KEPT])

    @property
    def kept_bytes(self):
        return int(self.counters[C_KEPT_BASES])

    def checksum(self):
        """Position-weighted 63-bit checksum of the packed output stream (bases, then qualities) and of res[], computed on the device:
        what bench.py prints for its timed launches and tests/test_gpu_parity.py pins against an oracle-verified run."""
        return stream_checksum(self.engine.torch, self.res, self.out_bases, self.out_qual, self.kept_bytes)

    def to_host(self):
        """numpy copies trimmed to the kept counts (same keys as the oracle's run_pipeline)."""
        c = self.counters
        kept, nbytes = int(c[C_KEPT]), int(c[C_KEPT_BASES])
        d = dict(res=self.res.cpu().numpy().view(np.uint32), counters=c)
        d["out_bases"] = self.out_bases[:nbytes].cpu().numpy() if self.out_bases is not None else None
        d["out_qual"] = self.out_qual[:nbytes].cpu().numpy() if self.out_qual is not None else None
        d["out_len"] = self.out_len[:kept].cpu().numpy().view(np.uint16) if self.out_len is not None else None
        d["kept_index"] = self.kept_index[:kept].cpu().numpy().view(np.uint32) if self.kept_index is not None else None
        d["out_off"] = self.out_off[:kept].cpu().numpy().view(np.uint64) if self.out_off is not None else None
        return d


class Engine:
    """One context on one GPU (one process per GPU)."""

    def __init__(self, device_id=0):
        import torch
        if not torch.cuda.is_available():
            raise FxgError("no HIP device visible: the fastx engine has no CPU fallback")
        self.torch = torch
        self.lib = load_library()
        self.device_id = device_id
        self.device = torch.device("cuda", device_id)
        h = C.c_void_p()
        rc = self.lib.fxg_ctx_create(device_id, C.byref(h))
        if rc != 0:
            raise FxgError("fxg_ctx_create(%d) failed: %d" % (device_id, rc))
        self.ctx = h
        self.use_torch_stream()

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.fxg_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise FxgError("fxg error %d: %s" % (rc, self.lib.fxg_last_error(self.ctx).decode(errors="replace")))

    def use_torch_stream(self):
        """Enqueue on torch's current stream so tensors allocated by torch are ordered with the kernels.

        torch's DEFAULT stream has the null handle, which the C-ABI cannot adopt.  The engine then runs on a torch SIDE stream
        and every launch is bracketed by stream waits in both directions (`_after_torch` / `_before_torch`): the kernels see what
        torch enqueued before them, and whatever torch enqueues afterwards (including frees and reuse of the tensors by the
        caching allocator, which follow the order of torch's stream) comes after the kernels.  No host synchronisation either way.
        Run under `torch.cuda.stream(s)` / `torch.cuda.set_stream` (as bench.py does) to share one stream outright."""
        cur = self.torch.cuda.current_stream(self.device)
        self._side = None if cur.cuda_stream else self.torch.cuda.Stream(device=self.device)
        s = (self._side or cur).cuda_stream
        self._check(self.lib.fxg_set_stream(self.ctx, C.c_void_p(s)))

    def _after_torch(self):
        if self._side is not None:
            self._side.wait_stream(self.torch.cuda.current_stream(self.device))

    def _before_torch(self):
        if self._side is not None:
            self.torch.cuda.current_stream(self.device).wait_stream(self._side)

    def sync(self):
        self._check(self.lib.fxg_sync(self.ctx))

    def device_info(self):
        cus, mem = C.c_int(), C.c_size_t()
        name = C.create_string_buffer(128)
        self._check(self.lib.fxg_device_info(self.ctx, C.byref(cus), C.byref(mem), name, 128))
        return dict(compute_units=cus.value, total_mem=mem.value, name=name.value.decode())

    def scan_recoveries(self):
        """Compacting launches of this context whose waits ran out and that were done again without the scanner (include/fxg.h)."""
        return int(self.lib.fxg_scan_recoveries(self.ctx))

    def last_launch(self):
        name = C.create_string_buffer(128)
        g, b, l, t = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._check(self.lib.fxg_last_launch_info(self.ctx, name, 128, C.byref(g), C.byref(b), C.byref(l), C.byref(t)))
        return dict(kernel=name.value.decode(), grid=g.value, block=b.value, lds=l.value, tile_reads=t.value)

    def quality_stats(self, bases, qual, lens=None, fixed_len=None, hist=None, cols=None, sync=True):
        """fastx_quality_stats: adds the batch to hist[cols][5][128] (int64 device tensor, created zeroed when None) and returns it.
        sync=False leaves the kernel in flight (torch work enqueued afterwards is still ordered behind it)."""
        n, stride = bases.shape
        if hist is None:
            hist = self.torch.zeros((cols or stride, 5, 128), dtype=self.torch.int64, device=self.device)
        self._after_torch()                                   # e.g. the fill above ran on torch's stream
        b = FxgBatch(bases.data_ptr(), qual.data_ptr() if qual is not None else None,
                     lens.data_ptr() if lens is not None else None, int(fixed_len or stride), stride, n)
        self._check(self.lib.fxg_run_quality_stats(self.ctx, C.byref(b), hist.data_ptr(), hist.shape[0]))
        self._before_torch()
        if sync:
            self.sync()
        return hist

    def set_clip_history(self, on=True):
        """Reference-exact clipping of variable-length input: reads form one sequence across run() calls (fxg.h)."""
        self._check(self.lib.fxg_set_clip_history(self.ctx, int(on)))

    def set_profiling(self, on=True):
        self._check(self.lib.fxg_set_profiling(self.ctx, int(on)))

    def last_kernel_ms(self):
        ms = C.c_float()
        self._check(self.lib.fxg_last_kernel_ms(self.ctx, C.byref(ms)))
        return ms.value

    def profiled_kernel_ms(self, cap=64):
        """Durations (ms) of the last profiled launches, oldest first (waits for them): the timed loop's own launches."""
        ms, n = (C.c_float * cap)(), C.c_uint32()
        self._check(self.lib.fxg_profiled_kernel_ms(self.ctx, ms, cap, C.byref(n)))
        return [float(ms[i]) for i in range(n.value)]

    def timer_start(self):
        self._check(self.lib.fxg_timer_start(self.ctx))

    def timer_stop(self):
        ms = C.c_float()
        self._check(self.lib.fxg_timer_stop(self.ctx, C.byref(ms)))
        return ms.value

    # ---- data ----
    def empty(self, n, dtype=None):
        return self.torch.empty(int(n), dtype=dtype or self.torch.uint8, device=self.device)

    def synth(self, seed, first, n, read_len, with_adapter=False, stride=None, want_qual=True):
        """Deterministic synthetic reads generated on the device. Returns (bases, qual) uint8 [n, stride]."""
        stride = stride or read_len
        bases = self.torch.empty((n, stride), dtype=self.torch.uint8, device=self.device)
        qual = self.torch.empty((n, stride), dtype=self.torch.uint8, device=self.device) if want_qual else None
        self._after_torch()
        self._check(self.lib.fxg_synth_generate(self.ctx, seed, first, n, read_len, int(with_adapter), bases.data_ptr(),
                                                qual.data_ptr() if want_qual else None, stride))
        self._before_torch()
        return bases, qual

    def upload(self, arr):
        t = self.torch.from_numpy(np.ascontiguousarray(arr).view(np.uint8).reshape(-1))
        return t.to(self.device)

    def alloc_outputs(self, n, stride, compact=True, meta=True, has_qual=True):
        t = self.torch
        o = dict(res=t.empty(n, dtype=t.int32, device=self.device), counters=t.zeros(NCOUNTERS, dtype=t.int64, device=self.device))
        o["out_bases"] = t.empty(n * stride + 16, dtype=t.uint8, device=self.device) if compact else None
        o["out_qual"] = t.empty(n * stride + 16, dtype=t.uint8, device=self.device) if (compact and has_qual) else None
        o["out_len"] = t.empty(n, dtype=t.int16, device=self.device) if (compact and meta) else None
        o["kept_index"] = t.empty(n, dtype=t.int32, device=self.device) if (compact and meta) else None
        o["out_off"] = t.empty(n, dtype=t.int64, device=self.device) if (compact and meta) else None
        return o

    def run(self, bases, qual, params, lens=None, fixed_len=None, compact=True, meta=True, outputs=None):
        """Enqueue one pipeline pass.  bases/qual: uint8 device tensors [n, stride]; lens: int16 device tensor or None."""
        n, stride = bases.shape
        if outputs is None:
            outputs = self.alloc_outputs(n, stride, compact, meta, qual is not None)
        self._after_torch()
        o = outputs
        b = FxgBatch(bases.data_ptr(), qual.data_ptr() if qual is not None else None,
                     lens.data_ptr() if lens is not None else None, int(fixed_len or stride), stride, n)
        ptr = lambda k: (o[k].data_ptr() if o.get(k) is not None else None)
        fo = FxgOut(ptr("res"), ptr("out_bases"), ptr("out_qual"), ptr("out_len"), ptr("kept_index"), ptr("out_off"), ptr("counters"))
        self._check(self.lib.fxg_run_pipeline(self.ctx, C.byref(b), C.byref(params), C.byref(fo)))
        self._before_torch()
        return Result(self, o["res"], o.get("out_bases"), o.get("out_qual"), o.get("out_len"), o.get("kept_index"),
                      o.get("out_off"), o["counters"])

    # ---- FASTQ text on the device ----
    def text_upload(self, text, at_eof=True):
        """Copy a block of FASTQ text to the device (16 bytes of slack; a final newline is appended at end of input)."""
        if at_eof and text and not text.endswith(b"\n"):
            text = text + b"\n"
        t = self.torch.zeros(len(text) + 16, dtype=self.torch.uint8, device=self.device)
        t[:len(text)] = self.torch.frombuffer(bytearray(text), dtype=self.torch.uint8).to(self.device)
        return t, len(text)

    def fastq_index(self, d_text, text_len, at_eof=True, cap_records=None, fasta=False):
        """Index a block of FASTQ (or two-line FASTA) text.  Returns (TextIndex, lens, info); TextIndex.starts / .ends are the
        line starts and the line ends after chomp, TextIndex.flags the per-record flags (bit 0: numeric quality line)."""
        lpr = 2 if fasta else 4
        cap_records = cap_records or (text_len // (4 if fasta else 7) + 2)
        cap_lines = lpr * cap_records + 1
        line = self.torch.zeros(2 * cap_lines, dtype=self.torch.int32, device=self.device)
        lens = self.torch.empty(cap_records, dtype=self.torch.int16, device=self.device)
        flags = self.torch.zeros(cap_records, dtype=self.torch.uint8, device=self.device)
        info = FxgTextInfo()
        self._after_torch()
        self._check(self.lib.fxg_fastq_index(self.ctx, d_text.data_ptr(), text_len, int(at_eof), lpr, line.data_ptr(), cap_lines, lens.data_ptr(),
                                             flags.data_ptr(), C.byref(info)))
        return TextIndex(line, cap_lines, flags, lpr), lens, info

    def fastq_pack(self, d_text, text_len, ix, n, stride, qoffset=33, want_qual=True):
        nbytes = (n * stride + 15) // 16 * 16
        want_qual = want_qual and ix.lpr == 4
        bases = self.torch.empty(nbytes, dtype=self.torch.uint8, device=self.device)
        qual = self.torch.empty(nbytes, dtype=self.torch.uint8, device=self.device) if want_qual else None
        irr = C.c_uint32()
        self._after_torch()
        self._check(self.lib.fxg_fastq_pack(self.ctx, d_text.data_ptr(), text_len, ix.lpr, ix.line.data_ptr(), ix.cap_lines, ix.flags.data_ptr(), n, stride,
                                            qoffset, bases.data_ptr(), qual.data_ptr() if want_qual else None, C.byref(irr)))
        return bases[:n * stride].view(n, stride), (qual[:n * stride].view(n, stride) if want_qual else None), irr.value

    def fastq_format(self, d_text, text_len, ix, n, res, fwd_start=0, packed=None, reverse=False, rows_qual=None, qoffset=33, out_fasta=False):
        out = self.torch.empty(text_len + n + 16, dtype=self.torch.uint8, device=self.device)
        nb = C.c_uint64()
        pb, pq, po = (packed[0].data_ptr(), packed[1].data_ptr() if packed[1] is not None else None, packed[2].data_ptr()) if packed else (None, None, None)
        self._after_torch()
        self._check(self.lib.fxg_fastq_format(self.ctx, d_text.data_ptr(), ix.lpr, ix.line.data_ptr(), ix.cap_lines, ix.flags.data_ptr(), n, res.data_ptr(),
                                              fwd_start, int(reverse), pb, pq, po, rows_qual.data_ptr() if rows_qual is not None else None,
                                              rows_qual.shape[1] if rows_qual is not None else 0, qoffset, int(out_fasta), out.data_ptr(), C.byref(nb)))
        return out[:nb.value]

    def fasta_weights(self, d_text, ix, n, res):
        w = (C.c_uint64 * 8)()
        self._after_torch()
        self._check(self.lib.fxg_fasta_weights(self.ctx, d_text.data_ptr(), ix.line.data_ptr(), ix.cap_lines, n, res.data_ptr(), C.byref(w)))
        return list(w)

    def read_counters(self, d_counters):
        host = (C.c_uint64 * NCOUNTERS)()
        self._check(self.lib.fxg_read_counters(self.ctx, d_counters.data_ptr() if d_counters is not None else None, C.byref(host)))
        return np.array(list(host), dtype=np.uint64)
