"""Plan compiler: turns the hot path into a STATIC list of C-ABI launches over pre-allocated device buffers.

Every Recorder method allocates its output once (PyTorch is only the device allocator here), builds the argument block once and
appends ONE record to a segment of a `BcPlan` - the plan runtime of libblobctrl_hip (csrc/plan.hip, include/blobctrl_hip.h).
Nothing of the replay is Python: `Segment.run` is one `bc_step` call (an eager loop over the records, or one `hipGraphLaunch` once
the segment was captured), `Recorder.capture_loop` turns a whole sequence of segments (the N-step denoise loop) into ONE graph, and
`Recorder.save` writes the relocatable `.bcplan` a plain C host can load and run (`bc_plan_load`).
"""
import ctypes as C
import json
import os
import struct
from typing import List, Optional

import torch

from . import _lib
from .options import opt
from ._lib import BcGemm


_TUNING_PATH = os.environ.get("BC_TUNING_FILE") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_tuning.json")
_TUNING = None
_M64 = (1 << 64) - 1


def tuning_table():
    """Measured best (tile_cfg, splitk) per GEMM shape, written by tools/tune_gemm.py on an MI355X (optional)."""
    global _TUNING
    if _TUNING is None:
        _TUNING = {}
        if os.path.exists(_TUNING_PATH) and not os.environ.get("BC_NO_TUNING"):
            with open(_TUNING_PATH) as f:
                _TUNING = json.load(f).get("shapes", {})
    return _TUNING


def decode_gn_tot(tot: torch.Tensor) -> torch.Tensor:
    """GroupNorm statistics totals [..., GN_TOT_WORDS] int64 (include/blobctrl_hip.h) -> (sum, sum of squares) [..., 2] float64."""
    t = tot.to(torch.float64).cpu()
    w = torch.tensor([2.0 ** -60, 2.0 ** -20, 2.0 ** 20], dtype=torch.float64)
    out = torch.stack([(t[..., 0:3] * w).sum(-1), (t[..., 3:6] * w).sum(-1)], -1)
    poisoned = (tot[..., 2].cpu().abs() >= 2 ** 45) | (tot[..., 5].cpu().abs() >= 2 ** 45)      # bc_gn_tot_read's rule
    out[poisoned] = float("nan")
    return out


def gn_tot_slots(sums: torch.Tensor) -> torch.Tensor:
    """Per-channel (sum, sum of squares) [..., C, 2] -> per totals BLOCK [..., C / cg, 2] (csrc/bc_common.h bc_gn_cg: a producer's table whose
    width is a multiple of 320 holds sums of 10-channel blocks, spread over the block's ten slots; tests compare block sums on both sides)."""
    C = sums.shape[-2]
    cg = 10 if C % 320 == 0 else 1
    return sums.reshape(*sums.shape[:-2], C // cg, cg, 2).sum(-2)


def encode_gn_tot(sums: torch.Tensor) -> torch.Tensor:
    """(sum, sum of squares) [..., 2] -> totals [..., GN_TOT_WORDS] int64, sliced the way the kernels do it (csrc/bc_common.h
    bc_gn_slices): tools and tests that stand in for a producer."""
    d = sums.to(torch.float64)
    h = torch.floor(d * 2.0 ** -20)
    r = d - h * 2.0 ** 20
    m = torch.floor(r * 2.0 ** 20)
    r2 = r - m * 2.0 ** -20
    sl = torch.stack([torch.floor(r2 * 2.0 ** 60), m, h], -1).to(torch.int64)          # [..., 2, 3]
    return sl.reshape(*sums.shape[:-1], 6).contiguous()


def ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream_array(stream, side=None, extra=()):
    """Stream of every recorded stream id: 0 = `stream`, 1 = `side`, 2.. = `extra`; missing ones fall back to `stream`."""
    ss = [stream, side if side is not None else stream] + [e if e is not None else stream for e in extra]
    return (C.c_void_p * len(ss))(*ss), len(ss)


_MODULE_STREAMS = {}


def run_graphed(seg: "Segment", device):
    """Replay a module's forward segment (VAE / CLIP / DINOv2 / trunk shells) as ONE hipGraph launch, ordered after the caller's
    current stream and before whatever the caller enqueues next.  First call: eager warm-up + capture on a private stream.
    (A ViT-L forward is ~300 dependent launches of a few microseconds each: replayed one by one from Python it took 64 ms.)"""
    device = torch.device(device)
    cur = torch.cuda.current_stream(device)
    cs = _MODULE_STREAMS.get(device)
    if cs is None:
        cs = _MODULE_STREAMS[device] = torch.cuda.Stream(device)
    cs.wait_stream(cur)
    if not seg.captured and not os.environ.get("BC_NO_MODULE_GRAPHS"):
        seg.run(cs.cuda_stream)                       # warm-up (lazy kernel attributes must be set outside the capture)
        cs.synchronize()
        seg.capture(cs.cuda_stream)
    seg.run(cs.cuda_stream)
    cur.wait_stream(cs)


class Segment:
    """An ordered list of launches inside a BcPlan; replayable eagerly or as a hipGraph."""

    def __init__(self, rec: "Recorder", name: str):
        self.rec, self.name = rec, name
        self.id = rec.lib.bc_plan_segment(rec.plan, name.encode())
        if self.id < 0:
            _lib.check(1, "bc_plan_segment")
        self.captured = False
        self.flops = 0          # algorithmic 2*MAC of the GEMM / attention launches recorded here
        self.kinds = {}
        self.meta = []          # per launch: dict(kind, flops, variant, shape, bytes, sid)

    def __len__(self):
        return len(self.meta)

    def run(self, stream: int, side: int = None, extra=()):
        """Replay on `stream`; launches recorded for the side streams go to `side` / `extra` (default: same stream, i.e. serial)."""
        arr, n = _stream_array(stream, side, extra)
        rc = self.rec.lib.bc_step(self.rec.plan, self.id, arr, n)
        if rc:
            _lib.check(rc, f"bc_step({self.name})")

    def capture(self, stream: int, side: int = None, extra=()):
        """Capture this segment into a hipGraph on `stream` (a non-default stream).  Side-stream work joins the capture
        through the recorded fork / signal / wait events and becomes parallel branches of the graph."""
        arr, n = _stream_array(stream, side, extra)
        _lib.check(self.rec.lib.bc_plan_capture(self.rec.plan, self.id, arr, n), f"bc_plan_capture({self.name})")
        self.captured = True

    def run_eager(self, stream: int, side: int = None, extra=()):
        """Replay the launch list itself even when a graph exists (diagnostics)."""
        if not self.captured:
            return self.run(stream, side, extra)
        self.release()
        try:
            self.run(stream, side, extra)
        finally:
            self.capture(stream, side, extra) if stream else None

    def run_timed(self, stream: int):
        """Serial eager replay with a HIP event pair around every launch (recorded on `stream`, the stream the kernels run
        on).  Returns a list of (meta, milliseconds)."""
        n = len(self.meta)
        ms = (C.c_float * max(1, n))()
        _lib.check(self.rec.lib.bc_plan_run_timed(self.rec.plan, self.id, stream, ms), "bc_plan_run_timed")
        return [(m, ms[i]) for i, m in enumerate(self.meta)]

    def run_timed_kernels(self, stream: int):
        """As run_timed, with every split-K GEMM divided into (main kernel, reducer): list of (meta, ms_main, ms_reduce)."""
        n = len(self.meta)
        a, b = (C.c_float * max(1, n))(), (C.c_float * max(1, n))()
        _lib.check(self.rec.lib.bc_plan_run_timed_kernels(self.rec.plan, self.id, stream, a, b), "bc_plan_run_timed_kernels")
        return [(m, a[i], b[i]) for i, m in enumerate(self.meta)]

    def run_marked(self, stream: int, side: int, extra, marks):
        """Concurrent eager replay on the segment's own streams with a timestamp behind each launch index in `marks` (ascending):
        milliseconds from the start of the replay, one per mark (include/blobctrl_hip.h: bc_plan_run_marked)."""
        arr, n = _stream_array(stream, side, extra)
        mk = (C.c_int * max(1, len(marks)))(*marks)
        ms = (C.c_float * max(1, len(marks)))()
        _lib.check(self.rec.lib.bc_plan_run_marked(self.rec.plan, self.id, arr, n, mk, len(marks), ms), "bc_plan_run_marked")
        return [ms[i] for i in range(len(marks))]

    def enable(self, index: int, on: bool):
        """Diagnostics (ablation probes): skip / restore one launch; re-capture afterwards."""
        _lib.check(self.rec.lib.bc_plan_enable(self.rec.plan, self.id, index, 1 if on else 0), "bc_plan_enable")

    def release(self):
        if self.captured:
            self.rec.lib.bc_plan_release(self.rec.plan, self.id)
            self.captured = False


class Recorder:
    def __init__(self, device: torch.device):
        self.lib = _lib.load()
        self.device = torch.device(device)
        h = C.c_void_p()
        _lib.check(self.lib.bc_plan_create(C.byref(h)), "bc_plan_create")
        self.plan = h
        self.seg: Optional[Segment] = None
        self.segments: List[Segment] = []
        self.keep = []                      # keeps tensors referenced by the records alive
        self._workspace = set()
        self._storages = {}                 # storage address -> uint8 view of every storage the records point into (memory map)
        self.named = {}                     # name -> tensor (I/O buffers a C host looks up with bc_plan_buffer)
        self._slab = {}                     # split-K scratch PER STREAM id (the branches run concurrently)
        self._slab_elems = {}
        self.num_cu = 256
        if self.device.type == "cuda":
            info = (C.c_int * 4)()
            _lib.check(self.lib.bc_device_info(info), "bc_device_info")
            self.num_cu = info[0]
        self.bytes_allocated = 0
        self.tots = {}                      # data_ptr of an activation -> its GroupNorm statistics totals [B][C][GN_TOT_WORDS] (int64)
        self.tots_per_channel = set()       # ids of the tables a bc_gn_stats pass (or a test standing in for a producer) filled per channel
        self._tot_chunk = {}                # (segment id, stream id) -> [chunk tensor, used words]: the arenas new tables are carved from
        self.sid = 0                        # stream id new launches are recorded for (0 main, 1 side)
        self.events = []
        self.loop_graphs = []

    def close(self):
        """Destroy the plan (graphs, events).  Buffers are torch tensors and go with the recorder."""
        if self.plan is not None:
            for g in self.loop_graphs:
                self.lib.bc_graph_destroy(g)
            self.loop_graphs = []
            self.lib.bc_plan_destroy(self.plan)
            self.plan = None

    # ------------------------------------------------------------------ buffers
    def empty(self, *shape, dtype=torch.float16, name=None):
        return self._track(torch.empty(*shape, dtype=dtype, device=self.device), name)

    def zeros(self, *shape, dtype=torch.float16, name=None):
        return self._track(torch.zeros(*shape, dtype=dtype, device=self.device), name)

    def _track(self, t, name=None):
        self.bytes_allocated += t.numel() * t.element_size()
        self._workspace.add(t.untyped_storage().data_ptr())        # the plan's own activations / per-edit inputs: no contents saved
        return self.register(t, name)

    def register(self, t: torch.Tensor, name=None):
        """Put the STORAGE behind a tensor (a fresh allocation, a weight arena a view points into, a table) on the plan's memory map,
        so that `save` can express every pointer of the launch records as (buffer, offset).  Called for every tensor a record uses."""
        if t is not None and torch.is_tensor(t):
            st = t.untyped_storage()
            key = st.data_ptr()
            if key not in self._storages and st.nbytes() > 0:
                self._storages[key] = torch.empty(0, dtype=torch.uint8, device=t.device).set_(st)
            if name:
                self.named[name] = t
        return t

    @property
    def buffers(self):
        return list(self._storages.values())

    def begin(self, name: str) -> Segment:
        self.seg = Segment(self, name)
        self.segments.append(self.seg)
        return self.seg

    # ------------------------------------------------------------------ streams / events
    def new_event(self):
        e = self.lib.bc_plan_new_event(self.plan)
        if e < 0:
            _lib.check(1, "bc_plan_new_event")
        self.events.append(e)
        return e

    def signal(self, ev):
        """Record `ev` on the stream currently being recorded for."""
        self._add_op(_lib.OP_SIGNAL, "i", (ev,), "event_record")
        self.seg.meta[-1]["ev"] = ev

    def wait(self, ev):
        """Make the stream currently being recorded for wait on `ev`."""
        self._add_op(_lib.OP_WAIT, "i", (ev,), "event_wait")
        self.seg.meta[-1]["ev"] = ev

    def _add_op(self, op, sig, args, kind, flops=0, variant="", shape=None, bytes_=0, rocprof=None):
        assert len(sig) == len(args), (op, sig, args)
        words = (C.c_uint64 * max(1, len(args)))()
        for k, (c, v) in enumerate(zip(sig, args)):
            if c == "p":
                words[k] = 0 if v is None else (self.register(v).data_ptr() if torch.is_tensor(v) else int(v))
            elif c == "f":
                words[k] = struct.unpack("<I", struct.pack("<f", float(v)))[0]
            else:
                words[k] = int(v) & _M64
        idx = self.lib.bc_plan_add_op(self.plan, self.seg.id, self.sid, op, words, len(args))
        if idx < 0 or idx != len(self.seg.meta):
            _lib.check(1, f"bc_plan_add_op({kind})")
        self._push(kind, flops, variant, shape, bytes_, rocprof)

    def _push(self, kind, flops=0, variant="", shape=None, bytes_=0, rocprof=None):
        self.seg.flops += flops
        self.seg.kinds[kind] = self.seg.kinds.get(kind, 0) + 1
        # bytes_ = algorithmic HBM bytes of the launch (operands once: each activation / weight read once + the result written once)
        # rocprof = the kernel's name as rocprofv3 prints it (template instantiation): ties bench.py's buckets to profiles/*.csv
        self.seg.meta.append(dict(kind=kind, flops=flops, variant=variant, shape=shape, bytes=bytes_, sid=self.sid, rocprof=rocprof or variant))

    def capture_loop(self, segments, stream, side=None, extra=()):
        """ONE hipGraph for a whole sequence of segments (e.g. prologue + the N denoise steps); returns the graph handle for
        `bc_graph_launch`.  Per-step scalars are read through the device step counter, so the replay takes no arguments."""
        ids = (C.c_int * len(segments))(*[sg.id for sg in segments])
        arr, n = _stream_array(stream, side, extra)
        g = C.c_void_p()
        _lib.check(self.lib.bc_plan_capture_loop(self.plan, ids, len(segments), arr, n, C.byref(g)), "bc_plan_capture_loop")
        self.loop_graphs.append(g)
        return g

    def destroy_loop_graph(self, g):
        """Destroy one whole-loop graph exec (the caller has synchronised the device) and forget its handle."""
        self.lib.bc_graph_destroy(g)
        self.loop_graphs = [h for h in self.loop_graphs if h is not g]

    def save(self, path: str):
        """Write the relocatable plan file: buffer table (named I/O buffers, weights / tables with their contents, zero-filled
        workspace) + launch records with (buffer, offset) pointers.  Loaded by `bc_plan_load` (C hosts: tests/c/plan_edit.c)."""
        for t in self._slab.values():
            self.register(t)
        bufs = self.buffers
        names = {}
        for n, t in self.named.items():
            assert t.data_ptr() == t.untyped_storage().data_ptr(), f"named buffer {n} must own its storage"
            names[t.data_ptr()] = n
        arr = (_lib.BcPlanBuffer * len(bufs))()
        hold = []
        for i, t in enumerate(bufs):
            nbytes = t.numel() * t.element_size()
            arr[i].name = names.get(t.data_ptr(), "").encode()
            arr[i].address = t.data_ptr()
            arr[i].bytes = nbytes
            if nbytes and t.data_ptr() not in self._workspace:         # weights, tables, constants: stored with their contents
                host = t.detach().cpu().contiguous()
                hold.append(host)
                arr[i].host_data = host.data_ptr()
        _lib.check(self.lib.bc_plan_save(self.plan, path.encode(), arr, len(bufs)), "bc_plan_save")

    def reserve_slab(self, elems: int):
        """Shared split-K scratch of the CURRENT stream: consumed by the reduce kernel that immediately follows on that
        stream, so one buffer per stream is enough."""
        sid = self.sid
        if elems > self._slab_elems.get(sid, 0):
            self._slab[sid] = torch.empty(elems, dtype=torch.float32, device=self.device)
            self._workspace.add(self._slab[sid].untyped_storage().data_ptr())
            self._slab_elems[sid] = elems
            _lib.check(self.lib.bc_plan_set_slab(self.plan, sid, self._slab[sid].data_ptr()), "bc_plan_set_slab")

    # ------------------------------------------------------------------ GroupNorm statistics totals
    TOT_CHUNK_WORDS = 1 << 20               # 8 MiB of int64 per (segment, stream): > all statistics tables of one trunk forward at batch 8

    def new_tot(self, B, Cc):
        """A statistics table [B][C][GN_TOT_WORDS] (int64 words, include/blobctrl_hip.h) for the producer(s) of one activation,
        carved from the current (segment, stream) chunk.  A chunk is zeroed by ONE bc_memset_zero recorded where it is first used -
        before every producer that adds to it - so a replay starts from zero totals with one extra launch per segment and stream."""
        n = B * Cc * _lib.GN_TOT_WORDS
        # one open chunk per (segment, stream): recording that switches back to a stream goes on carving from that stream's chunk
        # (a fresh 8 MiB chunk + memset per switch grew with the number of stream switches)
        key = (self.seg.id, self.sid)
        ch = self._tot_chunk.get(key)
        if ch is None or ch[1] + n > ch[0].numel():
            t = self.zeros(max(self.TOT_CHUNK_WORDS, n), dtype=torch.int64)
            self.keep.append(t)
            self._op("bc_memset_zero", (t, t.numel() * 8), "memset", variant="memset_zero", shape=("memset", t.numel() * 8))
            ch = self._tot_chunk[key] = [t, 0]
        view = ch[0][ch[1]:ch[1] + n].view(B, Cc, _lib.GN_TOT_WORDS)
        ch[1] += (n + 31) // 32 * 32        # (256-byte aligned tables)
        return view

    def dup_halves(self, items):
        """bc_dup_halves: `items` = up to six (tensor laid out [2][...], bytes of one half); the first half of each is copied over
        its second half in one launch (the CFG-invariant prefix of the UNet fanned out to the image pair, engine.record_forward)."""
        assert 0 < len(items) <= 6
        args = []
        for t, nbytes in items:
            assert nbytes % 16 == 0 and t.numel() * t.element_size() >= 2 * nbytes
            args += [t, nbytes]
        args += [None, 0] * (6 - len(items))
        self.keep.append(tuple(t for t, _ in items))
        total = sum(b for _, b in items)
        self._op("bc_dup_halves", tuple(args), "dup_halves", variant="dup_halves_kernel", shape=("dup_halves", total), bytes_=2 * total)

    # ------------------------------------------------------------------ GEMM family
    def plan_gemm(self, M, N, K, fast, mode, tile_cfg=0, splitk=None):
        """(tile_cfg, splitk, bm, bn) for a GEMM: tuning table first, then the library's cost model (bc_gemm_plan)."""
        sk = -1 if splitk is None else splitk
        if tile_cfg == 0 and splitk is None and fast:
            hit = tuning_table().get(f"{mode}|{M}|{N}|{K}")
            if hit:
                tile_cfg, sk = int(hit[0]), int(hit[1])
        c, s_, bm, bn = C.c_int(tile_cfg), C.c_int(sk), C.c_int(0), C.c_int(0)
        _lib.check(self.lib.bc_gemm_plan(M, N, K, 1 if fast else 0, C.byref(c), C.byref(s_), C.byref(bm), C.byref(bn)),
                   "bc_gemm_plan")
        return c.value, s_.value, bm.value, bn.value

    def gemm(self, *, A, W, M, N, K, out=None, out_mode=_lib.OUT_F16, ldc=None, A2=None, C1=0, lda=None, lda2=0,
             conv=None, bias=None, rowvec=None, ld_rowvec=0, rowvec_idx=None, rowvec_step=0, rows_per_batch=0, act=_lib.ACT_NONE, colscale=None,
             alpha=1.0, alpha_dev=None, alpha_idx=None, alpha_bstride=0, R=None, ldr=0, R2=None, ldr2=0, r2_xmin=0, r2_bmod=1,
             out_w=0, splitk=None, kind="gemm", a_offset=0, w_offset=0, out_offset=0, want_gn=False, tile_cfg=0, ldw=None,
             a_affine=None, a_act=_lib.ACT_NONE, a_gn=None, ln_colsum=None, ln_eps=1e-5, C_t=None, ldc_t=0, n_t0=0,
             w_bstride=0, vec_bstride=0, sm_group=0, sm_valid=0, sm_keep=0, gn_tot_out=None):
        """Record one bc_gemm.  `conv` = dict(Cin, Hin, Win, Hv, Wv, Hout, Wout, stride) for the 3x3 gather mode.
        Pointer offsets are in ELEMENTS of the respective tensor."""
        n_out = N // 2 if act == _lib.ACT_GEGLU else (N // sm_group * sm_keep if sm_group else N)
        g = BcGemm()
        g.A = A.data_ptr() + a_offset * A.element_size()
        g.A2 = ptr(A2)
        g.a_mode = _lib.A_CONV3X3 if conv else _lib.A_DENSE
        g.M, g.N, g.K = M, N, K
        g.lda = lda if lda is not None else (K if not conv else 0)
        g.lda2, g.C1 = lda2, C1
        g.a_affine, g.a_act = ptr(a_affine), a_act
        if conv:
            g.Cin, g.Hin, g.Win = conv["Cin"], conv["Hin"], conv["Win"]
            g.Hv, g.Wv = conv.get("Hv", conv["Hin"]), conv.get("Wv", conv["Win"])
            g.Hout, g.Wout, g.stride = conv["Hout"], conv["Wout"], conv.get("stride", 1)
            g.conv_nopad_lo = 1 if conv.get("nopad_lo") else 0
        g.W = W.data_ptr() + w_offset * W.element_size()
        g.ldw = ldw if ldw is not None else K
        g.bias = ptr(bias)
        g.rowvec = ptr(rowvec) if not isinstance(rowvec, int) else rowvec
        g.ld_rowvec, g.rows_per_batch = ld_rowvec, rows_per_batch
        g.rowvec_idx, g.rowvec_step = ptr(rowvec_idx), rowvec_step
        g.act = act
        g.colscale = ptr(colscale)
        g.alpha = alpha
        g.alpha_dev, g.alpha_idx, g.alpha_bstride = ptr(alpha_dev), ptr(alpha_idx), alpha_bstride
        g.R, g.ldr = (ptr(R) if not isinstance(R, int) else R), ldr
        g.R2, g.ldr2, g.r2_xmin, g.r2_bmod, g.out_w = ptr(R2), ldr2, r2_xmin, r2_bmod, out_w
        g.out_mode = out_mode
        g.C = out.data_ptr() + out_offset * out.element_size()
        g.ldc = ldc if ldc is not None else (n_t0 if C_t is not None else n_out)
        g.ln_colsum, g.ln_eps = ptr(ln_colsum), ln_eps
        g.C_t, g.ldc_t, g.n_t0 = ptr(C_t), ldc_t, n_t0
        g.w_bstride, g.vec_bstride, g.sm_group, g.sm_valid, g.sm_keep = w_bstride, vec_bstride, sm_group, sm_valid, sm_keep
        # mirror of the C-side fast-path eligibility (bc_gemm)
        fast = K % 64 == 0
        mode = "dense"
        if conv:
            hv, wv = conv.get("Hv", conv["Hin"]), conv.get("Wv", conv["Win"])
            ups = (hv, wv) != (conv["Hin"], conv["Win"])
            fast = fast and conv["Cin"] % 64 == 0 and (not ups or ((hv, wv) == (2 * conv["Hin"], 2 * conv["Win"])
                                                                  and conv.get("stride", 1) == 1))
            mode = "ups" if ups else f"conv{conv.get('stride', 1)}"
        elif A2 is not None:
            fast = fast and C1 % 64 == 0
        if os.environ.get("BC_GEMM_GENERIC"):
            fast = False
        if tile_cfg in (_lib.TILE_HALO, _lib.TILE_WREG):
            # LDS-resident input-halo convolution (conv_halo.hip): split-K counts 64-channel chunks; fill ~one workgroup per CU
            nch = conv["Cin"] // 64
            base = (M // 128) * (N // 160)
            if splitk is None:
                # Workgroups a split pass aims for and the fewest 64-channel chunks per workgroup.  With conv_wreg.hip (tap loop 867
                # instead of 1445 cycles) a workgroup's fixed costs weigh more and a pass that leaves CUs to the other trunk is worth
                # more than one that fills the chip: 192 / 3 instead of round 2's 256 / 2 - the UNet's 32 x 64 convolutions (128 pixel x
                # column tiles) now run unsplit, the 8 x 16 level in 5 instead of 10 splits - is 9.53 -> 9.29 ms per step (same box,
                # two rounds; 128 ... 192 within 0.05 ms of each other, 224: 9.56, 384: 9.79).
                # Round 5, re-swept on the final kernels (two boxes, two rounds each, ms per step): 192: 9.386 / 9.382 and 9.160 / 9.148;
                # 144: 9.354 / 9.339; 128: 9.323 / 9.330 and 9.081 / 9.087; 112: 9.070 / 9.049; 96: 9.474 / 9.493 - the UNet's 16 x 32
                # convolutions in 2 instead of 3 splits, BlobNet's in 4 instead of 6 (fewer slabs for the reducer); batch 2: 15.58 / 15.74
                # vs 15.85 / 15.78; batch 8 and 768^2 batch 4 unchanged (52.8 / 52.7 vs 52.7 / 52.6; 80.5 / 80.4 vs 80.5 / 80.4).
                target = opt("halo_ctas") or (128 if tile_cfg == _lib.TILE_WREG else 256)
                min_cps = opt("halo_min_cps") or (3 if tile_cfg == _lib.TILE_WREG else 2)
                # workgroups from which one pass is taken unsplit: 2/3 of the target.  (From half fill - BC_HALO_FULL=128 - the 64 x 128
                # BlobNet and 32 x 64 UNet convolutions run as one pass instead of two K halves + a reducer: 13 reducer launches and
                # 0.7 GB of slab traffic less, and the step gains 0.4 % (10.59 -> 10.55 ms, round 3, same box) because the other trunk's
                # kernels fill the idle CUs - but this kernel's own average goes from 37.5 to 45.8 us (0.24 -> 0.19 of peak) and the
                # serialised kernel time of a step rises by 0.76 ms: not the default.)
                full = opt("halo_full") or target * 2 // 3
                # conv_wreg.hip: with five chunks or fewer a half-filled pass is also the faster one in isolation (64 x 128 BlobNet: 43.8 us
                # unsplit vs 45.7 us as two K halves + reducer, cold weights); with ten chunks the split wins (32 x 64 UNet: 61.4 vs 54.5)
                short = tile_cfg == _lib.TILE_WREG and base >= target // 2 and nch <= 5
                splitk = 1 if (base >= full or short) else max(1, min(nch // min_cps, -(-target // base)))
            splitk = max(splitk, -(-nch // self.lib.bc_conv_halo_max_chunks()))     # (the workgroup's affine table lives in LDS)
            cps = -(-nch // max(1, splitk))
            cfg, sk, bm, bn = tile_cfg, -(-nch // cps), 128, 160
            fast, mode = True, "halo"
            if a_gn is not None:
                # GroupNorm in front of the convolution: a_gn = dict(x1, C1, x2, C2, B, HW, G, eps, gamma, beta).  The finalize runs
                # inside the convolution's prologue - every workgroup reads the statistics totals (six words per channel) of the groups
                # overlapping its channel span - whenever that span fits the kernel's scratch, else as its own launch.
                G_ = a_gn["G"]
                span = cps * 64 + 2 * (conv["Cin"] // G_)
                t1, t2 = self.gn_sources(a_gn["x1"], a_gn["C1"], a_gn["x2"], a_gn["C2"], a_gn["B"], a_gn["HW"], a_gn["G"])
                fin_max = 2752 if tile_cfg == _lib.TILE_WREG else 712          # (FIN_MAX_CH of conv_wreg.hip / conv_halo.hip)
                if span <= fin_max and not opt("gn_finalize_launch"):
                    g.a_tot1, g.a_tot2 = ptr(t1), ptr(t2)
                    g.a_gamma, g.a_beta, g.a_groups, g.a_eps = ptr(a_gn["gamma"]), ptr(a_gn["beta"]), G_, a_gn["eps"]
                    self.keep.append((t1, t2, a_gn["gamma"], a_gn["beta"]))
                    for t in (t1, t2, a_gn["gamma"], a_gn["beta"]):
                        self.register(t)
                    mode = "halo_gnfin"
                else:
                    a_affine = self.gn_affine(a_gn["x1"], a_gn["C1"], a_gn["x2"], a_gn["C2"], a_gn["B"], a_gn["HW"], G_, a_gn["eps"],
                                              a_gn["gamma"], a_gn["beta"])
                    g.a_affine = ptr(a_affine)
        elif tile_cfg in _lib.GW_TILES:
            # small-M projection with the weights streamed into VGPRs (gemm_wreg.hip); W is the stream packed for this configuration
            assert not conv and a_affine is None and splitk in (None, 1) and rowvec is None
            cfg, sk, bm, bn = tile_cfg, 1, 64, 64 * _lib.GW_TILES[tile_cfg]
            fast, mode = True, "gw" + ("_ln" if ln_colsum is not None else "") + ("_qkv" if C_t is not None else "") + ("_gn" if a_gn is not None else "") + \
                ("_softmax" if sm_group else "") + ("_wimg" if w_bstride else "")
            if a_gn is not None:          # GroupNorm(x) -> projection: finalize in the kernel's prologue from the statistics totals of x
                t1, _ = self.gn_sources(a_gn["x1"], a_gn["C1"], None, 0, a_gn["B"], a_gn["HW"], a_gn["G"])
                g.a_tot1 = ptr(t1)
                g.a_gamma, g.a_beta, g.a_groups, g.a_eps = ptr(a_gn["gamma"]), ptr(a_gn["beta"]), a_gn["G"], a_gn["eps"]
                self.keep.append((t1, a_gn["gamma"], a_gn["beta"]))
                for t in (t1, a_gn["gamma"], a_gn["beta"]):
                    self.register(t)
        elif tile_cfg == _lib.TILE_G256:
            # large-M dense projection on 256 x 256 tiles, 8 waves, 8-phase LDS-DMA pipeline, persistent workgroups (gemm256.hip)
            assert not conv and a_affine is None and splitk in (None, 1)
            cfg, sk, bm, bn = tile_cfg, 1, 256, 256
            fast, mode = True, "g256" + ("_qkv" if C_t is not None else "") + ("_2src" if A2 is not None else "")
        else:
            assert a_affine is None and (not conv or A2 is None), "fused GroupNorm prologue / two-source conv need TILE_HALO / TILE_WREG"
            cfg, sk, bm, bn = self.plan_gemm(M, N, K, fast, mode, tile_cfg, splitk)
        g.splitk = sk
        g.tile_cfg = cfg
        if sk > 1:
            self.reserve_slab(sk * M * N)
        part = None
        if want_gn:
            rpb = rows_per_batch if rows_per_batch > 0 else (conv["Hout"] * conv["Wout"] if conv else M)
            vec = out_mode == _lib.OUT_F16 and n_out % 8 == 0 and g.ldc % 8 == 0 and (R is None or ldr % 8 == 0) and \
                (R2 is None or ldr2 % 8 == 0)
            slab_rows = bm if sk == 1 else 32            # (a workgroup's rows must lie inside one image)
            if vec and (fast or sk > 1) and N % 4 == 0 and rpb % slab_rows == 0 and M % rpb == 0 and \
                    not os.environ.get("BC_GEMM_TILE"):
                # (gn_tot_out: the caller's table - the first half of a [2 B'] table that bc_dup_halves fans out afterwards)
                part = gn_tot_out if gn_tot_out is not None else self.new_tot(M // rpb, n_out)
                assert tuple(part.shape[:2]) == (M // rpb, n_out)
                g.gn_tot = part.data_ptr()
                self.tots[g.C] = part

        idx = self.lib.bc_plan_add_gemm(self.plan, self.seg.id, self.sid, C.byref(g))
        if idx < 0 or idx != len(self.seg.meta):
            _lib.check(1, "bc_plan_add_gemm")
        refs = (A, A2, W, out, bias, R, R2, rowvec, rowvec_idx, colscale, alpha_dev, alpha_idx, part, a_affine, ln_colsum, C_t)
        self.keep.append(refs)
        for t in refs:
            self.register(t)
        variant = ("conv_halo_kernel<" if cfg == _lib.TILE_HALO else "conv_wreg_kernel<" if cfg == _lib.TILE_WREG else
                   "gemm_wreg_kernel<" if cfg in _lib.GW_TILES else "gemm256_kernel<" if cfg == _lib.TILE_G256 else "gemm_fast_kernel<" if fast else "gemm_kernel<") + _lib.TILE_NAMES[cfg] + "," + mode + ">" + \
            ("+splitk_reduce" if sk > 1 else "")
        # the instantiation rocprofv3 reports for this launch (csrc/*.hip launchers)
        FAST = {1: "256, 128, 4, 2, 3", 2: "128, 128, 2, 2, 3", 3: "128, 128, 2, 2, 2", 4: "256, 64, 4, 1, 2", 5: "256, 64, 4, 1, 3",
                6: "128, 64, 2, 2, 3", 7: "64, 64, 2, 2, 4"}
        if cfg in (_lib.TILE_HALO, _lib.TILE_WREG):
            rp = ("conv_wreg_kernel" if cfg == _lib.TILE_WREG else "conv_halo_kernel") + f"<{2 if g.a_tot1 else 1 if g.a_affine else 0}>"
        elif cfg == _lib.TILE_G256:
            rp = "gemm256_kernel"
        elif cfg in _lib.GW_TILES:
            rp = f"gemm_wreg_kernel<{_lib.GW_TILES[cfg]}, {10 if _lib.GW_TILES[cfg] == 5 else 20}, {'true' if sm_group else 'false'}>"
        elif fast:
            ups_ = bool(conv) and (conv.get("Hv", conv["Hin"]), conv.get("Wv", conv["Win"])) != (conv["Hin"], conv["Win"])
            rp = f"gemm_fast_kernel<{FAST[cfg]}, {'true' if conv else 'false'}, {'true' if ups_ else 'false'}>"
        else:
            rp = "gemm_kernel<" + ("256, 64, 4, 1" if bn == 64 else "128, 128, 2, 2") + ">"
        # algorithmic HBM bytes: activation (a convolution reads each input pixel once), weights and the result once
        a_elems = (M // (conv["Hout"] * conv["Wout"])) * conv["Hin"] * conv["Win"] * conv["Cin"] if conv else M * K
        alg_bytes = 2 * (a_elems + N * K + M * n_out) + (2 * M * n_out if R is not None else 0)
        self._push(kind, 2 * M * N * K, variant, (mode, M, N, K, sk), alg_bytes, rp)
        return out

    # ------------------------------------------------------------------ norms
    def _op(self, name, args, kind, **meta):
        self._add_op(_lib.OPS[name], _lib.op_signature(name), args, kind, **meta)

    def gn_sources(self, x1, C1, x2, C2, B, HW, G):
        """GroupNorm statistics totals of (x1 | x2) for a GroupNorm of G groups: the tables the producers' epilogues added to when they
        exist AND can serve it, else a recorded bc_gn_stats pass.  Returns (tot1, tot2 or None).
        A producer's table of a width that is a multiple of 320 holds sums of 10-channel BLOCKS, spread over the block's ten slots (csrc/
        bc_common.h bc_gn_cg): it serves a GroupNorm whose groups are unions of whole blocks - every one of the UNet / BlobNet - and no other."""
        srcs = []
        c2_ = C2 if x2 is not None else 0
        cpg = (C1 + c2_) // G
        for x, c, off in ((x1, C1, 0), (x2, c2_, C1)):
            if x is None:
                srcs.append(None)
                continue
            hit = self.tots.get(x.data_ptr())
            blocks_ok = c % 320 != 0 or id(hit) in self.tots_per_channel or (cpg % 10 == 0 and off % 10 == 0)
            if hit is not None and hit.shape[1] == c and blocks_ok:
                srcs.append(hit)
            else:
                tot = self.new_tot(B, c)
                self.keep.append((x, tot))
                self._op("bc_gn_stats", (x, c, B, HW, tot), "gn_stats", variant="gn_stats_kernel",
                         shape=("gn_stats", B, HW, c), bytes_=B * HW * c * 2)
                self.tots[x.data_ptr()] = tot
                self.tots_per_channel.add(id(tot))            # (a statistics pass writes one slot per channel whatever the width)
                srcs.append(tot)
        return tuple(srcs)

    def groupnorm(self, x1, C1, x2, C2, B, HW, G, eps, gamma, beta, silu, out=None):
        """GroupNorm(+SiLU) of the channel-concat (x1 | x2) as its own pass.  The statistics come from the producing GEMM's epilogue
        when it emitted them (self.tots), otherwise a standalone bc_gn_stats pass is recorded."""
        c2 = C2 if x2 is not None else 0
        Cc = C1 + c2
        if out is None:
            out = self.empty(B, HW, Cc)
        n_before = self.seg.kinds.get("gn_stats", 0)
        t1, t2 = self.gn_sources(x1, C1, x2, c2, B, HW, G)
        own_stats = self.seg.kinds.get("gn_stats", 0) > n_before
        fused = not os.environ.get("BC_GN_UNFUSED") and Cc // G <= 96          # (the fused kernel's LDS staging holds <= 256 channels)
        self.keep.append((x1, x2, t1, t2, gamma, beta, out))
        kind = "groupnorm" + ("" if own_stats else "_fused_stats")
        if fused:
            self._op("bc_gn_apply_fused", (t1, C1, t2, c2, x1, x2, B, HW, G, eps, gamma, beta, 1 if silu else 0, out), kind,
                     variant="gn_apply_fused_kernel", shape=("gn", B, HW, Cc), bytes_=2 * B * HW * Cc * 2)
        else:
            ab = self.empty(B, Cc, 2, dtype=torch.float32)
            self.keep.append(ab)
            self._op("bc_gn_finalize", (t1, C1, t2, c2, B, HW, G, eps, gamma, beta, ab), "gn_finalize",
                     variant="gn_finalize_kernel", shape=("gn_ab", B, HW, Cc))
            self._op("bc_gn_apply", (x1, C1, x2, c2, B, HW, ab, 1 if silu else 0, out), kind, variant="gn_apply_kernel",
                     shape=("gn", B, HW, Cc), bytes_=2 * B * HW * Cc * 2)
        return out

    def gn_affine(self, x1, C1, x2, C2, B, HW, G, eps, gamma, beta):
        """Per-(image, channel) affine (rstd*gamma, beta - mean*rstd*gamma) of GroupNorm over the channel-concat (x1 | x2), from the
        statistics totals: ab [B][C1+C2][2] fp32.  The consumer (row-chain, or a convolution whose channel span is too wide for the
        in-kernel finalize) applies it while it stages its input, so the normalised activation never goes through HBM."""
        c2 = C2 if x2 is not None else 0
        Cc = C1 + c2
        ab = self.empty(B, Cc, 2, dtype=torch.float32)
        t1, t2 = self.gn_sources(x1, C1, x2, c2, B, HW, G)
        self.keep.append((x1, x2, t1, t2, ab, gamma, beta))
        self._op("bc_gn_finalize", (t1, C1, t2, c2, B, HW, G, eps, gamma, beta, ab), "gn_finalize",
                 variant="gn_finalize_kernel", shape=("gn_ab", B, HW, Cc))
        return ab

    def layernorm(self, x, rows, Cc, gamma, beta, eps, out=None, ldx=None, ldy=None):
        if out is None:
            out = self.empty(rows, Cc)
        self.keep.append((x, gamma, beta, out))
        self._op("bc_layernorm", (x, rows, Cc, ldx or Cc, gamma, beta, eps, out, ldy or Cc), "layernorm", variant="layernorm_kernel",
                 shape=("ln", rows, Cc), bytes_=2 * rows * Cc * 2)
        return out

    # ------------------------------------------------------------------ attention
    def attention(self, Q, K, Vt, out, B, heads, d, Nq, Nkv, ldq, ldk, ldvt, ldo, qbs, kbs, vbs, obs, scale,
                  q_off=0, k_off=0, causal=False):
        self.keep.append((Q, K, Vt, out))
        self._op("bc_attention_causal" if causal else "bc_attention",
                 (Q.data_ptr() + q_off * 2, K.data_ptr() + k_off * 2, Vt, out, B, heads, d, Nq, Nkv, ldq, ldk, ldvt, ldo, qbs, kbs, vbs,
                  obs, scale), "attention", flops=4 * B * heads * Nq * Nkv * d, variant=f"attn_fwd_kernel<{d}>",
                 shape=(B, heads, d, Nq, Nkv), bytes_=2 * B * heads * d * (2 * Nq + 2 * Nkv))
        return out

    # ------------------------------------------------------------------ row-chain (fused transformer-block glue, csrc/rowchain.hip)
    def rowchain(self, kind, Cc, M, rows_per_batch, x, wstream, vec, out0, out1=None, out2=None, ldvt=0, affine=None, gn_in=None, res=None, res2=None,
                 r2=None, r2_xmin=0, r2_bmod=1, out_w=0, gn_tot=None, ln_eps=1e-5, alpha=1.0, alpha_dev=None, alpha_idx=None,
                 alpha_bstride=0, part=None, nsplit=1):
        """`part` / `nsplit`: the split form of the block end (CHAIN_OUT_FF writes, CHAIN_OUT_TAIL reads the fp32 partial sums
        [nsplit][M][C] of the feed-forward: include/blobctrl_hip.h)."""
        # (algorithmic work: OUT_FF / OUT_FFP repeat to_out [proj_out, zero-conv] in every slice - that is not counted)
        mult = {_lib.CHAIN_IN: 4, _lib.CHAIN_MID: 2, _lib.CHAIN_OUT: 14 + (1 if out1 is not None else 0), _lib.CHAIN_OUT_FF: 13,
                _lib.CHAIN_OUT_TAIL: 1 + (1 if out1 is not None else 0), _lib.CHAIN_OUT_FFP: 14 + (1 if out1 is not None else 0)}[kind]
        # gn_in = (totals of x, gamma, beta, groups, eps): CHAIN_IN finalizes the GroupNorm in front of proj_in in its own prologue
        g_tot, g_gamma, g_beta, g_groups, g_eps = gn_in if gn_in is not None else (None, None, None, 0, 0.0)
        refs = (x, wstream, vec, out0, out1, out2, affine, res, res2, r2, gn_tot, alpha_dev, alpha_idx, part, g_tot, g_gamma, g_beta)
        self.keep.append(refs)
        for t in refs:
            self.register(t)
        name = {_lib.CHAIN_IN: "in", _lib.CHAIN_MID: "mid", _lib.CHAIN_OUT: "out", _lib.CHAIN_OUT_FF: f"out_ff/{nsplit}", _lib.CHAIN_OUT_TAIL: "out_tail",
                _lib.CHAIN_OUT_FFP: f"out_ffp/{nsplit}"}[kind]
        zk = (_lib.CHAIN_OUT, _lib.CHAIN_OUT_TAIL, _lib.CHAIN_OUT_FFP)
        self._op("bc_rowchain", (kind, Cc, M, rows_per_batch, x, affine, g_tot, g_gamma, g_beta, g_groups, g_eps, res, res2, r2, r2_xmin, r2_bmod, out_w, wstream, vec, out0, out1, out2,
                                 ldvt, gn_tot, ln_eps, alpha, alpha_dev, alpha_idx, alpha_bstride, part, nsplit), "rowchain",
                 flops=2 * M * Cc * Cc * mult,
                 variant=f"rowchain_kernel<{name}{',zero' if kind in zk and out1 is not None else ''}>",
                 shape=("rowchain_" + name, M, Cc, mult * Cc), bytes_=2 * (mult * Cc * Cc + 2 * M * Cc),
                 rocprof=f"rowchain_kernel<{Cc}, {kind}, {'true' if kind in zk and out1 is not None else 'false'}>")
        return out0

    def rowchain_sum(self, Cc, M, rows_per_batch, part, nsplit, out0, gn_tot=None, out1=None):
        """The reduction behind CHAIN_OUT_FFP (include/blobctrl_hip.h: bc_rowchain_sum): out0 [, out1] = the nsplit fp16 partial outputs
        [, zero-conv partials] summed in slice order; the output's GroupNorm statistics are added to gn_tot."""
        refs = (part, out0, gn_tot, out1)
        self.keep.append(refs)
        for t in refs:
            self.register(t)
        nout = 2 if out1 is not None else 1
        self._op("bc_rowchain_sum", (Cc, M, rows_per_batch, part, nsplit, out0, gn_tot, out1), "rowchain_sum", variant="rowchain_sum_kernel",
                 shape=("rowchain_sum", M, Cc, nsplit), bytes_=2 * nout * (nsplit + 1) * M * Cc, rocprof=f"rowchain_sum_kernel<{Cc}>")
        return out0

    def rowchain_kv_stream(self, ck, cvt, B, T, Cc, ldvt):
        """K rows [B * T][Cc] / V^T [B][Cc][ldvt] of a block's projected context -> the per-(image, wave) fragment streams CHAIN_MIDX
        reads (bc_rowchain_pack_kv; recorded where the context is projected: once per edit)."""
        frags = self.lib.bc_rowchain_kv_frags(Cc)
        if frags < 0:
            _lib.check(1, "bc_rowchain_kv_frags")
        out = self.empty(B * (Cc // 80) * frags * 64 * 8)
        self.keep.append((ck, cvt, out))
        self._op("bc_rowchain_pack_kv", (ck, Cc, cvt, ldvt, B, T, Cc, out), "pack_kv", variant="rowchain_pack_kv_kernel",
                 shape=("pack_kv", B, T, Cc), bytes_=2 * (2 * B * T * Cc) + out.numel() * 2, rocprof=f"rowchain_pack_kv_kernel<{Cc}>")
        return out

    def ctx_fold(self, ck, cvt, B, T, Cc, ldvt, heads, scale, wq, bq, wo):
        """The prompt of an edit folded into the cross-attention weights of one block (bc_ctx_fold, include/blobctrl_hip.h): returns
        (wqk [B][stream], qk_colsum [B][128 heads], qk_bias [B][128 heads], vwo [B][stream]); recorded where the context is projected."""
        N = 128 * heads
        s_qk, s_vo = self.lib.bc_gemm_wreg_stream_elems(N, Cc), self.lib.bc_gemm_wreg_stream_elems(Cc, 80 * heads)
        wqk, vwo = self.zeros(B, s_qk), self.zeros(B, s_vo)               # (zeros: the tail the register ring reads past the end)
        cs, qb = self.zeros(B, N, dtype=torch.float32), self.zeros(B, N, dtype=torch.float32)
        refs = (ck, cvt, wq, bq, wo, wqk, cs, qb, vwo)
        self.keep.append(refs)
        for t in refs:
            self.register(t)
        self._op("bc_ctx_fold", (ck, Cc, cvt, ldvt, B, T, Cc, heads, scale, wq, bq, wo, wqk, cs, qb, vwo), "ctx_fold", variant="ctx_fold_kernel",
                 flops=2 * 2 * B * 80 * heads * Cc * (Cc // heads), shape=("ctx_fold", B, T, Cc), bytes_=2 * (2 * Cc * Cc + B * (N + 80 * heads) * Cc),
                 rocprof="ctx_fold_qk_kernel")
        return wqk, cs, qb, vwo

    def rowchain_midx(self, Cc, M, rows_per_batch, x, wstream, vec, kvstream, n_ctx, heads, scale, out0, out1, res, ln_eps=1e-5):
        """CHAIN_MID with the block's cross-attention inside (include/blobctrl_hip.h: bc_rowchain_midx): out0 = h1, out1 = attn2's
        attention output rows."""
        assert heads * (Cc // heads) == Cc and heads == 8, "CHAIN_MIDX is laid out for 8 heads"
        refs = (x, wstream, vec, kvstream, out0, out1, res)
        self.keep.append(refs)
        for t in refs:
            self.register(t)
        B = M // rows_per_batch
        self._op("bc_rowchain_midx", (Cc, M, rows_per_batch, x, res, wstream, vec, kvstream, n_ctx, scale, out0, out1, ln_eps), "rowchain",
                 flops=2 * M * Cc * Cc * 2 + 4 * M * n_ctx * Cc, variant="rowchain_kernel<midx>",
                 shape=("rowchain_midx", M, Cc, 2 * Cc), bytes_=2 * (2 * Cc * Cc + 3 * M * Cc + 2 * B * n_ctx * Cc),
                 rocprof=f"rowchain_kernel<{Cc}, {_lib.CHAIN_MIDX}, false>")
        return out1

    # ------------------------------------------------------------------ glue
    def call(self, name, *args, kind=None, keep=()):
        """Record a generic `bc_<name>(*args, stream)` launch (any entry point listed in _lib.OPS)."""
        self.keep.append(keep)
        for t in keep if isinstance(keep, (tuple, list)) else (keep,):
            self.register(t)
        self._op(name, args, kind or name)
