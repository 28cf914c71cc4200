"""Launch recorder: turns the hot path into a STATIC list of C-ABI kernel launches over pre-allocated device buffers.

Every method allocates its output once (PyTorch is only the device allocator here), builds the ctypes argument block
once, and appends a zero-argument-ish launch closure `fn(stream)` to the current segment.  Running a segment is a flat
loop over closures; because buffers and argument blocks never change, a segment can be captured into a hipGraph
(`Segment.capture`) and replayed with one `hipGraphLaunch`.
"""
import ctypes as C
import json
import os
from typing import Callable, List, Optional

import torch

from . import _lib
from ._lib import BcGemm


_TUNING_PATH = os.environ.get("BC_TUNING_FILE") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_tuning.json")
_TUNING = None


def tuning_table():
    """Measured best (tile_cfg, splitk) per GEMM shape, written by tools/tune_gemm.py on an MI355X (optional)."""
    global _TUNING
    if _TUNING is None:
        _TUNING = {}
        if os.path.exists(_TUNING_PATH) and not os.environ.get("BC_NO_TUNING"):
            with open(_TUNING_PATH) as f:
                _TUNING = json.load(f).get("shapes", {})
    return _TUNING


def ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


class Segment:
    """An ordered list of launches; replayable eagerly or as a hipGraph."""

    def __init__(self, name: str):
        self.name = name
        self.calls: List[Callable] = []
        self.graph = None
        self.flops = 0          # algorithmic 2*MAC of the GEMM / attention launches recorded here
        self.kinds = {}
        self.meta = []          # per call: dict(kind, flops, variant)
        self.sids = []          # per call: 0 = main stream, 1 = side stream

    @staticmethod
    def _streams(stream, side, extra):
        """Stream of every recorded stream id: 0 = `stream`, 1 = `side`, 2.. = `extra`; missing ones fall back to `stream`."""
        return (stream, side if side is not None else stream) + tuple(e if e is not None else stream for e in extra) + (stream,) * 4

    def run(self, stream: int, side: int = None, extra=()):
        """Replay on `stream`; launches recorded for the side streams go to `side` / `extra` (default: same stream, i.e. serial)."""
        if self.graph is not None:
            _lib.check(_lib.load().bc_graph_launch(self.graph, stream), "bc_graph_launch")
            return
        streams = self._streams(stream, side, extra)
        for fn, sid in zip(self.calls, self.sids):
            fn(streams[sid])

    def capture(self, stream: int, side: int = None, extra=()):
        """Capture this segment into a hipGraph on `stream` (a non-default stream).  Side-stream work joins the capture
        through the recorded fork / signal / wait events and becomes parallel branches of the graph."""
        lib = _lib.load()
        streams = self._streams(stream, side, extra)
        _lib.check(lib.bc_graph_begin(stream), "bc_graph_begin")
        try:
            for fn, sid in zip(self.calls, self.sids):
                fn(streams[sid])
        finally:
            g = C.c_void_p()
            rc = lib.bc_graph_end(stream, C.byref(g))
        _lib.check(rc, "bc_graph_end")
        self.graph = g

    def run_timed(self, stream: int):
        """Eager replay with a HIP event pair around every launch (events are recorded on `stream`, the stream the
        kernels run on).  Returns a list of (meta, milliseconds)."""
        lib = _lib.load()
        evs = []
        for fn in self.calls:          # serial replay on ONE stream: isolates every kernel for the timing table
            a, b = C.c_void_p(), C.c_void_p()
            _lib.check(lib.bc_event_create(C.byref(a)), "bc_event_create")
            _lib.check(lib.bc_event_create(C.byref(b)), "bc_event_create")
            _lib.check(lib.bc_event_record(a, stream), "bc_event_record")
            fn(stream)
            _lib.check(lib.bc_event_record(b, stream), "bc_event_record")
            evs.append((a, b))
        out = []
        for (a, b), m in zip(evs, self.meta):
            ms = C.c_float()
            _lib.check(lib.bc_event_elapsed_ms(a, b, C.byref(ms)), "bc_event_elapsed_ms")
            out.append((m, ms.value))
            lib.bc_event_destroy(a)
            lib.bc_event_destroy(b)
        return out

    def release(self):
        if self.graph is not None:
            _lib.load().bc_graph_destroy(self.graph)
            self.graph = None


class Recorder:
    def __init__(self, device: torch.device):
        self.lib = _lib.load()
        self.device = device
        self.seg: Optional[Segment] = None
        self.keep = []                      # keeps ctypes blocks / tensors alive
        self._slab = {}                     # split-K scratch PER STREAM id (the branches run concurrently)
        self._slab_elems = {}
        info = (C.c_int * 4)()
        _lib.check(self.lib.bc_device_info(info), "bc_device_info")
        self.num_cu = info[0]
        self.bytes_allocated = 0
        self.parts = {}                     # data_ptr of a GEMM output -> (per-channel GroupNorm partials, nslab)
        self.sid = 0                        # stream id new launches are recorded for (0 main, 1 side)
        self.events = []

    # ------------------------------------------------------------------ buffers
    def empty(self, *shape, dtype=torch.float16):
        t = torch.empty(*shape, dtype=dtype, device=self.device)
        self.bytes_allocated += t.numel() * t.element_size()
        return t

    def zeros(self, *shape, dtype=torch.float16):
        t = torch.zeros(*shape, dtype=dtype, device=self.device)
        self.bytes_allocated += t.numel() * t.element_size()
        return t

    def begin(self, name: str) -> Segment:
        self.seg = Segment(name)
        return self.seg

    # ------------------------------------------------------------------ streams / events
    def new_event(self):
        e = C.c_void_p()
        _lib.check(self.lib.bc_event_create_sync(C.byref(e)), "bc_event_create_sync")
        self.events.append(e)
        return e

    def signal(self, ev):
        """Record `ev` on the stream currently being recorded for."""
        lib = self.lib

        def fn(stream):
            rc = lib.bc_event_record(ev, stream)
            if rc:
                _lib.check(rc, "bc_event_record")
        self._push(fn, "event_record")

    def wait(self, ev):
        """Make the stream currently being recorded for wait on `ev`."""
        lib = self.lib

        def fn(stream):
            rc = lib.bc_stream_wait_event(stream, ev)
            if rc:
                _lib.check(rc, "bc_stream_wait_event")
        self._push(fn, "event_wait")

    def _push(self, fn, kind, flops=0, variant="", shape=None, bytes_=0):
        self.seg.calls.append(fn)
        self.seg.sids.append(self.sid)
        self.seg.flops += flops
        self.seg.kinds[kind] = self.seg.kinds.get(kind, 0) + 1
        # bytes_ = algorithmic HBM bytes of an HBM-bound launch (each activation read once + written once, fp16)
        self.seg.meta.append(dict(kind=kind, flops=flops, variant=variant, shape=shape, bytes=bytes_))

    def reserve_slab(self, elems: int):
        """Shared split-K scratch of the CURRENT stream: consumed by the reduce kernel that immediately follows on that
        stream, so one buffer per stream is enough."""
        sid = self.sid
        if elems > self._slab_elems.get(sid, 0):
            self._slab[sid] = torch.empty(elems, dtype=torch.float32, device=self.device)
            self._slab_elems[sid] = elems

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
             a_affine=None, a_act=_lib.ACT_NONE, a_gn=None):
        """Record one bc_gemm.  `conv` = dict(Cin, Hin, Win, Hv, Wv, Hout, Wout, stride) for the 3x3 gather mode.
        Pointer offsets are in ELEMENTS of the respective tensor."""
        n_out = N // 2 if act == _lib.ACT_GEGLU else N
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
        g.ldc = ldc if ldc is not None else n_out
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
        if tile_cfg == _lib.TILE_HALO:
            # LDS-resident input-halo convolution (conv_halo.hip): split-K counts 64-channel chunks; fill ~one workgroup per CU
            nch = conv["Cin"] // 64
            base = (M // 128) * (N // 160)
            if splitk is None:
                target = int(os.environ.get("BC_HALO_CTAS", "256"))
                splitk = 1 if base * 3 >= target * 2 else max(1, min(nch // 2, -(-target // base)))
            cps = -(-nch // max(1, splitk))
            cfg, sk, bm, bn = _lib.TILE_HALO, -(-nch // cps), 128, 160
            fast, mode = True, "halo"
            if a_gn is not None:
                # GroupNorm in front of the convolution: a_gn = dict(x1, C1, x2, C2, B, HW, G, eps, gamma, beta).  The finalize runs
                # inside the convolution's prologue when the workgroup's channel span fits its scratch, else as its own launch.
                G_ = a_gn["G"]
                span = cps * 64 + 2 * (conv["Cin"] // G_)
                (pa1, ns1), (pa2, ns2) = self.gn_sources(a_gn["x1"], a_gn["C1"], a_gn["x2"], a_gn["C2"], a_gn["B"], a_gn["HW"])
                # Every workgroup re-reduces span x nslab partials: measured a LOSS against one bc_gn_finalize launch when that is
                # 170 KB per workgroup (64 x 128 level: 63.9 vs 55.5 us per conv), a gain when it is a few KB (low-resolution levels)
                vol = span * max(ns1, ns2) * 8
                limit = int(os.environ.get("BC_GN_FINALIZE_IN_KERNEL_BYTES", "0"))      # (default off: no gain in the step, see DESIGN)
                if span <= 712 and vol <= limit:
                    g.a_part1, g.a_ns1, g.a_part2, g.a_ns2 = ptr(pa1), ns1, ptr(pa2), ns2
                    g.a_gamma, g.a_beta, g.a_groups, g.a_eps = ptr(a_gn["gamma"]), ptr(a_gn["beta"]), G_, a_gn["eps"]
                    self.keep.append((pa1, pa2, a_gn["gamma"], a_gn["beta"]))
                    mode = "halo_gnfin"
                else:
                    a_affine = self.gn_affine(a_gn["x1"], a_gn["C1"], a_gn["x2"], a_gn["C2"], a_gn["B"], a_gn["HW"], G_, a_gn["eps"],
                                              a_gn["gamma"], a_gn["beta"])
                    g.a_affine = ptr(a_affine)
        else:
            assert a_affine is None and (not conv or A2 is None), "fused GroupNorm prologue / two-source conv need TILE_HALO"
            cfg, sk, bm, bn = self.plan_gemm(M, N, K, fast, mode, tile_cfg, splitk)
        g.splitk = sk
        g.tile_cfg = cfg
        if sk > 1:
            self.reserve_slab(sk * M * N)
        rec = self
        part = None
        if want_gn:
            rpb = rows_per_batch if rows_per_batch > 0 else (conv["Hout"] * conv["Wout"] if conv else M)
            vec = out_mode == _lib.OUT_F16 and n_out % 8 == 0 and g.ldc % 8 == 0 and (R is None or ldr % 8 == 0) and \
                (R2 is None or ldr2 % 8 == 0)
            slab_rows = bm if sk == 1 else 32
            if vec and (fast or sk > 1) and N % 4 == 0 and rpb % slab_rows == 0 and M % rpb == 0 and \
                    not os.environ.get("BC_GEMM_TILE"):
                nslab = rpb // slab_rows
                part = self.empty(M // rpb, nslab, n_out, 2, dtype=torch.float32)
                g.gn_part = part.data_ptr()
                self.parts[g.C] = (part, nslab)

        sid = self.sid

        def fn(stream, g=g, lib=self.lib):
            if g.splitk > 1:
                g.slab = rec._slab[sid].data_ptr()
            rc = lib.bc_gemm(C.byref(g), stream)
            if rc:
                _lib.check(rc, "bc_gemm")

        self.keep.append((g, A, A2, W, out, bias, R, R2, rowvec, colscale, alpha_dev, alpha_idx, part, a_affine))
        variant = ("conv_halo_kernel<" if cfg == _lib.TILE_HALO else "gemm_fast_kernel<" if fast else "gemm_kernel<") + _lib.TILE_NAMES[cfg] + "," + mode + ">" + \
            ("+splitk_reduce" if sk > 1 else "")
        self._push(fn, kind, 2 * M * N * K, variant, (mode, M, N, K, sk))
        return out

    # ------------------------------------------------------------------ norms
    def groupnorm(self, x1, C1, x2, C2, B, HW, G, eps, gamma, beta, silu, out=None):
        """GroupNorm(+SiLU) of the channel-concat (x1 | x2).  Per-channel partial statistics are taken from the producing
        GEMM's epilogue when it emitted them (self.parts), otherwise a standalone bc_gn_stats pass is recorded."""
        lib = self.lib
        c2 = C2 if x2 is not None else 0
        Cc = C1 + c2
        ab = self.empty(B, Cc, 2, dtype=torch.float32)
        if out is None:
            out = self.empty(B, HW, Cc)
        stats_calls = []
        srcs = []
        for x, c in ((x1, C1), (x2, c2)):
            if x is None:
                srcs.append((None, 0))
                continue
            hit = self.parts.get(x.data_ptr())
            if hit is not None and hit[0].shape[2] == c:
                srcs.append((hit[0], hit[1]))
            else:
                nslab = (HW + 127) // 128
                part = self.empty(B, nslab, c, 2, dtype=torch.float32)
                stats_calls.append((x.data_ptr(), c, part.data_ptr(), nslab))
                srcs.append((part, nslab))
        (pa1, ns1), (pa2, ns2) = srcs
        p1, p2, pab, pg, pb, po = ptr(x1), ptr(x2), ptr(ab), ptr(gamma), ptr(beta), ptr(out)
        pp1, pp2 = ptr(pa1), ptr(pa2)
        fused = not os.environ.get("BC_GN_UNFUSED") and Cc // G <= 96          # (the fused kernel's LDS staging holds <= 256 channels)

        def fn(stream):
            rc = 0
            for (px, c, pp, ns) in stats_calls:
                rc = rc or lib.bc_gn_stats(px, c, B, HW, pp, ns, stream)
            if fused:
                rc = rc or lib.bc_gn_apply_fused(pp1, ns1, C1, pp2, ns2, c2, p1, p2, B, HW, G, eps, pg, pb, 1 if silu else 0, po,
                                                 stream)
            else:
                rc = rc or lib.bc_gn_finalize(pp1, ns1, C1, pp2, ns2, c2, B, HW, G, eps, pg, pb, pab, stream)
                rc = rc or lib.bc_gn_apply(p1, C1, p2, c2, B, HW, pab, 1 if silu else 0, po, stream)
            if rc:
                _lib.check(rc, "groupnorm")

        self.keep.append((x1, x2, pa1, pa2, ab, gamma, beta, out))
        nbytes = 2 * B * HW * Cc * 2 + sum(B * HW * c * 2 for (_, c, _, _) in stats_calls)     # stats pass re-reads its source
        self._push(fn, "groupnorm" + ("" if stats_calls else "_fused_stats"), variant="gn_apply_fused_kernel",
                   shape=("gn", B, HW, Cc), bytes_=nbytes)
        return out

    def gn_sources(self, x1, C1, x2, C2, B, HW):
        """Per-channel partial statistics of (x1 | x2): the producers' epilogue partials when they exist, else a recorded
        bc_gn_stats pass.  Returns ((part1, nslab1), (part2, nslab2))."""
        lib = self.lib
        srcs = []
        for x, c in ((x1, C1), (x2, C2 if x2 is not None else 0)):
            if x is None:
                srcs.append((None, 0))
                continue
            hit = self.parts.get(x.data_ptr())
            if hit is not None and hit[0].shape[2] == c:
                srcs.append((hit[0], hit[1]))
            else:
                nslab = (HW + 127) // 128
                part = self.empty(B, nslab, c, 2, dtype=torch.float32)
                px, pp = x.data_ptr(), part.data_ptr()

                def fn(stream, px=px, c=c, pp=pp, nslab=nslab):
                    rc = lib.bc_gn_stats(px, c, B, HW, pp, nslab, stream)
                    if rc:
                        _lib.check(rc, "bc_gn_stats")
                self.keep.append((x, part))
                self._push(fn, "gn_stats", variant="gn_stats_kernel", shape=("gn_stats", B, HW, c), bytes_=B * HW * c * 2)
                self.parts[x.data_ptr()] = (part, nslab)
                srcs.append((part, nslab))
        return tuple(srcs)

    def gn_affine(self, x1, C1, x2, C2, B, HW, G, eps, gamma, beta):
        """Per-(image, channel) affine (rstd*gamma, beta - mean*rstd*gamma) of GroupNorm over the channel-concat (x1 | x2), from the
        producers' per-channel partial statistics: ab [B][C1+C2][2] fp32.  The consumer convolution applies it (and SiLU) while it
        stages its input halo (BcGemm.a_affine), so the normalised activation never goes through HBM."""
        lib = self.lib
        c2 = C2 if x2 is not None else 0
        Cc = C1 + c2
        ab = self.empty(B, Cc, 2, dtype=torch.float32)
        stats_calls, srcs = [], []
        for x, c in ((x1, C1), (x2, c2)):
            if x is None:
                srcs.append((None, 0))
                continue
            hit = self.parts.get(x.data_ptr())
            if hit is not None and hit[0].shape[2] == c:
                srcs.append((hit[0], hit[1]))
            else:
                nslab = (HW + 127) // 128
                part = self.empty(B, nslab, c, 2, dtype=torch.float32)
                stats_calls.append((x.data_ptr(), c, part.data_ptr(), nslab))
                srcs.append((part, nslab))
        (pa1, ns1), (pa2, ns2) = srcs
        pp1, pp2, pab, pg, pb = ptr(pa1), ptr(pa2), ptr(ab), ptr(gamma), ptr(beta)

        def fn(stream):
            rc = 0
            for (px, c, pp, ns) in stats_calls:
                rc = rc or lib.bc_gn_stats(px, c, B, HW, pp, ns, stream)
            rc = rc or lib.bc_gn_finalize(pp1, ns1, C1, pp2, ns2, c2, B, HW, G, eps, pg, pb, pab, stream)
            if rc:
                _lib.check(rc, "gn_affine")

        self.keep.append((x1, x2, pa1, pa2, ab, gamma, beta))
        self._push(fn, "gn_finalize", variant="gn_finalize_kernel", shape=("gn_ab", B, HW, Cc),
                   bytes_=sum(B * HW * c * 2 for (_, c, _, _) in stats_calls))
        return ab

    def layernorm(self, x, rows, Cc, gamma, beta, eps, out=None, ldx=None, ldy=None):
        lib = self.lib
        if out is None:
            out = self.empty(rows, Cc)
        px, pg, pb, po = ptr(x), ptr(gamma), ptr(beta), ptr(out)
        ldx = ldx or Cc
        ldy = ldy or Cc

        def fn(stream):
            rc = lib.bc_layernorm(px, rows, Cc, ldx, pg, pb, eps, po, ldy, stream)
            if rc:
                _lib.check(rc, "bc_layernorm")

        self.keep.append((x, gamma, beta, out))
        self._push(fn, "layernorm", variant="layernorm_kernel", shape=("ln", rows, Cc), bytes_=2 * rows * Cc * 2)
        return out

    # ------------------------------------------------------------------ attention
    def attention(self, Q, K, Vt, out, B, heads, d, Nq, Nkv, ldq, ldk, ldvt, ldo, qbs, kbs, vbs, obs, scale,
                  q_off=0, k_off=0, causal=False):
        lib = self.lib
        entry = lib.bc_attention_causal if causal else lib.bc_attention
        pq = Q.data_ptr() + q_off * 2
        pk = K.data_ptr() + k_off * 2
        pv, po = ptr(Vt), ptr(out)

        def fn(stream):
            rc = entry(pq, pk, pv, po, B, heads, d, Nq, Nkv, ldq, ldk, ldvt, ldo, qbs, kbs, vbs, obs, scale, stream)
            if rc:
                _lib.check(rc, "bc_attention")

        self.keep.append((Q, K, Vt, out))
        self._push(fn, "attention", 4 * B * heads * Nq * Nkv * d, f"attn_fwd_kernel<{d}>", (B, heads, d, Nq, Nkv))
        return out

    # ------------------------------------------------------------------ glue
    def call(self, name, *args, kind=None, keep=()):
        """Record a generic `bc_<name>(*args, stream)` launch with pre-marshalled arguments."""
        f = getattr(self.lib, name)

        def fn(stream):
            rc = f(*args, stream)
            if rc:
                _lib.check(rc, name)

        self.keep.append(keep)
        self._push(fn, kind or name)
