"""(GPU) Run-to-run bit-exactness of the halo convolutions: every shape NREP times on the same inputs, outputs compared bitwise."""
import math
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
from blobctrl_amd import _lib  # noqa: E402
from blobctrl_amd.launch import Recorder, encode_gn_tot  # noqa: E402
from blobctrl_amd.weights import pack_conv_wreg  # noqa: E402

dev = torch.device("cuda:0")
rec = Recorder(dev)
stream = torch.cuda.current_stream().cuda_stream
NREP = int(os.environ.get("NREP", "30"))
SHAPES = [(2, 64, 128, 320, 0, 320), (2, 64, 128, 640, 320, 320), (1, 64, 128, 320, 0, 320), (2, 32, 64, 640, 0, 640),
          (2, 16, 32, 1280, 0, 1280), (2, 8, 16, 1280, 1280, 1280), (1, 8, 16, 1280, 0, 1280)]
thrash = torch.zeros(64 << 20, dtype=torch.float32, device=dev)
bad = 0
for name, cfg in (("wreg", _lib.TILE_WREG), ("halo", _lib.TILE_HALO)):
    for (B, H, W, C1, C2, Co) in SHAPES:
        Cin, HW, M = C1 + C2, H * W, B * H * W
        torch.manual_seed(1)
        x1 = torch.randn(B, HW, C1, device=dev, dtype=torch.float16)
        x2 = torch.randn(B, HW, C2, device=dev, dtype=torch.float16) if C2 else None
        wt = (torch.randn(Co, 9 * Cin, device=dev) / math.sqrt(9 * Cin)).half()
        wm = pack_conv_wreg(wt) if cfg == _lib.TILE_WREG else wt
        bias = torch.randn(Co, device=dev)
        gamma, beta = torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev)
        conv = dict(Cin=Cin, Hin=H, Win=W, Hout=H, Wout=W, stride=1)
        for t, c in ((x1, C1), (x2, C2)):
            if t is not None:
                ns = HW // 128
                f = t.float().view(B, ns, 128, c)
                rec.tots[t.data_ptr()] = encode_gn_tot(torch.stack([f.sum((1, 2)), (f * f).sum((1, 2))], -1)).to(t.device)
        seg = rec.begin(f"{name}{B}{H}{C1}{C2}")
        kw = dict(A2=x2, C1=C1, lda2=C2) if C2 else {}
        out = rec.gemm(A=x1, lda=C1, W=wm, M=M, N=Co, K=9 * Cin, out=rec.empty(M, Co), bias=bias, conv=conv, rows_per_batch=HW,
                       tile_cfg=cfg, a_act=_lib.ACT_SILU, want_gn=True,
                       a_gn=dict(x1=x1, C1=C1, x2=x2, C2=C2, B=B, HW=HW, G=32, eps=1e-5, gamma=gamma, beta=beta), **kw)
        ref = None
        ndiff = 0
        for r in range(NREP):
            if r % 3 == 1:
                thrash.add_(1.0)                       # different cache state / timing
            seg.run(stream)
            torch.cuda.synchronize()
            o = out.clone()
            if ref is None:
                ref = o
            elif not torch.equal(o, ref):
                ndiff += 1
                d = (o.float() - ref.float()).abs()
                worst = (int(d.argmax()) // Co, int(d.argmax()) % Co, float(d.max()), int((d > 0).sum()))
        print(f"{name} B{B} {H}x{W} {C1}+{C2}->{Co} sk={seg.meta[-1]['shape'][-1]}: {ndiff} of {NREP - 1} repeats differ" +
              (f" (worst at row {worst[0]} col {worst[1]}: {worst[2]:.4g}, {worst[3]} elements)" if ndiff else ""), flush=True)
        bad += ndiff
sys.exit(1 if bad else 0)
