"""CPU-side tests: C-ABI library loads and exports every declared symbol, host logic (weight packing, LoRA merge,
scheduler tables, plan bookkeeping).  No compute calls are made here (no GPU in this tier)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from blobctrl_amd import _lib, synth, weights
from blobctrl_amd.schedulers import DDIMTable, TableScheduler, UniPCTable
from tests.common import g

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_header_symbol():
    lib = _lib.load()
    header = open(os.path.join(REPO, "include", "blobctrl_hip.h")).read()
    declared = set(re.findall(r"\b(bc_[a-z0-9_]+)\s*\(", header)) - {"bc_half", "bc_stream"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/blobctrl_hip.h but not exported"
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    assert lib.bc_version() >= 100
    assert lib.bc_sizeof_gemm() == C.sizeof(_lib.BcGemm)


def test_product_path_has_no_cpu_fallback():
    from blobctrl_amd.modules import BlobNetModel
    from blobctrl_amd.engine import TrunkConfig
    with pytest.raises(_lib.BlobCtrlHipError):
        BlobNetModel({}, TrunkConfig(in_channels=13, is_blobnet=True), device="cpu")
    from blobctrl_amd.splat import splat_features
    with pytest.raises(_lib.BlobCtrlHipError):
        splat_features(torch.tensor([0.5]), torch.tensor([0.5]), torch.eye(2)[None, None] * 0.01, torch.tensor([[1.0]]),
                       score_size=(8, 8), return_d_score=True, device="cpu")
    # and nothing in the package imports the oracle
    for root, _, files in os.walk(os.path.join(REPO, "blobctrl_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_pack_conv3x3_layout():
    w = g(1, 6, 5, 3, 3)
    p = weights.pack_conv3x3(w)
    assert p.shape == (6, 72)
    for ky in range(3):
        for kx in range(3):
            blk = p[:, (ky * 3 + kx) * 8:(ky * 3 + kx + 1) * 8]
            assert torch.equal(blk[:, :5], w[:, :, ky, kx]) and float(blk[:, 5:].abs().max()) == 0.0


def test_interleave_geglu_roundtrip():
    C_ = 16
    w, b = g(1, 8 * C_, C_), g(2, 8 * C_)
    wi, bi = weights.interleave_geglu(w, b)
    n = 4 * C_
    for j in range(n):
        grp, r = divmod(j, 32)
        assert torch.equal(wi[grp * 64 + r], w[j]) and torch.equal(wi[grp * 64 + 32 + r], w[n + j])
        assert bi[grp * 64 + r] == b[j] and bi[grp * 64 + 32 + r] == b[n + j]


def test_lora_merge_closed_form():
    sd = {"a.to_q.weight": g(1, 16, 8), "c.conv.weight": g(2, 12, 4, 3, 3), "c.conv.bias": g(3, 12)}
    lora = {"a.to_q.lora_A.weight": g(4, 4, 8), "a.to_q.lora_B.weight": g(5, 16, 4),
            "c.conv.lora_A.weight": g(6, 2, 4, 3, 3), "c.conv.lora_B.weight": g(7, 12, 2, 1, 1)}
    m = weights.merge_lora(sd, lora, alphas={"a.to_q": 8.0})
    np.testing.assert_allclose(m["a.to_q.weight"], sd["a.to_q.weight"] + 2.0 * lora["a.to_q.lora_B.weight"] @ lora["a.to_q.lora_A.weight"], rtol=1e-5, atol=1e-6)
    ref = sd["c.conv.weight"] + torch.einsum("or,rikl->oikl", lora["c.conv.lora_B.weight"].flatten(1), lora["c.conv.lora_A.weight"])
    np.testing.assert_allclose(m["c.conv.weight"], ref, rtol=1e-5, atol=1e-5)      # alpha defaults to r => scale 1
    assert torch.equal(m["c.conv.bias"], sd["c.conv.bias"])
    # conv LoRA == running the low-rank branch at runtime (peft semantics)
    x = g(8, 1, 4, 6, 6)
    y_rt = torch.nn.functional.conv2d(x, sd["c.conv.weight"], padding=1) + torch.nn.functional.conv2d(
        torch.nn.functional.conv2d(x, lora["c.conv.lora_A.weight"], padding=1), lora["c.conv.lora_B.weight"])
    np.testing.assert_allclose(torch.nn.functional.conv2d(x, m["c.conv.weight"], padding=1), y_rt, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("kind", ["unipc", "ddim"])
@pytest.mark.parametrize("n", [1, 2, 5, 20, 50])
def test_scheduler_tables_reproduce_reference_trajectories(golden_dir, kind, n):
    s = TableScheduler(kind)
    s.set_timesteps(n)
    assert torch.isfinite(s.table_impl.table()).all()
    if n not in (5, 20, 50):
        return
    z = np.load(os.path.join(golden_dir, "schedulers.npz"))
    assert (s.timesteps.numpy() == z[f"{kind}_{n}_timesteps"]).all()
    traj = z[f"{kind}_{n}_traj"]
    x = torch.from_numpy(traj[0])
    for i in range(n):
        x = s.step(g(100 + i, 1, 4, 8, 8), s.timesteps[i], x)[0]
        assert np.abs(x.numpy() - traj[i + 1]).max() <= 5e-6 * np.abs(traj[i + 1]).max()


def test_synth_weights_are_deterministic_and_schema_sized():
    sh = synth.trunk_param_shapes(5, (320, 640, 1280, 1280), 2, 768, 4, blobnet=False)
    # SD-1.5 UNet (859 520 964 parameters) + the 5th conv_in channel (320 * 9, inf:233-249); BlobNet: SURVEY Appendix B
    assert len(sh) == 686 and sum(int(np.prod(v)) for v in sh.values()) == 859_520_964 + 320 * 9 == 859_523_844
    bs = synth.trunk_param_shapes(1029, (320, 640, 1280, 1280), 2, None, None, blobnet=True)
    assert len(bs) == 626 and sum(int(np.prod(v)) for v in bs.values()) == 845_019_520
    a = synth.synth_tensor("down_blocks.0.resnets.0.conv1.weight", (8, 4, 3, 3), 7)
    b = synth.synth_tensor("down_blocks.0.resnets.0.conv1.weight", (8, 4, 3, 3), 7)
    assert np.array_equal(a, b) and abs(float(a[0, 0, 0, 0]) - float(synth.synth_tensor("x", (8, 4, 3, 3), 7)[0, 0, 0, 0])) > 0


def test_blobnet_keep_window():
    from blobctrl_amd.pipeline import blobnet_keep
    assert blobnet_keep(50, 0.0, 0.9).count(1.0) == 45          # script default (inf:306-307) -> 45 active steps
    assert blobnet_keep(50, 0.0, 1.0) == [1.0] * 50
    assert blobnet_keep(6, 0.0, 0.67) == [1.0, 1.0, 1.0, 1.0, 0.0, 0.0]


def test_library_load_does_not_create_a_second_hip_runtime():
    """Loading libblobctrl_hip.so before torch used to pull in the system libamdhip64 next to torch's bundled copy; the second HIP
    runtime then reports "no ROCm-capable device" (seen in `__graft_entry__.smoke()` on the GPU box).  `_lib.load()` imports torch
    first - one runtime, whatever the import order of the caller."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from blobctrl_amd import _lib\n"
            "_lib.load()\n"
            "import torch\n"
            "libs = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l})\n"
            "print(len(libs), libs)\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().startswith("1 "), out.stdout


def test_no_hot_kernel_uses_scratch(tmp_path):
    """Code-object metadata of the built library: the GEMM / norm / elementwise kernels must not touch scratch (an accumulator
    array demoted to scratch once cost 25x: DESIGN 3.1), the attention kernels at most the few tail spills of the 128-VGPR
    build.  Reads the embedded gfx950 code objects with the ROCm LLVM tools (skipped when they are absent)."""
    import shutil
    import subprocess
    from blobctrl_amd import _lib
    llvm = "/opt/rocm/lib/llvm/bin"
    bundler, readelf = os.path.join(llvm, "clang-offload-bundler"), os.path.join(llvm, "llvm-readelf")
    if not (os.path.exists(bundler) and os.path.exists(readelf) and shutil.which("objcopy")):
        pytest.skip("ROCm LLVM tools not available")
    fat = str(tmp_path / "fat.bin")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", _lib.LIB_PATH, fat])
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [i for i in range(len(blob)) if blob.startswith(magic, i)]
    assert len(starts) >= 5, "expected one offload bundle per translation unit with device code"
    seen = {}
    for n, (a, b) in enumerate(zip(starts, starts[1:] + [len(blob)])):
        part, co = str(tmp_path / f"b{n}.bin"), str(tmp_path / f"b{n}.co")
        open(part, "wb").write(blob[a:b])
        subprocess.check_call([bundler, "--unbundle", "--type=o", f"--input={part}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               f"--output={co}"], stderr=subprocess.DEVNULL)
        notes = subprocess.run([readelf, "--notes", co], capture_output=True, text=True).stdout
        name = None
        for line in notes.splitlines():
            line = line.strip()
            if line.startswith(".name:"):
                name = line.split()[-1]
            elif line.startswith(".private_segment_fixed_size:") and name:
                seen[name] = int(line.split()[-1])
    assert len(seen) >= 40, f"only {len(seen)} kernels found in the code objects"
    for name, scratch in seen.items():
        limit = 128 if "attn_fwd_kernel" in name else 0
        assert scratch <= limit, f"{name}: {scratch} bytes of scratch per lane"


def _vgprs(tok):
    import re
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan_pending_load_hazards(listing):
    """Linear scan of one kernel's disassembly: every global / LDS load's destination registers stay 'pending' until an s_waitcnt that
    retires the request (vmcnt / lgkmcnt are in-order counters: a wait for N leaves the N youngest requests of that kind in flight).  Returns
    the instructions that touch a pending register.  Compiler-made loads never show up here (the compiler waits before a use); hand-written
    `asm volatile` loads - whose waits are hand-counted - are what this guards (csrc/gemm256.hip: lds_read16_raw / ld16_raw)."""
    import re
    vm, lgkm, bad = [], [], []          # in-order queues of (destination registers, text)
    for line in listing:
        t = line.split("//")[0].strip()
        if not t or t.endswith(":") or t.startswith("<"):
            continue
        op = t.split()[0]
        toks = re.findall(r"v\[\d+:\d+\]|v\d+", t)
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", t)
            if m:
                vm = vm[len(vm) - int(m.group(1)):] if int(m.group(1)) < len(vm) else vm
            m = re.search(r"lgkmcnt\((\d+)\)", t)
            if m:
                lgkm = lgkm[len(lgkm) - int(m.group(1)):] if int(m.group(1)) < len(lgkm) else lgkm
            continue
        used = set()
        for k in toks:
            used |= _vgprs(k)
        pend = set().union(*[d for d, _ in vm], *[d for d, _ in lgkm]) if (vm or lgkm) else set()
        if used & pend:
            bad.append(t)
        if op.startswith(("global_load", "buffer_load", "scratch_load")):
            vm.append((set() if "_lds_" in op else _vgprs(toks[0]), t))
        elif op.startswith(("global_store", "buffer_store", "global_atomic", "scratch_store")):
            vm.append((set(), t))
        elif op.startswith("ds_read") or op.startswith("ds_bpermute") or op.startswith("ds_permute") or op.startswith("ds_swizzle"):
            lgkm.append((_vgprs(toks[0]), t))
        elif op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_memtime"):
            lgkm.append((set(), t))
    return bad


def test_pending_load_scan_catches_a_use_before_its_wait():
    ok = ["global_load_dwordx4 v[4:7], v[0:1], off", "global_store_dwordx4 v[2:3], v[8:11], off", "s_waitcnt vmcnt(1)", "v_add_f32 v4, v4, v5"]
    assert scan_pending_load_hazards(ok) == []
    early = ["global_load_dwordx4 v[4:7], v[0:1], off", "global_load_dwordx4 v[12:15], v[0:1], off", "s_waitcnt vmcnt(1)", "v_add_f32 v12, v4, v5"]
    assert scan_pending_load_hazards(early) == ["v_add_f32 v12, v4, v5"]
    lds = ["ds_read_b128 v[4:7], v1", "v_mov_b32 v9, v4", "s_waitcnt lgkmcnt(0)", "v_mov_b32 v9, v4"]
    assert scan_pending_load_hazards(lds) == ["v_mov_b32 v9, v4"]


def test_hand_counted_loads_of_gemm256_are_awaited_before_use(tmp_path):
    """csrc/gemm256.hip issues its epilogue's slab read-back and residual loads from inline assembly with hand-counted waits (the compiler's
    own would be vmcnt(0): DESIGN 3.11).  The compiler does not know those registers are in flight - a register copy or an early use placed
    between request and wait would read garbage.  Disassemble the built kernels and check that nothing touches such a register early."""
    import shutil
    import subprocess
    from blobctrl_amd import _lib
    llvm = "/opt/rocm/lib/llvm/bin"
    bundler, objdump = os.path.join(llvm, "clang-offload-bundler"), os.path.join(llvm, "llvm-objdump")
    if not (os.path.exists(bundler) and os.path.exists(objdump) and shutil.which("objcopy")):
        pytest.skip("ROCm LLVM tools not available")
    fat = str(tmp_path / "fat.bin")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", _lib.LIB_PATH, fat])
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [i for i in range(len(blob)) if blob.startswith(magic, i)]
    kernels = {}
    for n, (a, b) in enumerate(zip(starts, starts[1:] + [len(blob)])):
        part, co = str(tmp_path / f"b{n}.bin"), str(tmp_path / f"b{n}.co")
        open(part, "wb").write(blob[a:b])
        subprocess.check_call([bundler, "--unbundle", "--type=o", f"--input={part}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               f"--output={co}"], stderr=subprocess.DEVNULL)
        dis = subprocess.run([objdump, "-d", co], capture_output=True, text=True).stdout
        if "gemm256_kernel" not in dis:
            continue
        name = None
        for line in dis.splitlines():
            if line.endswith(">:") and "<" in line:
                name = line[line.index("<") + 1:-2]
                kernels[name] = []
            elif name:
                kernels[name].append(line)
    g256 = {k: v for k, v in kernels.items() if "gemm256_kernel" in k}
    assert len(g256) == 2, sorted(kernels)
    # the scan is linear, so it is run on the code object whose hand-counted region is straight-line: gemm256_kernel<false> (unrolled plain /
    # GEGLU / residual forms).  gemm256_kernel<true> keeps compiler-made loads under branches (a linear scan sees the other branch's register
    # writes as early uses); its only assembly loads are the slab reads, awaited in the statement after them
    (name, listing), = [(k, v) for k, v in g256.items() if "ILb0E" in k]
    assert sum("global_load_dwordx4" in l for l in listing) >= 16 and sum("ds_read_b128" in l for l in listing) >= 64, name
    bad = scan_pending_load_hazards(listing)
    assert not bad, f"{name}: {bad[:5]}"


def test_image_processors_match_the_reference_processors(golden_dir):
    """Dinov2ImageProcessor vs transformers' BitImageProcessor with the dinov2 preprocessor_config values, VaeImageProcessor.preprocess
    vs diffusers' (lanczos resize to (height, width), RGB, [-1, 1]) on a non-square image - fixture from tools/make_golden.py."""
    from PIL import Image
    from blobctrl_amd.image_processor import Dinov2ImageProcessor, VaeImageProcessor
    z = np.load(os.path.join(golden_dir, "pipeline_call.npz"))
    odd = Image.fromarray(z["odd"])
    px = Dinov2ImageProcessor().preprocess(images=odd, do_resize=True, return_tensors="pt", do_convert_rgb=True)
    assert px.pixel_values.shape == (1, 3, 224, 224) and dict(**px).keys() == {"pixel_values"}
    assert np.abs(px["pixel_values"].numpy() - z["odd_dino_pixels"]).max() < 2e-6
    v = VaeImageProcessor(vae_scale_factor=8, do_convert_rgb=True)
    assert np.abs(v.preprocess(odd, height=64, width=96).numpy() - z["odd_vae_pixels"]).max() < 1e-6
    x = torch.rand(2, 3, 8, 8) * 2 - 1
    assert v.postprocess(x, "latent") is x
    assert torch.equal(v.postprocess(x, "pt"), (x / 2 + 0.5).clamp(0, 1))
    assert v.postprocess(x, "np").shape == (2, 8, 8, 3) and v.postprocess(x, "pil")[0].size == (8, 8)


def test_scheduler_dropins_from_config_rules():
    """`UniPCMultistepScheduler.from_config(pipeline.scheduler.config)` (inf:276): explicitly set keys are inherited, defaulted
    ones are not (UniPC keeps its own linspace spacing although the source config says 'leading')."""
    from blobctrl_amd.schedulers import DDIMScheduler, UniPCMultistepScheduler
    d = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", steps_offset=1, clip_sample=False,
                      set_alpha_to_one=False)
    u = UniPCMultistepScheduler.from_config(dict(d.config, skip_prk_steps=True))
    u.set_timesteps(50)
    assert u.config.timestep_spacing == "linspace" and u.timesteps[:3].tolist() == [999, 979, 959] and u.order == 1
    back = DDIMScheduler.from_config(u.config)
    back.set_timesteps(50)
    assert back.timesteps[:3].tolist() == [981, 961, 941] and back.init_noise_sigma == 1.0
    import copy
    copy.deepcopy(u)                                     # pipe:1021 deep-copies the scheduler
    with pytest.raises(NotImplementedError):
        UniPCMultistepScheduler(solver_order=3)


def test_plan_runtime_op_signatures_match_the_library():
    """Plan runtime (csrc/plan.hip) without a GPU: every recordable entry point takes exactly the argument count that the ctypes
    signature table derives (a mismatch would scramble arguments at replay); wrong counts, unknown ops and bad ids are refused."""
    import ctypes as C
    lib = _lib.load()
    plan = C.c_void_p()
    assert lib.bc_plan_create(C.byref(plan)) == 0
    seg = lib.bc_plan_segment(plan, b"t")
    assert seg == 0 and lib.bc_plan_find_segment(plan, b"t") == 0 and lib.bc_plan_find_segment(plan, b"nope") == -1
    n = 0
    for name, op in _lib.OPS.items():
        if name == "bc_gemm":
            continue
        sig = _lib.op_signature(name)
        words = (C.c_uint64 * len(sig))()
        assert lib.bc_plan_add_op(plan, seg, 0, op, words, len(sig)) == n, name
        n += 1
        assert lib.bc_plan_add_op(plan, seg, 0, op, words, len(sig) - 1) < 0, name       # wrong count is refused
    ev = lib.bc_plan_new_event(plan)
    one = (C.c_uint64 * 1)(ev)
    assert lib.bc_plan_add_op(plan, seg, 1, _lib.OP_SIGNAL, one, 1) == n and lib.bc_plan_add_op(plan, seg, 0, _lib.OP_WAIT, one, 1) == n + 1
    assert lib.bc_plan_num_launches(plan, seg) == n + 2 and lib.bc_plan_num_launches(plan, 7) == -1
    assert lib.bc_plan_add_op(plan, seg, 0, 99, one, 1) < 0 and b"op 99" in lib.bc_last_error()
    assert lib.bc_plan_add_op(plan, seg, 9, _lib.OP_WAIT, one, 1) < 0                    # stream id out of range
    assert lib.bc_plan_enable(plan, seg, 0, 0) == 0 and lib.bc_plan_enable(plan, seg, 999, 0) != 0
    g = _lib.BcGemm()
    assert lib.bc_plan_add_gemm(plan, seg, 0, C.byref(g)) == n + 2
    assert lib.bc_plan_destroy(plan) == 0


def test_plan_compiles_and_saves_without_a_gpu(tmp_path):
    """The engine in compile-only mode records a whole tiny edit (prologue / active / inactive segments, two stream ids, events)
    on the CPU and writes a relocatable plan file; a launch that points outside the declared buffers is refused."""
    import struct
    from blobctrl_amd.pipeline import BlobCtrlEngine
    from tests.common import TINY, tiny_weights
    from tests.gpu_common import tiny_trunk_configs
    usd, bsd = tiny_weights()
    ucfg, bcfg = tiny_trunk_configs()
    eng = BlobCtrlEngine(usd, bsd, ucfg, bcfg, device="cpu", scheduler="unipc", compile_only=True)
    path = str(tmp_path / "t.bcplan")
    seq = eng.compile_plan(path, 1, 8, 8, 7, TINY["ctx"], 6, blobnet_control_guidance_end=0.67)
    assert seq == ["step_active"] * 4 + ["step_inactive"] * 2
    raw = open(path, "rb").read()
    magic, version, gsz, nbuf = struct.unpack("<IIII", raw[:16])
    assert magic == 0x4E4C5042 and version == 5 and gsz == C.sizeof(_lib.BcGemm) and nbuf > 50
    for name in (b"latents", b"ctx", b"fg_lat", b"bg_lat", b"fg_score", b"bg_score", b"feat16", b"step_idx", b"coef", b"hist"):
        assert name in raw
    P = eng.plan_for(1, 8, 8, 7, TINY["ctx"], 6)
    assert len(P.step_active) > len(P.step_inactive) > 100 and {m["sid"] for m in P.step_active.meta} == {0, 1}
    with pytest.raises(_lib.BlobCtrlHipError):
        BlobCtrlEngine(usd, bsd, ucfg, bcfg, device="cpu")                               # executing needs an MI355X


def test_timeline_tool_on_a_synthetic_trace(tmp_path, capsys):
    """tools/timeline.py (rocprofv3 kernel trace -> per-queue busy time / gaps / overlap of one step) on a hand-made trace."""
    import importlib.util
    import json
    import os
    rows = ["Kind,Agent_Id,Queue_Id,Stream_Id,Thread_Id,Dispatch_Id,Kernel_Id,Kernel_Name,Correlation_Id,Start_Timestamp,End_Timestamp"]
    t = 1000
    k = 0
    for step in range(4):                         # four steps, each: 3 kernels on queue 1, 2 overlapping on queue 2, then cfg_step
        for i in range(3):
            rows.append(f"KERNEL_DISPATCH,1,1,0,1,{k},{k},gemm_kernel(args),{k},{t},{t + 10_000}")
            t += 10_000 + 1_000                    # 1 us gap
            k += 1
        for i in range(2):
            s0 = t - 30_000 + i * 12_000
            rows.append(f"KERNEL_DISPATCH,1,2,0,1,{k},{k},conv_halo_kernel<1>(args),{k},{s0},{s0 + 8_000}")
            k += 1
        rows.append(f"KERNEL_DISPATCH,1,1,0,1,{k},{k},cfg_step_kernel(args),{k},{t},{t + 2_000}")
        t += 5_000
        k += 1
    d = tmp_path / "trace"
    d.mkdir()
    (d / "x_kernel_trace.csv").write_text("\n".join(rows) + "\n")
    spec = importlib.util.spec_from_file_location("timeline", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                           "tools", "timeline.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import sys
    argv = sys.argv
    sys.argv = ["timeline.py", str(d), str(tmp_path / "out.json")]
    try:
        mod.main()
    finally:
        sys.argv = argv
    out = json.loads((tmp_path / "out.json").read_text())
    assert out["kernels_in_step"] == 6 and set(out["queues"]) == {"1", "2"}
    assert out["queues"]["1"]["kernels"] == 4 and abs(out["queues"]["1"]["busy_us"] - 32.0) < 1e-6
    assert out["queues"]["2"]["busy_us"] == 16.0 and out["two_or_more_running_us"] > 0


@pytest.mark.parametrize("C", [320, 640])
def test_rowchain_weight_streams_follow_the_kernels_consumption_order(C):
    """weights.pack_rowchain (host side of csrc/rowchain.hip): per wave ONE linear stream of 1-KiB MFMA fragments in the order the
    kernel consumes them; lane l of fragment (k-step s, tile t) holds W[row0(wave, t) + (l & 15)][32 s + 8 (l >> 4) : + 8]; the
    feed-forward section interleaves, per 128-wide hidden chunk, the (value tile, gate tile) passes of ff.net.0.proj with the K-slice
    of ff.net.2; lengths match bc_rowchain_stream_frags (checked against the library in tests/test_rowchain_gpu.py)."""
    import torch
    from blobctrl_amd import synth, weights
    boc = (C, 2 * C)
    sd = synth.synth_state_dict(synth.trunk_param_shapes(4 + 1 + 3, boc, 1, None, None, blobnet=True), 3)
    pw = weights.PackedTrunk(sd, "cpu", boc, 1)
    p, bp = "down_blocks.0.attentions.0.", "down_blocks.0.attentions.0.transformer_blocks.0."
    nw, ks = C // 80, C // 32
    g = 5 * ks
    w_in, v_in = weights.pack_rowchain(pw, p, 0)
    assert tuple(w_in.shape) == (nw, 4 * g + weights.RC_RPAD, 64, 8) and v_in.numel() == 3 * C
    qk = pw.h[bp + "attn1.to_qk.weight"]
    mats = [pw.h[p + "proj_in.weight"], qk[:C], qk[C:], pw.h[bp + "attn1.to_v.weight"]]
    rng = np.random.Generator(np.random.PCG64(1))
    for _ in range(200):
        seg, wave, s, t, lane = rng.integers(4), rng.integers(nw), rng.integers(ks), rng.integers(5), rng.integers(64)
        frag = w_in[wave, seg * g + s * 5 + t, lane]
        ref = mats[seg][80 * wave + 16 * t + (lane & 15), 32 * s + 8 * (lane >> 4): 32 * s + 8 * (lane >> 4) + 8]
        assert torch.equal(frag, ref)
    assert float(w_in[:, 4 * g:].abs().max()) == 0.0                      # ring padding
    assert torch.equal(v_in[:C], pw.f[p + "proj_in.bias"]) and torch.equal(v_in[C:2 * C], pw.f[bp + "norm1.weight"])
    # OUT chain of a BlobNet block (no cross-attention: attn1.to_out) with its zero-conv
    w_o, v_o = weights.pack_rowchain(pw, p, 2, "blobnet_down_blocks.1")
    hpw, tp, nch = 128 // nw, (128 // nw) // 16, 4 * C // 128
    per_chunk = tp * 2 * ks + 5 * 4
    assert per_chunk == 60 and w_o.shape[1] == 2 * g + nch * per_chunk + g + weights.RC_RPAD
    sdw1, sdw2 = sd[bp + "ff.net.0.proj.weight"].half(), sd[bp + "ff.net.2.weight"].half()
    for _ in range(200):
        c, wave, lane = rng.integers(nch), rng.integers(nw), rng.integers(64)
        base = g + c * per_chunk
        ps, s, vg = rng.integers(tp), rng.integers(ks), rng.integers(2)      # ff.net.0.proj pass `ps`: tile 0 = value rows, tile 1 = gate rows
        j = 128 * c + hpw * wave + 16 * ps + (lane & 15)
        ref = sdw1[(4 * C if vg else 0) + j, 32 * s + 8 * (lane >> 4): 32 * s + 8 * (lane >> 4) + 8]
        assert torch.equal(w_o[wave, base + ps * 2 * ks + s * 2 + vg, lane], ref)
        s2, t = rng.integers(4), rng.integers(5)                             # ff.net.2: K-slice of the chunk
        k0 = 128 * c + 32 * s2 + 8 * (lane >> 4)
        assert torch.equal(w_o[wave, base + tp * 2 * ks + s2 * 5 + t, lane], sdw2[80 * wave + 16 * t + (lane & 15), k0:k0 + 8])
    # fp32 vector: to_out bias, norm3 gamma / beta, GEGLU bias per (chunk, wave, pass: value 16 | gate 16), ff.net.2 bias, proj_out bias, zero-conv bias
    assert v_o.numel() == 3 * C + 8 * C + 3 * C
    b1 = sd[bp + "ff.net.0.proj.bias"]
    c, wave, ps = 3, nw - 1, tp - 1
    off = 3 * C + ((c * nw + wave) * tp + ps) * 32
    j0 = 128 * c + hpw * wave + 16 * ps
    assert torch.equal(v_o[off:off + 16], b1[j0:j0 + 16]) and torch.equal(v_o[off + 16:off + 32], b1[4 * C + j0:4 * C + j0 + 16])
    assert torch.equal(v_o[-C:], sd["blobnet_down_blocks.1.bias"])
    # OUT_FFP (kind 6): per slice z: to_out, the slice's hidden chunks, then proj_out and the zero-conv - the one-launch stream's own
    # segments regrouped per slice; same fp32 vector
    for nsplit in (2, 5 if C == 320 else 4):
        w_p, v_p = weights.pack_rowchain(pw, p, 6, "blobnet_down_blocks.1", nsplit)
        per = nch // nsplit
        assert tuple(w_p.shape) == (nsplit, nw, g + per * per_chunk + 2 * g + weights.RC_RPAD, 64, 8) and torch.equal(v_p, v_o)
        for z in range(nsplit):
            assert torch.equal(w_p[z, :, :g], w_o[:, :g])                                                         # to_out
            assert torch.equal(w_p[z, :, g:g + per * per_chunk], w_o[:, g + z * per * per_chunk: g + (z + 1) * per * per_chunk])
            assert torch.equal(w_p[z, :, g + per * per_chunk: g + per * per_chunk + 2 * g], w_o[:, g + nch * per_chunk: g + nch * per_chunk + 2 * g])
            assert float(w_p[z, :, -weights.RC_RPAD:].abs().max()) == 0.0


@pytest.mark.parametrize("N,Cin", [(160, 64), (320, 192)])
def test_conv_wreg_fragment_streams(N, Cin):
    """weights.pack_conv_wreg (host side of csrc/conv_wreg.hip, = bc_conv_wreg_pack on the device: tests/test_conv_halo_gpu.py): per
    160-column block four column groups of 3 | 2 | 2 | 3 MFMA column tiles x 2 K halves, each ONE contiguous stream
    [chunk][kx][ky][tile][lane][8]; lane l of a fragment holds W[n0 + 16 tile + (l & 15)][(ky * 3 + kx) * Cin + 64 chunk + 32 kg + 8 (l >> 4) : + 8]."""
    from blobctrl_amd.weights import WREG_TILES, pack_conv_wreg
    w = torch.arange(N * 9 * Cin, dtype=torch.float32).reshape(N, 9 * Cin) % 2039      # (exact in fp16)
    w = w.half()
    s = pack_conv_wreg(w).reshape(-1, 64, 8)                                           # [fragment][lane][8]
    nch = Cin // 64
    rng = np.random.Generator(np.random.PCG64(2))
    for _ in range(300):
        blk, grp, kg = rng.integers(N // 160), rng.integers(4), rng.integers(2)
        t0, nt = WREG_TILES[grp]
        chunk, kx, ky, t, lane = rng.integers(nch), rng.integers(3), rng.integers(3), rng.integers(nt), rng.integers(64)
        stream0 = (blk * 20 + 2 * t0 + kg * nt) * nch * 9                              # first fragment of this wave's stream
        frag = stream0 + ((chunk * 3 + kx) * 3 + ky) * nt + t
        n = blk * 160 + (t0 + t) * 16 + (lane & 15)
        k = (ky * 3 + kx) * Cin + chunk * 64 + kg * 32 + 8 * (lane >> 4)
        assert torch.equal(s[frag, lane], w[n, k:k + 8]), (blk, grp, kg, chunk, kx, ky, t, lane)
    assert sorted(s.reshape(-1).tolist()) == sorted(w.reshape(-1).tolist())            # a permutation: nothing lost, nothing doubled


def test_rowchain_split_block_end_streams():
    """weights.pack_rowchain for the two-launch block end (csrc/rowchain.hip BC_CHAIN_OUT_FF / BC_CHAIN_OUT_TAIL): slice z of OUT_FF =
    to_out + the hidden chunks [z, z + 1) * NCH / nsplit of the one-launch stream, OUT_TAIL = its proj_out (+ zero-conv) tail; the fp32
    vector is the one-launch vector."""
    import torch
    from blobctrl_amd import synth, weights
    C, boc = 640, (640, 1280)
    sd = synth.synth_state_dict(synth.trunk_param_shapes(4 + 1 + 3, boc, 1, None, None, blobnet=True), 3)
    pw = weights.PackedTrunk(sd, "cpu", boc, 1)
    p = "down_blocks.0.attentions.0."
    w_one, v_one = weights.pack_rowchain(pw, p, 2, "blobnet_down_blocks.1")
    nw, ks, nch, per_chunk = C // 80, C // 32, 4 * C // 128, 60
    g = 5 * ks
    for nsplit in (2, 4, 5):
        w_ff, v_ff = weights.pack_rowchain(pw, p, 3, "blobnet_down_blocks.1", nsplit)
        per = nch // nsplit
        assert tuple(w_ff.shape) == (nsplit, nw, g + per * per_chunk + weights.RC_RPAD, 64, 8) and torch.equal(v_ff, v_one)
        for z in range(nsplit):
            assert torch.equal(w_ff[z][:, :g], w_one[:, :g])                                            # to_out, repeated in every slice
            assert torch.equal(w_ff[z][:, g:g + per * per_chunk], w_one[:, g + z * per * per_chunk:g + (z + 1) * per * per_chunk])
            assert float(w_ff[z][:, g + per * per_chunk:].abs().max()) == 0.0                           # ring padding
    w_t, v_t = weights.pack_rowchain(pw, p, 4, "blobnet_down_blocks.1")
    assert tuple(w_t.shape) == (nw, 2 * g + weights.RC_RPAD, 64, 8) and torch.equal(v_t, v_one)
    assert torch.equal(w_t[:, :2 * g], w_one[:, g + nch * per_chunk:g + nch * per_chunk + 2 * g])      # proj_out, zero-conv


def test_layernorm_fold_and_gemm_wreg_packing_on_the_host():
    """weights.fold_layernorm states Linear(LayerNorm(x)) = rstd (x W'^T - mean colsum) + bias' exactly (fp64), and pack_gemm_wreg is
    the permutation csrc/gemm_wreg.hip documents: lane l of fragment (column tile j, wave w, k-step s, tile t) holds
    W[64 nt j + 16 nt w + 16 t + (l & 15)][32 s + 8 (l >> 4) : + 8]."""
    from blobctrl_amd.weights import fold_layernorm, pack_gemm_wreg
    torch.manual_seed(0)
    K, N = 64, 48
    x = torch.randn(5, K, dtype=torch.float64) * 2 + 0.7
    w, b = torch.randn(N, K), torch.randn(N)
    gamma, beta = torch.rand(K) + 0.5, torch.randn(K) * 0.2
    w2, cs, b2 = fold_layernorm(w, b, gamma, beta)
    mean, var = x.mean(1, keepdim=True), x.var(1, unbiased=False, keepdim=True)
    rstd = (var + 1e-5).rsqrt()
    got = rstd * (x @ w2.double().t() - mean * cs.double()[None, :]) + b2.double()
    ref = torch.nn.functional.layer_norm(x, (K,), gamma.double(), beta.double(), 1e-5) @ w.double().t() + b.double()
    assert float((got - ref).abs().max()) < 2e-2 * float(ref.abs().max())        # (W' is rounded to fp16: that is the only difference)
    got64 = rstd * (x @ (w.double() * gamma.double()).t() - mean * (w.double() * gamma.double()).sum(1)[None, :]) + b2.double()
    assert float((got64 - ref).abs().max()) < 1e-5
    for nt in (2, 4, 5):
        Nn, Kk = 64 * nt * 2, 96
        W = torch.arange(Nn * Kk, dtype=torch.float32).reshape(Nn, Kk)
        s = pack_gemm_wreg(W, nt)
        assert s.numel() == Nn * Kk + 32 * 512 and float(s[Nn * Kk:].abs().max()) == 0.0
        f = s[:Nn * Kk].reshape(2, 4, Kk // 32, nt, 64, 8)
        for (j, wv, ks, t, l) in ((0, 0, 0, 0, 0), (1, 3, 2, nt - 1, 63), (1, 2, 1, 1, 37)):
            n = 64 * nt * j + 16 * nt * wv + 16 * t + (l & 15)
            k = 32 * ks + 8 * (l >> 4)
            assert torch.equal(f[j, wv, ks, t, l], W[n, k:k + 8])


@pytest.mark.parametrize("C,T", [(320, 77), (640, 17), (320, 80)])
def test_rowchain_kv_stream_layout_holds_every_context_value_once(C, T):
    """weights.pack_rowchain_kv (the host statement of bc_rowchain_pack_kv's layout, compared with the kernel in tests/test_rowchain_gpu.py):
    every K / V^T value of the T tokens sits in exactly one lane slot of its head's fragments, everything else - other heads' channels
    inside a shared k-step, keys >= T, the padding fragments - is zero."""
    import torch
    from blobctrl_amd.weights import pack_rowchain_kv
    B = 2
    k = torch.randint(1, 9, (B, T, C)).half()             # positive integers: sums are exact and no value is zero
    vt = torch.randint(1, 9, (B, C, 96)).half()
    o = pack_rowchain_kv(k, vt, T)
    D = C // 8
    hpw, qks, vtl = 80 // D, (2 if D == 40 else 3), (3 if D == 40 else 5)
    fh = 5 * qks + 3 * vtl
    assert o.shape == (B, C // 80, hpw * fh + 20, 64, 8) and float(o[:, :, hpw * fh:].abs().sum()) == 0.0
    for hh in range(hpw):
        kpart, vpart = o[:, :, hh * fh: hh * fh + 5 * qks], o[:, :, hh * fh + 5 * qks: (hh + 1) * fh]
        for w in range(C // 80):
            ch = slice(80 * w + D * hh, 80 * w + D * (hh + 1))
            assert int((kpart[:, w] != 0).sum()) == B * T * D and float(kpart[:, w].double().sum()) == float(k[:, :, ch].double().sum())
            assert int((vpart[:, w] != 0).sum()) == B * T * D and float(vpart[:, w].double().sum()) == float(vt[:, ch, :T].double().sum())


def test_prompt_folded_into_the_cross_attention_weights_is_the_same_attention():
    """weights.fold_cross_attention (the host statement of bc_ctx_fold): softmax(LN(x) QK^T + bias) VO^T over the padded keys of each head
    IS to_out(attention(to_q(LN(x)), K, V)) (attention.py:491-510, attention_processor.py:2191-2224) - up to the fp16 rounding of QK / VO -
    and padded keys hold zeros, so masking them in the softmax is the whole of the padding's effect."""
    from blobctrl_amd.weights import fold_cross_attention, fold_layernorm
    torch.manual_seed(3)
    B, T, C, heads, rows = 2, 13, 64, 4, 9
    D = C // heads
    x = torch.randn(B, rows, C) * 1.5 + 0.2
    k, v = torch.randn(B, T, C), torch.randn(B, T, C)
    Wq, Wo = torch.randn(C, C) / C ** 0.5, torch.randn(C, C) / C ** 0.5
    gamma, beta = torch.rand(C) + 0.5, torch.randn(C) * 0.2
    scale = D ** -0.5
    wq, _, bq = fold_layernorm(Wq, None, gamma, beta)
    vt = torch.zeros(B, C, 16)
    vt[:, :, :T] = v.transpose(1, 2)
    qk, cs, qb, vo = fold_cross_attention(k, vt, T, heads, scale, wq, bq, Wo)
    assert qk.shape == (B, heads * 128, C) and vo.shape == (B, C, heads * 80) and cs.shape == qb.shape == (B, heads * 128)
    assert float(qk.view(B, heads, 128, C)[:, :, T:].abs().max()) == 0.0 and float(vo.view(B, C, heads, 80)[..., T:].abs().max()) == 0.0
    # the folded form, as the two launches compute it
    mean, var = x.mean(-1, keepdim=True), x.var(-1, unbiased=False, keepdim=True)
    rstd = (var + 1e-5).rsqrt()
    s = rstd * (x @ qk.float().transpose(1, 2) - mean * cs[:, None, :]) + qb[:, None, :]
    s = s.view(B, rows, heads, 128)
    s[..., T:] = float("-inf")
    p = torch.softmax(s, -1)[..., :80].reshape(B, rows, heads * 80)        # (the launch writes the first 80 probabilities of a head)
    got = p @ vo.float().transpose(1, 2)
    # the reference form
    q = (torch.nn.functional.layer_norm(x, (C,), gamma, beta, 1e-5) @ Wq.t()).view(B, rows, heads, D).permute(0, 2, 1, 3)
    kh, vh = (t.view(B, T, heads, D).permute(0, 2, 1, 3) for t in (k, v))
    a = (torch.softmax(q @ kh.transpose(-1, -2) * scale, -1) @ vh).permute(0, 2, 1, 3).reshape(B, rows, C)
    ref = a @ Wo.t()
    assert float((got - ref).abs().max()) < 1e-2 * float(ref.abs().max())


def test_planner_options_parse_and_refuse_typos(monkeypatch):
    """blobctrl_amd/options.py (round 6): ONE variable, BC_PLAN, for everything the planner decides; defaults when unset; a typo or a
    non-integer value raises instead of silently running the default plan; no option exists that aliases weights or ablates a kernel
    (the two diagnostics that produced wrong results are gone from the package)."""
    from blobctrl_amd import options
    monkeypatch.delenv("BC_PLAN", raising=False)
    assert options.non_default() == {} and options.opt("rowchain") is True and options.opt("g256_min_tiles") == 64
    monkeypatch.setenv("BC_PLAN", "rowchain=0, ff_split_640=2;cfg_prefix=0")
    assert options.non_default() == {"rowchain": False, "ff_split_640": 2, "cfg_prefix": False}
    for bad in ("rowchian=0", "rowchain", "gw_maxm=big"):
        monkeypatch.setenv("BC_PLAN", bad)
        with pytest.raises(ValueError):
            options.effective()
    assert not any("alias" in k or "dbg" in k for k in options.TABLE)
    # the package reads no other BC_* planner switch: every os.environ / getenv name in it is on the short list of run-time switches
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "blobctrl_amd")
    names = set()
    for dp, _, fs in os.walk(root):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                names |= set(re.findall(r'(?:environ(?:\.get)?\(|environ\[|getenv\()\s*"(BC_[A-Z0-9_]+)"', src))
    allowed = {"BC_PLAN", "BC_NO_GRAPHS", "BC_LOOP_GRAPH", "BC_ONE_STREAM", "BC_NO_MODULE_GRAPHS", "BC_SPLIT_CFG", "BC_NO_TUNING", "BC_TUNING_FILE",
               "BC_GN_UNFUSED", "BC_GEMM_GENERIC", "BC_GEMM_TILE", "BC_ATTN_NO8", "BC_WREG_STAMPS", "BC_RC_STAMPS",
               "BC_HALO_DBG", "BC_HALO_STAMPS", "BC_G256_STAMPS"}        # (the last three only inside #ifdef BC_DIAGNOSTICS: conv_halo.hip, gemm256.hip)
    assert names <= allowed and len(names) <= 25, sorted(names - allowed)
