"""Does restricting the BlobNet stream to a subset of the CUs shorten the step?  (GPU, diagnostics.)

The step follows the UNet queue (tools/critical_path.py: 8.6 ms alone, BlobNet always ahead with up to 2.2 ms of slack) and the
1.4 ms above that is BlobNet's workgroups taking CUs the UNet's launches want.  Stream priorities change nothing on this part
(round 3; the switch is gone); this probe gives the BlobNet stream a CU mask (hipExtStreamCreateWithCUMask) and replays eager two-stream edits
(graph nodes do not inherit a stream's mask).  usage: python tools/cumask_probe.py [steps]"""
import ctypes
import os
import sys
import time

os.environ["BC_NO_GRAPHS"] = "1"
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench                                                        # noqa: E402


def hip_runtime():
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            return ctypes.CDLL(line.split()[-1])
    raise SystemExit("no HIP runtime loaded")


def masked_stream(hip, words):
    st = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), len(words), arr)
    if rc != 0:
        raise SystemExit(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(st.value)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    from blobctrl_amd.pipeline import BlobCtrlEngine
    from blobctrl_amd.splat import splat_features
    dev = torch.device("cuda:0")
    torch.cuda.init()
    hip = hip_runtime()
    ucfg, bcfg = bench.full_configs()
    usd, bsd = bench.synth_weights()
    h = w = 64
    inp = bench.synth_inputs(h, w, batch=1)
    score = splat_features(**inp["blob"], score_size=(h, w), return_d_score=True, device=str(dev))
    pipe = BlobCtrlEngine(usd, bsd, ucfg, bcfg, device=str(dev), scheduler="ddim")

    def edit():
        return pipe(inp["prompt"], inp["fg"], inp["bg"], score, inp["dino"], num_inference_steps=steps, latents=inp["latents"])

    def measure(label):
        edit()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            edit()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 2 / steps * 1e3
        print(f"{label:44s} {ms:7.3f} ms / step", flush=True)

    measure("no mask (eager, two streams)")
    full = [0xFFFFFFFF] * 8
    variants = {
        "BlobNet: CUs 0-127": [0xFFFFFFFF] * 4 + [0] * 4,
        "BlobNet: CUs 128-255": [0] * 4 + [0xFFFFFFFF] * 4,
        "BlobNet: even CUs": [0x55555555] * 8,
        "BlobNet: CUs 0-191": [0xFFFFFFFF] * 6 + [0] * 2,
        "BlobNet: 3 of every 4 CUs": [0x77777777] * 8,
        "BlobNet: CUs 0-63": [0xFFFFFFFF] * 2 + [0] * 6,
        "BlobNet: low half of every 32": [0x0000FFFF] * 8,
    }
    for label, words in variants.items():
        pipe.side_stream = masked_stream(hip, words)
        measure(label)
    pipe.side_stream = masked_stream(hip, full)
    measure("BlobNet: all 256 (masked-stream control)")
    # and the other way round: the UNet stream restricted, BlobNet free
    pipe.side_stream = torch.cuda.Stream(device=dev)
    pipe.stream = masked_stream(hip, [0xFFFFFFFF] * 7 + [0])
    measure("UNet: CUs 0-223, BlobNet free")


if __name__ == "__main__":
    main()
