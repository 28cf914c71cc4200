"""The plan-level C ABI from a plain C host: tools/make_plan_fixture.py compiles the tiny edit of
tests/golden/loop_tiny.npz into a relocatable `.bcplan` WITHOUT a GPU; tests/c/plan_edit.c (gcc, no Python, no torch) loads it with
bc_plan_load, feeds the inputs, runs the edit eagerly / with per-step graphs / as one whole-loop graph, and checks the final latents
against the reference loop's."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_host_runs_a_tiny_edit_from_a_plan_file(tmp_path):
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")          # the compile step must not need a GPU
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "make_plan_fixture.py"), str(tmp_path)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    exe = str(tmp_path / "plan_edit")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cc = subprocess.run(["gcc", "-O1", "-std=c11", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(REPO, "include"), "-I", f"{rocm}/include",
                         os.path.join(REPO, "tests", "c", "plan_edit.c"), "-o", exe, "-L", os.path.join(REPO, "blobctrl_amd"),
                         "-lblobctrl_hip", "-L", f"{rocm}/lib", "-lamdhip64", "-lm",
                         f"-Wl,-rpath,{os.path.join(REPO, 'blobctrl_amd')}", f"-Wl,-rpath,{rocm}/lib"], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-3000:]
    run = subprocess.run([exe, str(tmp_path / "tiny_edit.bcplan"), str(tmp_path / "tiny_edit_io.bin")], capture_output=True, text=True,
                         timeout=600)
    print(run.stdout)
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-3000:]
    assert run.stdout.count("max-abs err") == 4 and "OK" in run.stdout
