#!/bin/bash
# (GPU) compile and run tools/wreg_probe.hip: the main loop of a weights-to-VGPR halo convolution on synthetic operands.
mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form tools/wreg_probe.hip -o gpurun_out/wreg_probe 2> gpurun_out/wreg_probe.build.log || { cat gpurun_out/wreg_probe.build.log; exit 1; }
timeout 300 ./gpurun_out/wreg_probe | tee gpurun_out/wreg_probe.txt
rm -f gpurun_out/wreg_probe
