#!/bin/bash
# (GPU) compile and run tools/gemm256_probe.hip: the 256 x 256 AGPR-accumulator GEMM main loop on random operands.
mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -w tools/gemm256_probe.hip -o gpurun_out/gemm256_probe 2> gpurun_out/gemm256_probe.build.log || { cat gpurun_out/gemm256_probe.build.log; exit 1; }
timeout 300 ./gpurun_out/gemm256_probe | tee gpurun_out/gemm256_probe.txt
rm -f gpurun_out/gemm256_probe
