#!/bin/bash
# (GPU) compile and run tools/gemm8p_probe.hip: the 8-wave 8-phase 256 x 256 GEMM main loop (csrc/gemm256.hip) stand-alone on random operands.
mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -w tools/gemm8p_probe.hip -o gpurun_out/gemm8p_probe 2> gpurun_out/gemm8p_probe.build.log || { cat gpurun_out/gemm8p_probe.build.log; exit 1; }
timeout 300 ./gpurun_out/gemm8p_probe "$@" | tee gpurun_out/gemm8p_probe.txt
rm -f gpurun_out/gemm8p_probe
