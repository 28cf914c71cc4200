#!/bin/bash
# (GPU) compile and run tools/atomic_probe.hip: cost of a split-K reduction by 64-bit integer atomics inside the launch; workgroup -> XCD map.
mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/atomic_probe.hip -o gpurun_out/atomic_probe 2> gpurun_out/atomic_probe.build.log || { cat gpurun_out/atomic_probe.build.log; exit 1; }
timeout 300 ./gpurun_out/atomic_probe | tee gpurun_out/atomic_probe.txt
rm -f gpurun_out/atomic_probe
