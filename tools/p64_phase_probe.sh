#!/bin/bash
# probe build of the library (phase counters compiled into conv3x3_bf16_p64_kernel) on the GPU box, the probe, then the shipped build back
cd $GRAFT_REPO_ROOT/osvos-pytorch_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DP64_PROF -c conv3x3_bf16_p64.hip -o conv3x3_bf16_p64.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libosvos_hip.so *.o -ldl
(cd ../.. && OSVOS_AUTOBUILD=0 python tools/p64_phase_probe.py 2>&1 | grep -v amdgpu.ids)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -c conv3x3_bf16_p64.hip -o conv3x3_bf16_p64.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libosvos_hip.so *.o -ldl
