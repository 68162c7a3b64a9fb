#!/bin/bash
# compile the device code only and print the resource usage of the kernels matching $1 (default: all); ISA to /tmp/lpt_kernels.s
cd "$(dirname "$0")/../loupiote_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-math-errno $KDEFS --cuda-device-only -S device.hip -o /tmp/lpt_kernels.s -Rpass-analysis=kernel-resource-usage 2> /tmp/lpt_kernels.res
grep -E "error" -A6 /tmp/lpt_kernels.res | head -40
python3 ../../tools/kres.py /tmp/lpt_kernels.res | grep -E "${1:-.}"
