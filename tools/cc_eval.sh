#!/bin/bash
# compiles eval_kernel.hip alone and prints its register / scratch figures
cd "$(dirname "$0")/../gomokuai_amd/csrc" && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -x hip -c eval_kernel.hip -o eval_kernel.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | grep -E "VGPRs|Scratch|SGPRs|Occupancy|error|warning"
