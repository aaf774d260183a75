#!/bin/bash
# registers / scratch of every kernel of one csrc file: tools/resusage.sh kernels_conv_mfma_persist
cd "$(dirname "$0")/../sup3r_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Rpass-analysis=kernel-resource-usage -c $1.hip -o $1.o 2>&1 | grep -E "Function Name|VGPRs:|SGPRs:|ScratchSize|Spill|Occupancy|LDS Size" | paste - - - - - - - - | sed 's/[^ ]*\.hip:[0-9]*:[0-9]*: remark: //g;s/\[-Rpass-analysis=kernel-resource-usage\]//g' | sed 's/  */ /g' | cut -c1-330
