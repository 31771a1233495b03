#!/bin/bash
# Developer aid: register / scratch usage of every kernel of one source file (cross-compiles for gfx950, no GPU needed).
#   bash tools/kres.sh wn_respq.hip ["-DFLAG ..."]
cd "$(dirname "$0")/../music_amd/csrc" || exit 1
/opt/rocm/bin/hipcc $2 -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/kres.o 2>&1 |
  grep -E "error|Function Name|VGPRs:|Spill|ScratchSize|TotalSGPRs" |
  sed 's/remark: [a-z_0-9]*\.hip:[0-9]*:[0-9]*: //g; s/\[-Rpass-analysis=kernel-resource-usage\]//; s/[a-z_0-9]*\.hip:[0-9]*:[0-9]*: //; s/remark: *//' |
  paste - - - - - - | sed 's/Function Name: //; s/  */ /g' | cut -c1-220
