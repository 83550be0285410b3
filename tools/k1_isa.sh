#!/bin/bash
# dev tool: ISA + resource usage of deform_k1.hip (hipcc -S, no GPU needed).  tools/k1_isa.sh [kernel-substring] [min-block] [-DEXTRA]
cd /root/repo/elasticdeform_amd/csrc || exit 1
K=${1:-k1_fwd_kernelILi3ELb0ELb0ELi2}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. $3 -S --cuda-device-only -Rpass-analysis=kernel-resource-usage -o /tmp/k1.s deform_k1.hip 2>/tmp/k1.log
grep -E "error" -A6 /tmp/k1.log | head -40
grep -E "Function Name|VGPRs:|ScratchSize|VGPRs Spill|SGPRs Spill" /tmp/k1.log | grep -A4 "$K" | head -12
python3 /root/repo/tools/isa_blocks.py /tmp/k1.s "$K" ${2:-60}
