#!/bin/bash
# Ablation builds of c8_wgrad_tr_kernel (C8W_ABL bits: 1 no products, 2 no operand reads, 4 no LDS-DMA, 8 no edge masks) into
# side libraries theanet_amd/lib/ab/abl_<n>.so (run here: hipcc cross-compiles); on the GPU box:
#   for n in 0 1 2 4 3 6; do TN_HIP_LIB=$PWD/theanet_amd/lib/ab/abl_$n.so OP=wgrad WC=64 WK=64 WH=64 ITERS=300 python tools/dbg_c8.py; done
cd "$(dirname "$0")/../theanet_amd/csrc" && mkdir -p ../lib/ab
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unused-result -Wno-unused-value -Wno-pass-failed"
objs=$(ls build/*.o | grep -v conv_c8.o | tr '\n' ' ')
for n in "$@"; do
  ( /opt/rocm/bin/hipcc $FL -DC8W_ABL=$n -c conv_c8.hip -o /tmp/conv_c8_abl$n.o 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/conv_c8_abl$n.o -o ../lib/ab/abl_$n.so -ldl && echo built abl_$n ) &
done
wait
