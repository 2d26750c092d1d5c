#!/bin/bash
# Clock evidence for DESIGN.md lesson 15: the shader clock (s_memtime against the constant 100 MHz
# s_memrealtime) inside the product's own matrix-core kernels and inside a bare MFMA loop, same box, back to back.
#   bash tools/clock_evidence.sh > profiles/r06_probe_clock.txt
cd "$(dirname "$0")/.."
export ITERS=${ITERS:-1500}     # the stamped launch is the last of a sustained run of them
echo "== bare MFMA loops (tools/probe/clock_probe.hip): no memory traffic, every CU busy"
[ -x tools/probe/clock_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/probe/clock_probe.hip -o tools/probe/clock_probe
tools/probe/clock_probe
echo
echo "== gemm_f32_dma, fc1 forward 4096 x 720 x 500 (TN_GEMM_DBG stamps; tools/dbg_gemm.py)"
python tools/dbg_gemm.py 4096 720 500 fwd 2>&1 | grep -v Compiling
echo
echo "== gemm_f32_dma, fc1 weight gradient 4096 x 720 x 500"
python tools/dbg_gemm.py 4096 720 500 wgrad 2>&1 | grep -v Compiling
echo
echo "== c8_conv_kernel, wide6 conv2 forward 64->64 @64x64, 128 images (TN_C8_DBG stamps; tools/dbg_c8.py)"
OP=fwd python tools/dbg_c8.py 2>&1 | grep -v Compiling
echo
echo "== c8_wgrad_kernel, wide6 conv2 weight gradient"
OP=wgrad python tools/dbg_c8.py 2>&1 | grep -v Compiling
echo
echo "== c8_wgrad_kernel, wide6 conv6 weight gradient 256->256 @16x16"
OP=wgrad WC=256 WK=256 WH=16 python tools/dbg_c8.py 2>&1 | grep -v Compiling
echo
echo "== rocm-smi while the bare fp32 loop runs"
(tools/probe/clock_probe > /dev/null &) ; sleep 1.5; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | tr -s ' '
wait
