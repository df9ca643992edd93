#!/bin/bash
# Builds the probe artefacts of tools/probe (git-ignored; they travel to the GPU box with gpurun).  Run from the repo root after build().
#   libmtl_prof.so     the product library with mtl_mfma.hip compiled -DMTL_X3_PROF (in-kernel s_memtime stall breakdown: conv_prof.py)
#   conv_step_model, ldsdma_rate, ldsdma_contend   stand-alone micro-benchmarks
set -e
H=/opt/rocm/bin/hipcc
C=meta-transfer-learning_amd/csrc
$H --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DMTL_X3_PROF -c $C/mtl_mfma.hip -o /tmp/mtl_mfma_prof.o
$H --offload-arch=gfx950 -shared -fPIC $(ls $C/build/*.o | grep -v "/mtl_mfma.o") /tmp/mtl_mfma_prof.o -o tools/probe/libmtl_prof.so
for p in conv_step_model ldsdma_rate ldsdma_contend; do $H --offload-arch=gfx950 -O3 -Wno-unused-value tools/probe/$p.hip -o tools/probe/$p; done
ls -la tools/probe/libmtl_prof.so tools/probe/conv_step_model tools/probe/ldsdma_rate tools/probe/ldsdma_contend
# per-wave phase breakdown of the bf16-split GEMM engine (tools/probe/gemm_prof.py)
$H --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DMTL_X3G_PROF -c $C/mtl_gemm_x3.hip -o /tmp/mtl_gemm_x3_prof.o
$H --offload-arch=gfx950 -shared -fPIC $(ls $C/build/*.o | grep -v "/mtl_gemm_x3.o") /tmp/mtl_gemm_x3_prof.o -o tools/probe/libmtl_gprof.so
# the 3x3 convolution consumers on 32 x 32 x 16 instructions (the round-3 form) for A/B runs: MTL_LIB=tools/probe/libmtl_m32.so
$H --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DX3H_M16=0 -c $C/mtl_mfma.hip -o /tmp/mtl_mfma_m32.o
$H --offload-arch=gfx950 -shared -fPIC $(ls $C/build/*.o | grep -v "/mtl_mfma.o") /tmp/mtl_mfma_m32.o -o tools/probe/libmtl_m32.so
# conv2's data gradient without the edge-column kernel (the matrix kernel covers all 161 bins) for A/B runs: MTL_LIB=tools/probe/libmtl_noedge.so
$H --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DMTL_DGRAD_EDGE=0 -c $C/mtl_mfma.hip -o /tmp/mtl_mfma_noedge.o
$H --offload-arch=gfx950 -shared -fPIC $(ls $C/build/*.o | grep -v "/mtl_mfma.o") /tmp/mtl_mfma_noedge.o -o tools/probe/libmtl_noedge.so
# the small-product engine with a forced tile / K-group choice (tools/probe/scan_g16.py): MTL_LIB=tools/probe/libmtl_g16probe.so
$H --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DMTL_G16_PROBE -c $C/mtl_gemm16.hip -o /tmp/mtl_gemm16_probe.o
$H --offload-arch=gfx950 -shared -fPIC $(ls $C/build/*.o | grep -v "/mtl_gemm16.o") /tmp/mtl_gemm16_probe.o -o tools/probe/libmtl_g16probe.so
