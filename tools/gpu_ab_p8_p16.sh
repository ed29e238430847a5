#!/bin/bash
# Round-4 opener (prepared at the end of round 3, not yet run): the 8-wave 256 x 256 GEMM with the P16 epilogues of this round on the
# QKV + RoPE (epilogue id 2) and routed SwiGLU (4) launches, against the 4-wave 128 x 128 kernels - same box, one call.
#   VB_BUILD_EXPERIMENTS=1 python -m versband_amd.build        (here, before gpurun: the experiments library travels with the snapshot)
#   gpurun --timeout 600 -- 'export VB_BUILD_EXPERIMENTS=1; bash tools/gpu_ab_p8_p16.sh'
# Read: per-launch us of gemm_bf16_p8_kernel<2,..,2> / <4,..,2> against gemm_bf16_glds_kernel<2,..,true> / <4,..,true> (66.6 / 58.8 us at the
# end of round 3) and the whole-run value.  DESIGN.md 5.0 "open", item 4, says why it might win now.
set -u
export VB_BUILD_EXPERIMENTS=1
M=$(( (1<<2) | (1<<4) ))
timeout 200 python -m pytest tests/test_gpu_path.py -q -x -m gpu -k "eight_wave" 2>&1 | tail -3
bash tools/gpu_ab.sh ab_p8p16 "gemm_bf16_p8|gemm_bf16_glds_kernel<2|gemm_bf16_glds_kernel<4" - "VB_GEMM_P8_MASK=$M VB_GEMM_P8_P16=$M" "VB_GEMM_P8_MASK=$M VB_GEMM_P8_DIRECT=$M" 2>&1 | grep -v "total kernel"
REPS=2 bash tools/gpu_ab_value.sh - "VB_GEMM_P8_MASK=$M VB_GEMM_P8_P16=$M"
