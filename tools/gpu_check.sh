#!/bin/bash
# One gpurun call: unit + path parity tests, smoke, short bench.  Logs land in gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch; print('torch', torch.__version__, 'gpu', torch.cuda.is_available(), torch.cuda.get_device_name(0) if torch.cuda.is_available() else '')" 2>&1 | tee gpurun_out/env.log
nproc | tee -a gpurun_out/env.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -rA --tb=short -p no:cacheprovider 2>&1 | tail -150 > gpurun_out/test_kernels.log
echo "kernels exit: $?" >> gpurun_out/test_kernels.log
tail -60 gpurun_out/test_kernels.log
timeout 900 python -m pytest tests/test_gpu_path.py -m gpu -q -rA --tb=short -s -p no:cacheprovider 2>&1 | tail -150 > gpurun_out/test_path.log
tail -60 gpurun_out/test_path.log
if [ "${1:-}" = "bench" ]; then
  timeout 600 python __graft_entry__.py --smoke 2>&1 | tail -5 | tee gpurun_out/smoke.log
  timeout 900 python bench.py --steps 1 --warmup 1 2>&1 | tail -5 | tee gpurun_out/bench.log
fi
