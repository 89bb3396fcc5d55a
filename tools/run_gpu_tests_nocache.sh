#!/bin/bash
# Allocator shake-out: the GPU tests with torch's caching allocator disabled, so that every tensor is its own hipMalloc and an over-read
# that the caching allocator hides inside a neighbouring block lands somewhere else (how round 4's conv_wide over-read would have shown
# much earlier).  hipGraph capture needs the caching allocator: graph / bench / RCCL / smoke tests are deselected.
#   gpurun -- 'bash tools/run_gpu_tests_nocache.sh'
export PYTORCH_NO_CUDA_MEMORY_CACHING=1 PYTORCH_NO_HIP_MEMORY_CACHING=1
cd "$(dirname "$0")/.." || exit 1
python -m pytest tests -q -m gpu -k "not graph and not bench and not rccl and not smoke" "$@"
