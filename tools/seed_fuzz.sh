#!/bin/bash
# Re-runs the bit-exact GPU parity tests on windows drawn from other seeds (SOS_TEST_SEED, see tests/conftest.py):
#   tools/seed_fuzz.sh [first] [last]        (on the GPU box, from the repository root; default seeds 1..8)
# The tests of these files assert bit-identity (or fp64-truth yardsticks measured on the same window), so they must hold on any seed.
A=${1:-1}; B=${2:-8}
FILES="tests/test_gpu_backend.py tests/test_gpu_immature.py tests/test_gpu_pixsel.py tests/test_gpu_undistort.py tests/test_gpu_keyframe_pipeline.py tests/test_gpu_marginalize.py tests/test_gpu_variants.py tests/test_gpu_tracker.py"
for k in $(seq $A $B); do
  r=$(SOS_TEST_SEED=$k python -m pytest $FILES -q -m gpu 2>&1 | grep -E "passed|failed" | tail -1)
  echo "seed $k: $r"
  if echo "$r" | grep -q failed; then SOS_TEST_SEED=$k python -m pytest $FILES -q -m gpu 2>&1 | grep -E "^FAILED|^E  " | head -20; fi
done
