#!/bin/bash
# the two mixture edge-case tests on 30 other input seeds (GPU box).  Prints one line per seed.
cd "$(dirname "$0")/.."
fail=0
for i in $(seq 1 ${1:-30}); do
  out=$(EBM_TEST_SEED_SHIFT=$((1000 * i)) timeout 300 python -m pytest tests/test_edge_cases_gpu.py -q -p no:cacheprovider \
        -k "native_rng_ragged_dims_langevin or lane_per_chain_mixture_hmc" 2>&1 | tail -1)
  echo "seed shift $((1000 * i)): $out"
  case "$out" in *failed*) fail=$((fail + 1));; esac
done
echo "seeds with a failure: $fail"
