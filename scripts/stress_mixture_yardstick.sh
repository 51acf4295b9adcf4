#!/bin/bash
# the two mixture edge-case tests on 30 other input seeds (GPU box).  Prints one line per seed (+ the
# statistics of every failed bar).
cd "$(dirname "$0")/.."
fail=0
for i in $(seq 1 ${1:-30}); do
  out=$(EBM_TEST_SEED_SHIFT=$((1000 * i)) timeout 300 python -m pytest tests/test_edge_cases_gpu.py -q -s -p no:cacheprovider \
        -k "native_rng_ragged_dims_langevin or lane_per_chain_mixture_hmc" 2>&1)
  echo "seed shift $((1000 * i)): $(echo "$out" | tail -1)"
  echo "$out" | grep -E "^E +(AssertionError|assert)" | cut -c1-400
  echo "$out" | grep YARDSTICK | python -c "
import sys, ast
rs = [ast.literal_eval(l.split('YARDSTICK', 1)[1].strip()) for l in sys.stdin]
if rs:
    print('   worst ratios over', len(rs), 'cases: median %.2f  q90 %.2f  max %.1f  per-chain %.1f' % (
        max(r['hip_med'] / r['ref_med'] for r in rs), max(r['hip_q90'] / r['ref_q90'] for r in rs),
        max(r['hip_max'] / r['ref_max'] for r in rs), max(r['chain_ratio_max'] for r in rs)))"
  case "$out" in *failed*) fail=$((fail + 1));; esac
done
echo "seeds with a failure: $fail"
