# same-box A/B of the dense GEMM kernel's size threshold (HAB_DENSE_MIN_MFLOP; default 4000)
for m in 4000 3000 2000 1000 500 4000 2000; do
  echo "== HAB_DENSE_MIN_MFLOP=$m"
  HAB_DENSE_MIN_MFLOP=$m python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2', d['value'], d['ms_per_step'], d['phases']['update_ms_per_minibatch'])"
done
