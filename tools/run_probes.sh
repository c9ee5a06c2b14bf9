for l in tests/debug/probe_libs/*.so; do
  echo "== $l"
  DABX_LIBRARY=$PWD/$l python bench.py --steps 20 --warmup 12 --no-cpu-baseline --no-pcie --no-legacy 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['kernel_ms_per_step'], d['fib_crc_bad'], d['payload_mismatch'])"
done
