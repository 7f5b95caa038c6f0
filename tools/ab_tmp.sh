timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for i in 1 2; do timeout 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep -o '"value[^,]*\|kernel_ms[^,]*' | tr '\n' ' '; echo; done
