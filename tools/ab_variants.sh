for v in ${VARIANTS:--1 20 21}; do
  TOPOLOW_SLAB_VARIANT=$v timeout -k 10 200 python bench.py --steps ${STEPS:-30} --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant $v', round(d['value']), round(d['roofline']['avg_launch_us'],1), round(d['roofline']['check_us'],1), d['final_mae'])"
done
