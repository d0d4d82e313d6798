for n in ${SIZES:-20000}; do for v in ${VARIANTS:--1 20}; do
  TOPOLOW_SLAB_VARIANT=$v timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --points $n 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('n $n variant $v', round(d['value'],1), round(d['roofline']['avg_launch_us'],1), round(d['roofline']['frac'],3))"
done; done
