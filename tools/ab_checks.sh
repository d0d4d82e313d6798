for v in 0 1 0 1; do
  TOPOLOW_SERIAL_CHECKS=$v timeout -k 10 200 python bench.py --steps ${STEPS:-90} --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('serial_checks $v', round(d['value']), round(d['roofline']['avg_launch_us'],1), round(d['roofline']['check_us'],1), d['final_mae'], d['final_mae_iteration'])"
done
