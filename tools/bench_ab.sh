# alternate the default bench line and the driver's flags a few times on one box (run-to-run spread vs. flag effect)
for rep in 1 2 3; do
  python bench.py --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default      ms_per_step %.4f  avg_launch %.4f  frac %.4f' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))"
  python bench.py --no-cpu --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps20/wu5  ms_per_step %.4f  avg_launch %.4f  frac %.4f' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))"
done
