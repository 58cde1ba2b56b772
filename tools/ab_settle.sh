for s in 1.0 2.5 4.0 1.0 2.5; do
  python tools/bench_fields.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline --min-seconds 2 --settle-seconds $s | cut -c1-200
done
