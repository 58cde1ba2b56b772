for a in "" "--no-prof" "--warmup 60 --no-prof" "--warmup 60" "" "--no-prof"; do
  python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --min-seconds 1 $a 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$a', '%.1f fps  %.3f ms/step  (sustained %.1f)' % (d['value'], d['ms_per_step'], d.get('sustained', {}).get('value', 0)), d.get('timing_detail'))"
done
