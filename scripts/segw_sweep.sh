for cfg in "human 1024 0" "human 1024 64" "human 512 0" "human 512 64" "human 2048 0" "human 2048 64" "quad 1024 0" "quad 1024 64" "quad 2048 0" "quad 2048 64"; do
  set -- $cfg
  python bench.py --robot $1 --bs $2 --segw $3 --no-cpu-baseline --no-boundary --steps 20 --repeats 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$1 bs=$2 segw=$3: %.3e env-steps/s  fwd %.3f ms  bwd %.3f ms  (ms/step %.3f)' % (d['value'], r['fwd_kernel']['avg_launch_ms'], r['avg_launch_ms'], d['ms_per_step']))"
done
