for v in 1000000 2000000 4000000 8000000; do python bench.py --width 3840 --height 2160 --steps 6 --warmup 2 --variant $v --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('4K variant $v: %.1f Msamples/s' % d['value'])"; done
