for q in 4 16; do for n in 512 1024; do for g in 2 4 8 16; do
  GPU_MAX_HW_QUEUES=$q python bench.py --envs $n --groups $g --steps 8 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('queues $q scenes $n groups $g: %.2f M env-steps/s, %.0f ms/round' % (d['value']/1e6, d['ms_per_step']))"
done; done; done
