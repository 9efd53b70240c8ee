for lib in stagger stagger_lds; do for st in 0 1 2 3; do
  r=$(PENGK_LIB=$PWD/ablation_libs/$lib.so PENGK_EM_STAGGER=$st python bench.py --no-e2e --no-cpu-baseline --config3-steps 0 --k4-patterns 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['components']
print('step %.4f em %.4f stress_serial %.3f' % (d['ms_per_step'], c['em_ms'], c['em_stress_serial_mode_ms']))")
  p=$(PENGK_LIB=$PWD/ablation_libs/$lib.so PENGK_EM_STAGGER=$st python tools/em_probe.py --reps 20 | head -1 | sed 's/.*median \([0-9.]*\) ms.*/\1/')
  w=$(PENGK_LIB=$PWD/ablation_libs/$lib.so PENGK_EM_STAGGER=$st python tools/em_probe.py --reps 5 --W 12 | head -1 | sed 's/.*median \([0-9.]*\) ms.*/\1/')
  echo "$lib stagger=$st: bench $r | probe16 $p ms | W12 $w ms"
done; done
