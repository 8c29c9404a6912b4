for w in "--non-confocal" "--mesh mannequin --bins 1024" "--mesh mannequin --bins 1024 --non-confocal" "--subdivide 1 --grid 32" "--subdivide 2 --grid 32" "--forward-only --grid 32"; do
python3 bench.py --steps 10 --warmup 5 --no-cpu-baseline --sustain-seconds 0.5 $w 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$w', 'sustained %.3f ms' % d['sustained_ms_per_step'], {k: round(v,3) for k,v in d['roofline']['kernel_ms'].items()}, d['config'].get('path',{}).get('backend'), d['config'].get('path',{}).get('grid_R'))"
done
