python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tail -5
bash tools/traffic_quick.sh liveorder 2>&1 | tail -4
for w in "" "--mesh mannequin --bins 1024" "--non-confocal"; do
python3 bench.py --steps 10 --warmup 5 --no-cpu-baseline --sustain-seconds 0.5 $w 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$w', 'sustained %.3f ms' % d['sustained_ms_per_step'], {k: round(v,3) for k,v in d['roofline']['kernel_ms'].items()}, (d.get('parity') or {}).get('pass'))"
done
