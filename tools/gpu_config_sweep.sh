for D in f32x3 bf16 f32; do
for cfgs in "--batch 1" "--batch 3" "--batch 5 --classes 0" "--batch 4 --classes 0" "--batch 8 --height 128 --width 256" "--batch 2 --height 256 --width 512" "--batch 6 --height 32 --width 64" "--batch 33"; do
  out=$(timeout 300 python bench.py --dtype $D $cfgs --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-live-traffic --profile-steps 0 2>&1 | tail -1)
  echo "$D $cfgs -> $(echo "$out" | python -c "import sys,json; l=sys.stdin.read().strip().splitlines()[-1]
try:
    d=json.loads(l); print(d['value'], d.get('losses_finite'))
except Exception as e: print('FAIL', l[:200])")"
done; done
