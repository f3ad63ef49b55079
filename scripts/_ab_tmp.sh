for i in 1 2; do
for O in "wgrad_zs=1" "wgrad_zs=2"; do
  for P in amp split16; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --precision $P --option $O 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$O $P', d['ms_per_step'])"
  done
done; done
