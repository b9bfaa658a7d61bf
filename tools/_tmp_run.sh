cd /root/repo
mkdir -p gpurun_out/r2k
for gd in 2 4 8; do for L in 6 10 20; do
  for rep in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-serial-extra --opt grid_div=$gd --frames-in-flight $L 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gd=$gd L=$L', round(d['value']), round(d['ms_per_step'],4), d['verified'])"
  done
done; done > gpurun_out/r2k/sweep20.txt 2>&1
python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-serial-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('K=200 default', round(d['value']), round(d['ms_per_step'],4))" >> gpurun_out/r2k/sweep20.txt
