for rep in 1 2; do
for b in 128 256 400 600; do
  OHGPU_EXP_MIN_BLOCK=$b timeout -k 10 200 python3 bench.py --check --steps 10 --warmup 3 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('min_block',$b, d['roofline']['kernel_avg_ms'], d.get('check'), d['config']['block_kernel_out_frames'])"
done; done
