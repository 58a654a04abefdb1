timeout -k 10 300 python -u -m pytest tests -m gpu -x -q 2>&1 | tail -3
for rps in 42 51 102; do
  python bench.py --reads-per-step $rps --steps 1 --warmup 1 --cpu-reads 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('reads', $rps, 'reads/s', round(d['value'],2), 'fill_ms', round(d['roofline']['avg_launch_ms'],1), 'step_ms', round(d['ms_per_step'],1), 'walk', round(d['roofline']['walk_kernel_ms_per_step'],1))"
done
STITCH_DEFINES="STITCH_PROFILE" python stitch_amd/build.py --force > /dev/null 2>&1
STITCH_PROFILE_DUMP=1 python bench.py --reads-per-step 51 --steps 1 --warmup 0 --cpu-reads 0 2>&1 | grep prof | cut -c1-260 | sed -n '1p;5p;9p'
