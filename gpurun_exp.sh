timeout -k 5 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
for k in 1 2 3; do
timeout -k 5 120 python bench.py --reads-per-step 64 --steps 2 --warmup 1 --cpu-reads 0 2>&1 | grep -i "reads_per" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   LC reads/s', round(d['value'],2), 'fill_ms', round(d['roofline']['avg_launch_ms'],1))"
done
cp /tmp/fill_local16_pre_lc.hip stitch_amd/csrc/fill_local16.hip 2>/dev/null || cp gpurun_pre_lc.hip stitch_amd/csrc/fill_local16.hip
python stitch_amd/build.py --force 2>&1 | grep -ci " error" || true
for k in 1 2 3; do
timeout -k 5 120 python bench.py --reads-per-step 64 --steps 2 --warmup 1 --cpu-reads 0 2>&1 | grep -i "reads_per" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   base reads/s', round(d['value'],2), 'fill_ms', round(d['roofline']['avg_launch_ms'],1))"
done
