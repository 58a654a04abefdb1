mkdir -p gpurun_out
timeout -k 5 500 python -m pytest tests/ -x -q -m gpu --timeout 120 2>&1 | tee gpurun_out/t1.log | tail -3
timeout -k 5 120 python bench.py --reads-per-step 64 --steps 2 --warmup 1 --cpu-reads 0 2>&1 | grep -i "reads_per" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   reads/s', round(d['value'],2), 'fill_ms', round(d['roofline']['avg_launch_ms'],1), 'walk', round(d['roofline']['walk_kernel_ms_per_step'],1))"
