timeout -k 10 300 python gpurun_exp.py
timeout -k 5 120 python bench.py --reads-per-step 64 --steps 2 --warmup 1 --cpu-reads 0 2>&1 | grep -i "reads_per" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   cfg2 reads/s', round(d['value'],2), 'fill_ms', round(d['roofline']['avg_launch_ms'],1))"
