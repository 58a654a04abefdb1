timeout -k 5 120 python bench.py --reads-per-step 64 --steps 3 --warmup 1 --cpu-reads 0 >/dev/null 2>&1
for v in new old new old; do
cp gpurun_fl_$v.hip stitch_amd/csrc/fill_local16.hip
python stitch_amd/build.py --force > gpurun_out/build_$v.log 2>&1 || (grep -B3 -A8 "error" gpurun_out/build_$v.log | head -30)
echo "[$v]"
timeout -k 5 120 python bench.py --reads-per-step 64 --steps 3 --warmup 1 --cpu-reads 0 2>&1 | grep -i "reads_per" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   reads/s', round(d['value'],2), 'fill_ms', round(d['roofline']['avg_launch_ms'],1))"
done
