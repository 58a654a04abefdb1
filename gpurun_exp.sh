for d in "" "STITCH_PRIO_YOUNG" "STITCH_PRIO_OLD"; do
  STITCH_DEFINES="$d" python stitch_amd/build.py --force > /dev/null 2>&1
  echo "defs=[$d]"
  python bench.py --reads-per-step 64 --steps 1 --warmup 1 --cpu-reads 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   reads/s', round(d['value'],2), 'fill_ms', round(d['roofline']['avg_launch_ms'],1))"
done
