STITCH_DEFINES="STITCH_PROFILE" python stitch_amd/build.py --force 2>&1 | grep -i " error" || true
STITCH_PROFILE_DUMP=1 timeout -k 5 120 python bench.py --reads-per-step 64 --steps 1 --warmup 0 --cpu-reads 0 2>&1 | grep "prof. job" | cut -c1-110 | head -40
