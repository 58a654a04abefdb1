set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r01b
python bench.py > $R/gpurun_out/r01b/bench.json 2> $R/gpurun_out/r01b/bench.err
tail -c 600 $R/gpurun_out/r01b/bench.json
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r01b/kt -o kt -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-reads 0 > $R/gpurun_out/r01b/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/r01b/pmc_fetch -o pf -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-reads 0 > $R/gpurun_out/r01b/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/r01b/pmc_write -o pw -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-reads 0 > $R/gpurun_out/r01b/pw.log 2>&1
find $R/gpurun_out/r01b -type f | head -30
