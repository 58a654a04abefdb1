set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r01e
python bench.py > $R/gpurun_out/r01e/bench.json 2> $R/gpurun_out/r01e/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r01e/kt -o kt -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-reads 0 > $R/gpurun_out/r01e/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/r01e/pmc_fetch -o pf -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-reads 0 > $R/gpurun_out/r01e/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/r01e/pmc_write -o pw -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-reads 0 > $R/gpurun_out/r01e/pw.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY -d $R/gpurun_out/r01e/pmc_sq -o ps -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-reads 0 > $R/gpurun_out/r01e/ps.log 2>&1
echo ok
