#!/bin/bash
O=gpurun_out/r4ye; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_regs.py -x -q -k "one_bit or repeated_launch" 2>&1 | tail -3 | tee $O/t1.txt && \
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-reads 0 2> $O/b1.err | tee $O/b1.json | cut -c1-200 && \
timeout -k 10 300 python tests/config_runs.py --config cfg5 --reads 100 --batch 100 2> $O/c5_100.err | tee $O/c5_100.json | cut -c1-330 && \
timeout -k 10 300 python tests/config_runs.py --config cfg5 --reads 70 --batch 70 2> $O/c5_70.err | tee $O/c5_70.json | cut -c1-330
