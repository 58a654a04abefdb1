mkdir -p gpurun_out
timeout -k 5 400 python -m pytest tests/ -x -q -m gpu --timeout 120 2>&1 | tee gpurun_out/t1.log | tail -2
timeout -k 10 300 python gpurun_exp.py
