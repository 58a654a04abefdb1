timeout -k 10 300 python gpurun_exp.py 2>&1 | tail -2 | cut -c1-300
