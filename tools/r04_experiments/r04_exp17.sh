#!/bin/bash
O=gpurun_out/r4z; mkdir -p $O
FUZZ_STREAM=1 timeout -k 10 200 python tools/fuzz_case.py 7087 STITCH_NO_JOIN=1 2>&1 | cut -c1-1500 | tee $O/case.txt
