#!/bin/bash
O=gpurun_out/r4v; mkdir -p $O
timeout -k 10 250 python tools/r04_dbg7.py classic_first 2>/dev/null | tee -a $O/log.txt
timeout -k 10 250 python tools/r04_dbg7.py stream_first 2>/dev/null | tee -a $O/log.txt
STITCH_STREAM_SETTLE_MS=3000 timeout -k 10 250 python tools/r04_dbg7.py settle 2>/dev/null | tee -a $O/log.txt
