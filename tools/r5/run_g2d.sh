#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python tools/r5/g2d.py "$@" > gpurun_out/g2d.log 2>&1
tail -30 gpurun_out/g2d.log
