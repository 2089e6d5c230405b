#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python tools/r5/s2d.py "$@" > gpurun_out/s2d.log 2>&1
tail -30 gpurun_out/s2d.log
