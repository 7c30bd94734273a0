#!/bin/bash
O=gpurun_out/suite
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; tail -15 $O/gpu_tests.log | cut -c1-400
echo done
