#!/bin/bash
# round 6: the full GPU suite (verbose) on the current build
O=gpurun_out/r06_full
mkdir -p $O
python -m pytest tests -m gpu -q -s > $O/gpu_tests_verbose.log 2>&1
tail -5 $O/gpu_tests_verbose.log
grep -n "FAILED\|ERROR" $O/gpu_tests_verbose.log | head -20
