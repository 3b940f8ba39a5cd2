#!/bin/bash
# Round 6, ninth call: the whole GPU suite (timed), then the 12-draw long-horizon robustness study of the final "high" layout on uint8 frames.
set -u
O=gpurun_out/r06j
mkdir -p $O
( time timeout 1500 python -m pytest tests -q -m gpu -x --durations=25 ) > $O/pytest_gpu.txt 2>&1
bash scripts/precision_robustness_long.sh 12 $O/precision_robustness_long.txt 512 "high:u8;high:u8@256;high;high:u8,nodither" > /dev/null 2>&1
tail -45 $O/pytest_gpu.txt
cat $O/precision_robustness_long.txt
