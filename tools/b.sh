#!/bin/bash
# dev tool: build the shipped library (and, with an argument, a profiling build: 1 = exp, 2 = stats); non-zero exit on failure
set -o pipefail
make -C /root/repo/elasticdeform_amd/csrc -j6 2>&1 | grep -E "error|Error" -A6 | head -40
[ ${PIPESTATUS[0]} -eq 0 ] || { echo BUILD FAILED; exit 1; }
if [ -n "$1" ]; then
  make -C /root/repo/elasticdeform_amd/csrc EXPERIMENTS=$1 -j6 2>&1 | grep -E "error|Error" -A6 | head -40
  [ ${PIPESTATUS[0]} -eq 0 ] || { echo EXP BUILD FAILED; exit 1; }
fi
echo BUILD OK
