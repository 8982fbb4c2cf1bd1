#!/bin/bash
# r03_z: the kernel-routing environment switches in child processes against the default outputs
set -o pipefail
timeout -k 10 600 python -u -m pytest tests/test_gpu_switches.py -x -q --timeout 400 --timeout-method thread 2>&1 | tail -8
