#!/bin/bash
# the fp32 matrix pipe with register operands that stay the same against operands that change with every MFMA (tools/probe/mfma_rate.hip,
# built beforehand into tools/probe/bin/): is ~120 TFLOP/s of the real kernels the pipe on real data?
timeout -k 10 120 tools/probe/bin/mfma_rate
