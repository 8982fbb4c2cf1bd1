#!/bin/bash
# which part of a real product kernel's tile loop takes the fp32 matrix pipe from ~146 to ~120 TFLOP/s? (tools/probe/mfma_feed.hip, built beforehand)
timeout -k 10 120 tools/probe/bin/mfma_feed
