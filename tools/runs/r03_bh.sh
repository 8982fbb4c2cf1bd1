#!/bin/bash
# the vocabulary projection on the weights-in-registers kernel against the LDS-DMA kernel
echo "default dispatch of tst_gemm_mfma_f32"; timeout -k 10 200 python tools/probe/vocab_wreg.py || exit 1
echo "TST_GEMM_WREG=1"; TST_GEMM_WREG=1 timeout -k 10 200 python tools/probe/vocab_wreg.py || exit 1
