#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "chunk_by_chunk or piece_by_piece" > gpurun_out/r35_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r35_pytest.txt
tail -25 gpurun_out/r35_pytest.txt
