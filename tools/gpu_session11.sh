#!/bin/bash
mkdir -p gpurun_out
timeout 300 python -u tools/diag_shapes.py 2>&1 | tail -40 | cut -c1-200
