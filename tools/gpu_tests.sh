#!/bin/bash
# gpurun -- 'bash tools/gpu_tests.sh' : GPU parity suite + smoke
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "AssertionError|assert |passed|failed|Error" | head -20
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
