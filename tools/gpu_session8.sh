#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "AssertionError|assert |passed|failed" | head -20
