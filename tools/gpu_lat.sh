#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 300 python tools/latency_one.py 2>&1 | grep -v amdgpu
