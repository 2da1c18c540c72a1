#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3
