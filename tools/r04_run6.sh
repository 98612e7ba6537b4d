#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04f; mkdir -p $O
timeout 600 python tools/malloc_probe.py > $O/malloc_probe.txt 2>&1; cat $O/malloc_probe.txt
