#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/${OUT:-r04k}
mkdir -p $O
timeout 300 python tools/train_host_profile.py 2>&1 | grep -v amdgpu.ids > $O/train_host_cprofile.txt
head -75 $O/train_host_cprofile.txt
