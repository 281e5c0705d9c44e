#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
python tools/slice_exp.py 2>&1 | grep -v amdgpu.ids
