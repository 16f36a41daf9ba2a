#!/bin/bash
O=gpurun_out/r3_run8; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
{ python3 tools/bf16_twin_probe.py 32768x3456x1024; python3 tools/bf16_twin_probe.py 32768x1024x1024; } 2>&1 | grep -v "DLRM\|amdgpu.ids" > $O/twin_probe.txt
cat $O/twin_probe.txt
