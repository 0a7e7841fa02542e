#!/bin/bash
# GPU box: rebuild libvitsmi.so with rq_spline.hip compiled WITHOUT the former FLAGS_rq_spline and run the tap.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/personalized_text-to-speech_amd/csrc
touch rq_spline.hip
make FLAGS_rq_spline="$1" > /tmp/make.log 2>&1 || { tail -20 /tmp/make.log; exit 1; }
cd $R
NREP=${NREP:-30} timeout -k 10 500 python3 tools/dbg_spline_tap.py
