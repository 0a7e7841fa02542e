#!/bin/bash
# Runs on the GPU box: A/B of one csrc file's build flags inside the whole step, same box, alternating builds, two rounds.
#   tools/ab_build.sh <file stem> "<flags A>" "<flags B>" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
stem=$1; shift
for round in 1 2; do
for V in "$@"; do
  cd $R/personalized_text-to-speech_amd/csrc && touch $stem.hip && make FLAGS_$stem="$V" > /dev/null
  cd $R && timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary > /tmp/ab.json 2> /tmp/ab.err
  python -c "import json;d=json.load(open('/tmp/ab.json'));print('$stem [$V]:', round(d['ms_per_step'],3), 'ms/step')"
done
done
cd $R/personalized_text-to-speech_amd/csrc && touch $stem.hip && make > /dev/null
