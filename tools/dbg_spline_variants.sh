#!/bin/bash
# GPU box: the double-launch tap (tools/dbg_spline_tap.py) against several builds of rq_spline.hip; prints, per build, how many of
# the identical back-to-back launch pairs disagreed.  usage: dbg_spline_variants.sh OUTDIR "flags A" "flags B" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
mkdir -p $O
i=0
for f in "$@"; do
  i=$((i+1))
  NREP=${NREP:-10} $R/tools/dbg_spline_tap.sh "$f" > $O/variant_$i.txt 2>&1
  echo "variant $i [$f]: pairs that disagree: $(grep -c 'WITHIN.*gh != gh_again' $O/variant_$i.txt)   $(tail -1 $O/variant_$i.txt)"
done
cd $R/personalized_text-to-speech_amd/csrc && touch rq_spline.hip && make > /dev/null 2>&1
