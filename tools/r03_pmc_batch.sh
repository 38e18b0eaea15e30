#!/bin/bash
# PMC passes (instruction mix, issue / wait fractions) of single layers: VERDICT r02 next #8 evidence -> gpurun_out/<tag>/pmc_*.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
for lo in "c1s fwd k_conv_pers" "c1s dgrad k_conv_patch" "c1s wgrad k_wgrad2<" "c3s fwd k_conv_patch" "c1t wgrad k_wgrad2<" "c1t fwd k_conv_pers"; do
  set -- $lo
  bash tools/pmc_run.sh $1 $2 r03_$1_$2 || exit 1
  python3 tools/pmc_parse.py r03_$1_$2 "$3" > $O/pmc_$1_$2.txt 2>&1
  rm -rf gpurun_out/pmc_r03_$1_$2
  echo "== $1 $2"; tail -18 $O/pmc_$1_$2.txt
done
