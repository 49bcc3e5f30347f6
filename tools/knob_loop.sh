#!/bin/bash
# how often does tests/_sweep_hash.py print something else than its usual line, under which knobs?  (round 5: this is how the race
# between two staged uploads of one array was found -- 16 of 40 runs differed; after the fix every run of every knob prints one line)
cd $GRAFT_REPO_ROOT
O=gpurun_out/knobs; rm -rf $O; mkdir -p $O
N=${N:-12}
run() { # tag env...
  tag=$1; shift
  for i in $(seq 1 $N); do env SWEEP_DEBUG=1 "$@" timeout 100 python3 tests/_sweep_hash.py > $O/${tag}_$i.txt 2>&1; done
  echo "== $tag: distinct outputs: $(md5sum $O/${tag}_*.txt | awk '{print $1}' | sort | uniq -c | awk '{printf "%s x%s  ", substr($2,1,6), $1}')"
}
run default X=1
run onestream PCL_DP_STREAM=0
run destroysync PCL_DESTROY_SYNC=1
run markers PCL_FEWER_MARKERS=0
run zeromain PCL_ZERO_ASYNC=0
echo "all knobs together: $(cat $O/*.txt | grep SWEEPHASH | sort | uniq -c)"
