set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03d; mkdir -p $O
cd $R
THRESHOLDS=96,112,128,160,192 timeout -k 10 300 python scripts/dev_xcd_plan_ab.py > $O/amazon_d64.json 2> $O/err.log || { tail -20 $O/err.log; exit 1; }
THRESHOLDS=64,96,128,192 DIM=128 timeout -k 10 300 python scripts/dev_xcd_plan_ab.py > $O/amazon_d128.json 2>> $O/err.log || { tail -20 $O/err.log; exit 1; }
THRESHOLDS=64,96,128,192 DIM=32 timeout -k 10 300 python scripts/dev_xcd_plan_ab.py > $O/amazon_d32.json 2>> $O/err.log || { tail -20 $O/err.log; exit 1; }
THRESHOLDS=64,96,128,192 PRESET=gowalla timeout -k 10 300 python scripts/dev_xcd_plan_ab.py > $O/gowalla_d64.json 2>> $O/err.log || { tail -20 $O/err.log; exit 1; }
THRESHOLDS=64,96,128,192 PRESET=yelp timeout -k 10 300 python scripts/dev_xcd_plan_ab.py > $O/yelp_d64.json 2>> $O/err.log || { tail -20 $O/err.log; exit 1; }
for f in amazon_d64 amazon_d128 amazon_d32 gowalla_d64 yelp_d64; do python -c "
import json,sys
d=json.load(open('$O/$f.json')); print('$f', json.dumps(d['ms']))"; done
