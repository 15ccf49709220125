#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c65
mkdir -p $O
cd $R
timeout 900 python tools/ab_block.py '[{"heavy_predict":480},{"heavy_predict":440},{"heavy_predict":400},{"heavy_predict":520},{"heavy_predict":360}]' 65536 3 1 > $O/ab_hp.txt 2>&1; cat $O/ab_hp.txt
timeout 900 python tools/ab_block.py '[{"prio_level":0},{"prio_level":3,"prio_light_items":64},{"prio_level":3,"prio_light_items":192},{"prio_level":2,"prio_light_items":192},{"prio_level":1,"prio_light_items":400}]' 65536 2 1 > $O/ab_prio.txt 2>&1; cat $O/ab_prio.txt
