#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c7
mkdir -p $O
cd $R
timeout 600 python tools/ab_knob.py '[{"prio_level":0,"prio_light_items":0},{"prio_level":3,"prio_light_items":8},{"prio_level":3,"prio_light_items":24},{"prio_level":3,"prio_light_items":64},{"prio_level":1,"prio_light_items":24},{"prio_level":3,"prio_light_items":200}]' 65536 4 25 > $O/sweep_prio.txt 2>&1
cat $O/sweep_prio.txt
timeout 600 python tools/ab_knob.py '[{"round_packets":256,"takeover_lanes":1},{"round_packets":128,"takeover_lanes":1},{"round_packets":64,"takeover_lanes":1},{"round_packets":256,"takeover_lanes":4},{"round_packets":128,"takeover_lanes":8},{"round_packets":512,"takeover_lanes":1}]' 65536 4 25 > $O/sweep_round.txt 2>&1
cat $O/sweep_round.txt
