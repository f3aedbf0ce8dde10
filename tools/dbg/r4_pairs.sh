#!/bin/bash
# round 4: chain variants on the shards of an 8-rank run (slot pairs, one-wavefront pixel search), and the table profile
echo "== table profile"; python3 tools/dbg/r4_table_profile.py 2>&1 | grep -v amdgpu | head -45
for v in "IMS_SLOT_PAIRS=0" "IMS_SLOT_PAIRS=1" "IMS_SLOT_PAIRS=1 IMS_PAIR_MAX_OBJECTS=8" "IMS_ROUND_WG=64 IMS_SLOT_PAIRS=1 IMS_PAIR_MAX_OBJECTS=8"; do
  echo "== shard replay: $v"; env $v python3 tools/dbg/shard_times.py 2>&1 | grep "world [48]"
done
