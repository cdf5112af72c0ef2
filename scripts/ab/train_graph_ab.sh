#!/bin/bash
# whole-chunk hipGraph replay of the training step with and without the side streams (fork / join edges), beside the eager step
cd "$(dirname "$0")/../.."
for rep in 1 2; do
  for cfg in "V2V_X=1|" "V2V_X=1|--train-graph" "V2V_WGRAD_STREAM=0 V2V_REPACK_ASYNC=0|--train-graph" "V2V_WGRAD_STREAM=0 V2V_REPACK_ASYNC=0 V2V_FLOWNET_LANES=0|--train-graph"; do
    envs=${cfg%%|*}; flag=${cfg##*|}
    env $envs timeout 900 python bench.py --mode train --steps 12 --warmup 3 --no-train-parity $flag 2>gpurun_out/train_graph_ab.err | python -c "
import sys, json; j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train [$envs $flag] run $rep:', j['value'], 'frames/s', j['ms_per_step'], 'ms/chunk', 'host issue', j['config'].get('host_issue_ms_per_step'), 'loss_G', j['config'].get('loss_G'))"
  done
done
