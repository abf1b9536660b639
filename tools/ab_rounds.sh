#!/bin/bash
# Round against round on ONE box: the previous round's tree (staged by hand at tools/_r04_tree: `git worktree add /tmp/r4src <commit>`, built, copied;
# git-ignored) against this tree, alternating -- the only comparison that survives the pool's +-5 % box-to-box spread.
#   tools/ab_rounds.sh reps > profiles/r05_ab_rounds.txt
reps=${1:-2}
B="--no-cpu-baseline --no-flip-rate --no-side-runs --no-roofline --other-modes= --steps 200 --warmup 60 --repeats 3"
for rep in $(seq $reps); do
for tree in tools/_r04_tree .; do
  python $tree/bench.py $B 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); c = d.get('configs2', {})
        print('$tree'.ljust(18), 'value', d['value'], d.get('repeats', {}).get('fps'), 'nodes', d['config']['graph_nodes'],
              '| configs2 f16', c.get('value'), 'f16r', c.get('with_fp16_skip_connections', {}).get('value'),
              '| batch 2x4 / 4x3', [o['value'] for o in d.get('other_configs', [])])
"
done
done
