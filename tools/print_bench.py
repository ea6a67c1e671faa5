#!/usr/bin/env python3
"""One line of a bench.py result: `python tools/print_bench.py result.json` (or the JSON line on stdin when no file is named;
a file argument never touches stdin -- a GPU-box command without a terminal would wait on it for ever)."""
import json
import sys

text = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
d = json.loads(text.strip().splitlines()[-1])
r = d.get('ms_per_render_fwd_bwd')
print(d['value'], d['ms_per_step'], r and {k: r.get(k) for k in ('median', 'graph_replay_median', 'kernel_sum')})
