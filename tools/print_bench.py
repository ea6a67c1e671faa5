import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d.get('ms_per_render_fwd_bwd') and {k: d['ms_per_render_fwd_bwd'][k] for k in ('median','graph_replay_median','kernel_sum')})
