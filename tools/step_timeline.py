#!/usr/bin/env python3
"""Per-step timeline from a rocprofv3 kernel trace of bench.py: for every launch of the replayed step graph the median duration
and the median idle gap in front of it (end of the previous kernel on the device -> its own start), plus the sum of both against
the step time -- where a step's time is NOT kernel time.

    cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-ms-per-render
    python3 tools/step_timeline.py gpurun_out/trace
"""
import csv
import glob
import re
import statistics
import sys


def short(name):
    m = re.search(r'(\w+_kernel)', name)
    return m.group(1) if m else name[:48]


def main(d):
    f = sorted(glob.glob(d + '/**/*kernel_trace.csv', recursive=True))[-1]
    rows = list(csv.DictReader(open(f)))
    ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])) for r in rows), key=lambda x: x[0])
    # the timed region: the last long run of steps -- find the repeating anchor (the network's forward launch)
    anchors = [i for i, e in enumerate(ev) if e[2] == 'fused_mlp_forward_kernel' or e[2] == 'sp_net_transpose_kernel']
    if len(anchors) < 10:
        print('no repeating step found'); return
    # steps = spans between consecutive anchors with the most common launch count
    spans = [(anchors[i], anchors[i + 1]) for i in range(len(anchors) - 1)]
    n_common = statistics.mode(b - a for a, b in spans)
    steps = [(a, b) for a, b in spans if b - a == n_common]
    # keep the steps of the replayed region: their total time is the smallest
    durs = sorted((ev[b][0] - ev[a][0], a, b) for a, b in steps)
    keep = [x for x in durs if x[0] <= durs[len(durs) // 2][0] * 1.15]
    print(f'{len(keep)} steps of {n_common} launches, median step {statistics.median(x[0] for x in keep) / 1e3:.1f} us')
    tot_k = tot_g = 0.0
    for j in range(n_common):
        ks, gs, name = [], [], None
        for _, a, b in keep:
            s, e, name = ev[a + j]
            prev_end = max(x[1] for x in ev[max(0, a + j - 4):a + j]) if a + j > 0 else s
            ks.append(e - s), gs.append(s - prev_end)
        k, g = statistics.median(ks) / 1e3, statistics.median(gs) / 1e3
        tot_k += k
        tot_g += max(g, 0.0)
        print(f'  {name:36s} kernel {k:7.2f} us   gap in front {g:6.2f} us')
    print(f'sum of kernels {tot_k:.1f} us, sum of positive gaps {tot_g:.1f} us')


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/trace')
