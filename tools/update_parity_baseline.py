"""Refresh tests/golden/parity_observed_baseline.json from GPU test sessions (their gpurun_out/parity_observed.json files).

    python tools/update_parity_baseline.py --reason "why the numerics changed" run1.json run2.json ... run5.json [--keep-old]

tests/conftest.py fails a session in which an entry left the band tests/parity_gate.py derives from this baseline (max and min
of the session worst case over N sessions), so a refresh is the one way to loosen that gate: it therefore REQUIRES a reason and
at least five sessions (`--allow-few` to override; the gate then uses 3x instead of 2x for those entries), appends the refresh
(date, reason, every entry that got looser, by how much) to tests/golden/parity_baseline_log.json, and every test session prints
that log's tail -- a silent loosening shows up in every run's output.

Per (test, tensor) the baseline keeps the maximum over the sessions of each recorded quantity, `untraced_min` (the smallest
session worst case: max - min is the spread the entry has shown) and `runs`.  Entries of tests that did not run keep their old row;
`--keep-old` also folds the OLD row's max / min into refreshed entries (use when the arithmetic did not change and the new sessions
only add samples)."""
import argparse
import datetime
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import parity_gate  # noqa: E402

DST = parity_gate.BASELINE
LOG = os.path.join(ROOT, 'tests', 'golden', 'parity_baseline_log.json')
GATED = ('untraced_max', 'flipped_pixels')


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('src', nargs='+', help='parity_observed.json of >= 5 GPU sessions of the same code')
    ap.add_argument('--reason', required=True, help='why the observed numbers changed (kept in parity_baseline_log.json)')
    ap.add_argument('--commit', default='', help='commit the observations were taken at')
    ap.add_argument('--allow-few', action='store_true', help='accept fewer than 5 sessions (entries are then gated at 3x)')
    ap.add_argument('--drop-stale', action='store_true', help='drop rows of keys none of the given sessions held (tests that no longer exist)')
    ap.add_argument('--keep-old', action='store_true', help="fold the old rows' max / min / runs into the refreshed entries")
    args = ap.parse_args()
    if len(args.src) < parity_gate.MIN_RUNS and not args.allow_few:
        ap.error(f'{len(args.src)} sessions given; the gate needs {parity_gate.MIN_RUNS} (or --allow-few)')
    old = {(r['test'], r['name']): r for r in parity_gate.load_baseline(DST) if 'untraced_max' in r}
    new = parity_gate.aggregate_sessions([json.load(open(s)) for s in args.src])
    looser, tighter = [], 0
    for k, r in new.items():
        b = old.get(k)
        if b is None:
            continue
        if args.keep_old:
            for f in parity_gate.MAX_FIELDS:
                if f in b:
                    r[f] = max(r.get(f, b[f]), b[f])
            r['untraced_min'] = min(r['untraced_min'], b.get('untraced_min', b['untraced_max']))
            r['runs'] += b.get('runs', 1)
        for f in GATED:
            if f in r and r[f] > b.get(f, 0):
                looser.append(dict(test=k[0], name=k[1], field=f, old=b.get(f, 0), new=r[f]))
            elif f in r and r[f] < b.get(f, 0):
                tighter += 1
    if args.drop_stale:
        stale = [k for k in old if k not in new]
        print(f'{len(stale)} stale rows dropped' + ''.join(f'\n  {k[0]} [{k[1][:50]}]' for k in stale[:5]))
        old = {}
    old.update(new)
    rows = sorted(old.values(), key=lambda r: (r['test'], r['name']))
    json.dump(rows, open(DST, 'w'), indent=0)
    log = json.load(open(LOG)) if os.path.exists(LOG) else []
    log.append(dict(date=datetime.date.today().isoformat(), reason=args.reason, commit=args.commit,
                    source=[os.path.relpath(os.path.abspath(s), ROOT) for s in args.src], sessions=len(args.src),
                    refreshed=len(new), tighter=tighter,
                    looser=sorted(looser, key=lambda e: -e['new'] / max(e['old'], 1e-30))[:40], looser_count=len(looser)))
    json.dump(log, open(LOG, 'w'), indent=1)
    print(f'{len(new)} entries refreshed from {len(args.src)} sessions ({len(looser)} looser, {tighter} tighter), {len(rows)} in {DST}')
    for e in log[-1]['looser'][:10]:
        print(f"  looser: {e['test']} [{e['name']}] {e['field']}: {e['old']:.3g} -> {e['new']:.3g}")
    noisy = [(r['untraced_max'] / max(r['untraced_min'], 1e-30), r) for r in rows if r.get('untraced_max', 0) > 1e-5 and r.get('runs', 1) > 1]
    for q, r in sorted(noisy, key=lambda e: -e[0])[:5]:
        print(f"  spread: {r['test'].split('::')[-1]} [{r['name'][:40]}] {r['untraced_min']:.2e} .. {r['untraced_max']:.2e} over {r['runs']} "
              f"-> limit {parity_gate.limit(r):.2e}")


if __name__ == '__main__':
    main()
