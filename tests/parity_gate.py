"""The parity drift gate: does a GPU test session's worst case per (test, tensor) still look like the committed baseline?

What is compared.  tests/helpers.FlipCensus records, per comparison against the oracle, `untraced_max` (the largest
max-norm-relative error over the elements NO branch flip explains) and `flipped_pixels`.  Most of these are deterministic to
~1 % run to run.  A few are sums of order-dependent float atomics over ill-conditioned scenes: across the 16 kept GPU sessions
of rounds 4-5 (profiles/r0[45]_*_parity_observed.json) `test_gpu_fuzz[13] dL_dmeans3D` read 4.2e-6 ... 2.7e-5 with unchanged
arithmetic, every other entry above 1.5e-5 stayed within 1.25x of itself.  Round 5's gate compared ONE sample with ONE sample
at 2x and failed the driver's session on exactly that entry (GPUTEST_r05 rc = 1 with 417 / 417 tests passed).

The gate now.  The baseline keeps, per key, the MAX and the MIN over N >= 5 sessions of the session's worst case, and N.
An observation r fails when

    r > max(FACTOR * b_max,  b_max + SPREAD * (b_max - b_min),  FLOOR)          FACTOR = 2 (3 while N < 5), SPREAD = 3

i.e. twice the worst of N sessions, widened by three times the spread the entry has SHOWN (zero for the deterministic ones), and
never below FLOOR = 3e-5 (a third of the 1e-4 tolerance the tests assert themselves).  A real 7e-5 -> 7e-4 jump (round 2's) is
10x over b_max and fails (tests/test_host_cpu.py::test_parity_gate_*); fuzz[13]'s limit is 9.6e-5 -- still inside the tests' own
1e-4.  Flipped pixels: more than 2 x baseline + 3 fails (the forward has no atomics: deterministic per build).
Refresh: tools/update_parity_baseline.py (requires >= 5 sessions and a reason; logged)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASELINE = os.path.join(ROOT, 'tests', 'golden', 'parity_observed_baseline.json')
FACTOR, FACTOR_FEW, MIN_RUNS, SPREAD, FLOOR = 2.0, 3.0, 5, 3.0, 3e-5
MAX_FIELDS = ('max_err', 'frac_over_tol', 'flipped_pixels', 'flipped_max_margin', 'traced_pixels', 'traced_rows',
              'rows_touching_a_flip', 'untraced_max')


def session_worst(records):
    """one session's records -> {(test, name): record with every MAX_FIELDS entry the maximum over the observations sharing
    the key}; only the census comparisons (those with `untraced_max`) are gated"""
    out = {}
    for r in records:
        if 'untraced_max' not in r:
            continue
        k = (r.get('test', ''), r['name'])
        cur = out.get(k)
        if cur is None:
            cur = out[k] = {f: r[f] for f in ('test', 'name', 'elements', 'tol') if f in r}
        for f in MAX_FIELDS:
            if f in r:
                cur[f] = max(cur.get(f, r[f]), r[f])
    return out


def aggregate_sessions(sessions):
    """list of sessions (each a list of records) -> baseline rows: per key the max over the sessions of every MAX_FIELDS entry,
    `untraced_min` = the SMALLEST session worst case, `runs` = the number of sessions that held the key"""
    out = {}
    for records in sessions:
        for k, r in session_worst(records).items():
            cur = out.get(k)
            if cur is None:
                cur = out[k] = dict(r, untraced_min=r['untraced_max'], runs=0)
            for f in MAX_FIELDS:
                if f in r:
                    cur[f] = max(cur.get(f, r[f]), r[f])
            cur['untraced_min'] = min(cur['untraced_min'], r['untraced_max'])
            cur['runs'] += 1
    return out


def limit(b):
    """the largest `untraced_max` an observation may show against baseline row b"""
    bmax = b['untraced_max']
    bmin = b.get('untraced_min', bmax)
    factor = FACTOR if b.get('runs', 1) >= MIN_RUNS else FACTOR_FEW
    return max(factor * bmax, bmax + SPREAD * (bmax - bmin), FLOOR)


def regressions(observed, baseline_rows):
    """messages, one per gated quantity of `observed` (one session's records) that left its baseline's band"""
    base = {(r['test'], r['name']): r for r in baseline_rows if 'untraced_max' in r}
    bad = []
    for k, r in session_worst(observed).items():
        b = base.get(k)
        if b is None:
            continue
        lim = limit(b)
        if r['untraced_max'] > lim:
            bad.append(f"{k[0]} [{k[1]}]: max error outside flips {r['untraced_max']:.2e} > limit {lim:.2e} "
                       f"(baseline max {b['untraced_max']:.2e}, min {b.get('untraced_min', b['untraced_max']):.2e} over "
                       f"{b.get('runs', 1)} sessions)")
        if 'flipped_pixels' in r and r['flipped_pixels'] > 2 * b.get('flipped_pixels', 0) + 3:
            bad.append(f"{k[0]} [{k[1]}]: {r['flipped_pixels']} flipped pixels, baseline {b.get('flipped_pixels', 0)}")
    return bad


def load_baseline(path=BASELINE):
    if not os.path.exists(path):
        return []
    with open(path) as f:
        return json.load(f)
