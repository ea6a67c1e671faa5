#!/usr/bin/env python3
"""Dense golden values of the reference's learning-rate schedule, by IMPORTING the reference (build container only):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_lr.py
``lr_schedule_dense.npz``: ``get_expon_lr_func`` (networks/gaussian_splatting.py:56-84) as float64 -- what ``update_learning_rate``
puts into ``group['lr']`` before every train step (train.py:140-141, gaussian_splatting.py:455-470, sk_gs.py:611-632) -- at every
step 0..64 and at every 89th step up to 45 000, for the two schedules the shipped configuration builds (`xyz`: cfg.lr x
lr_position_init x lr_spatial_scale -> lr_position_final over 30 000 steps, exps/default.yaml:60-63 with lr_spatial_scale 5,
sk_gs.py:583; `deform`: sk_gs.py:611-614 over lr_deform_max_steps 40 000) and one with an ease-in.  The device evaluation
(csrc/adam_update.h::lr_schedule_eval) must reproduce float32(value) bit for bit (tests/test_gpu_optim.py)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden  # noqa: E402  (stub finder only)


def main():
    assert os.path.isdir(make_golden.REF)
    sys.dont_write_bytecode = True
    sys.meta_path.insert(0, make_golden._Finder())
    sys.path.insert(0, make_golden.REF)
    import warnings
    warnings.filterwarnings('ignore')
    from networks.gaussian_splatting import get_expon_lr_func
    lr, scale = 1e-3, 5.0
    sets = {
        'xyz': dict(lr_init=lr * 0.16 * scale, lr_final=lr * 0.0016 * scale, lr_delay_steps=0, lr_delay_mult=0.01, max_steps=30_000),
        'deform': dict(lr_init=1.0 * lr * scale * 0.16, lr_final=lr * 0.0016 * 1.0, lr_delay_steps=0, lr_delay_mult=0.01, max_steps=40_000),
        'eased': dict(lr_init=1e-2, lr_final=1e-4, lr_delay_steps=200, lr_delay_mult=0.1, max_steps=5000),
    }
    steps = np.array(sorted(set(range(0, 65)) | set(range(0, 45_001, 89)) | {199, 200, 201, 4999, 5000, 5001, 29_999, 30_000, 30_001, 40_000}), np.int64)
    rec = {'steps': steps}
    for name, c in sets.items():
        f = get_expon_lr_func(**c)
        rec[name] = np.array([f(int(t)) for t in steps], np.float64)
        rec[name + '_args'] = np.array([c['lr_init'], c['lr_final'], c['lr_delay_steps'], c['lr_delay_mult'], c['max_steps']], np.float64)
    np.savez_compressed(os.path.join(HERE, 'lr_schedule_dense.npz'), **rec)
    print('wrote lr_schedule_dense.npz', len(steps), 'steps x', len(sets), 'schedules')


if __name__ == '__main__':
    main()
