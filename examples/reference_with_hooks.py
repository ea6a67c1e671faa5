#!/usr/bin/env python3
"""Run the UNMODIFIED reference (dnvtmf/SK_GS) on an MI355X: its own `train.py`, its own `networks/`, none of its files touched.

    python examples/reference_with_hooks.py /path/to/SK_GS [--no-accelerate] [--check] -- <the reference's own train.py arguments>

What happens, in this order (INTEGRATION.md sections 1-4):

  1. the checkout goes on `sys.path`, then `sk_gs_amd.install_reference_hooks(accelerate=True)` BEFORE anything of the reference is imported: its compiled extension (`my_ext._C._C`: the
     rasterizer, the frequency encoder, simple_knn under their pybind names) and the three packages it imports that do not exist for
     ROCm (`diff_gaussian_rasterization`, `lietorch`, `pytorch3d.ops`) resolve to this package.
  2. `import train` -- the reference's module, from the given checkout.
  3. (unless --no-accelerate) a post-import hook has applied `sk_gs_amd.accelerate_reference()` by then: seven pieces of the training
     step that are long chains of small torch launches get a fast path with the same arguments and results (loss, kinematic chain, LBS weights, both
     deform networks, the rasterizer adapter's swizzle, `torch.optim.Adam.step`) -- and (round 6) `SkeletonGaussianSplatting.render` +
     `ImageLoss.forward` + `SSIM_Loss.forward` route stages `sk` and `sp` into the package's whole fused step on the model's own parameters
     (sk_gs_amd/reference_fused.py; every call outside its conditions runs the reference's own `render`).
  4. `train.GaussianTrainTask().run()` -- the reference's own entry point (train.py:381-382) with the arguments behind `--`.

`--check`: stop after step 3 and print what was hooked and patched (no GPU needed: what tests/test_host_cpu.py runs in the build container).
The whole iteration restated on the same hooks is timed by `bench.py --reference-loop hooks | accelerated | fused` (133 -> 514 -> 2 003 it/s
at config #1; the reference's own speed note, train.py:383-389, has 1000 steps in 15-24 s).
"""
import os
import sys


def main(argv):
    if not argv or argv[0].startswith('-'):
        sys.exit(__doc__)
    ref = os.path.abspath(argv[0])
    rest = argv[1:]
    own, theirs = (rest[:rest.index('--')], rest[rest.index('--') + 1:]) if '--' in rest else (rest, [])
    assert os.path.isfile(os.path.join(ref, 'train.py')) and os.path.isdir(os.path.join(ref, 'my_ext')), f'{ref}: not a checkout of the reference'
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    sys.path.insert(0, ref)          # (before the hooks: install_as_my_ext_C plants a stand-in `my_ext` only where the reference's cannot be found)
    import sk_gs_amd
    fast = '--no-accelerate' not in own
    hooked = sk_gs_amd.install_reference_hooks(accelerate=fast)   # (accelerate: the fast paths are applied as the reference's modules arrive)
    import train  # noqa: E402  (the reference's)
    patched = sk_gs_amd.accelerate_reference() if fast else []    # (idempotent: here only to list what the post-import hook has patched)
    if '--check' in own:
        print('hooks    :', hooked)
        print('patched  :', patched)
        print('REFERENCE-READY')
        return
    sys.argv = [os.path.join(ref, 'train.py')] + theirs
    train.GaussianTrainTask().run()


if __name__ == '__main__':
    main(sys.argv[1:])
