"""GPU parity tests of the rasterizer: HIP kernels (through the C ABI / reference-named functions) vs the CPU oracle
on identical seeded inputs.

Tolerances (written here, as the task requires):
  * integer outputs (radii, num_rendered, per-tile ranges, sorted point lists): bit-exact;
  * per-Gaussian floats produced by the preprocess kernel (pixel position, conic, depth, rgb): bit-exact -- the kernel
    is compiled without FMA contraction in the oracle's evaluation order;
  * STRICT build of the blend kernels (no contraction, oracle operation order, reproducible exp) against the oracle
    with exp_mode=1: image, opacity and n_contrib bit-exact; gradients <= 1e-5 max-norm relative (only the
    summation order over pixels differs);
  * FAST (product) build against the literal oracle (libm exp): max-norm relative error (the reference's
    get_rel_error, my_ext/utils/test_utils.py:6-21) <= 1e-4, the north-star tolerance, on all but a bounded handful
    of elements affected by threshold flips (helpers.assert_close_robust).
"""
import numpy as np
import pytest
import torch

from helpers import FLIP_HARD, FlipCensus, assert_close_robust, oracle_backward, oracle_forward, rel_err, scene_inputs, to_np

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _C():
    from sk_gs_amd import _C
    return _C


def hip_forward(act, rs, extras=None, colors=None, cov3D=None):
    C = _C()
    e = torch.Tensor([])
    use_sh = colors is None
    out = C.rasterize_gaussians(
        rs.image_height, rs.image_width, rs.tanfovx, rs.tanfovy, rs.sh_degree, rs.scale_modifier, rs.prefiltered,
        rs.debug, rs.colmap, rs.viewmatrix, rs.projmatrix, rs.campos, act['means3D'], act['opacity'],
        act['sh'] if use_sh else e, act['scales'] if cov3D is None else e, act['rotations'] if cov3D is None else e,
        extras, e if use_sh else colors, e if cov3D is None else cov3D)
    return out


def hip_backward(fwd, act, rs, dL_dcolor, dL_dopacity, extras=None, dL_dextra=None, colors=None, cov3D=None, gm=None,
                 gc=None, go=None):
    C = _C()
    e = torch.Tensor([])
    use_sh = colors is None
    R, color, opacity, radii, geom, binning, img, out_extra = fwd
    return C.rasterize_gaussians_backward(
        rs.scale_modifier, rs.tanfovx, rs.tanfovy, rs.sh_degree, rs.debug, rs.colmap, rs.viewmatrix, rs.projmatrix,
        rs.campos, act['means3D'], e if use_sh else colors, extras, act['scales'] if cov3D is None else e,
        act['rotations'] if cov3D is None else e, e if cov3D is None else cov3D, act['sh'] if use_sh else e, R, radii,
        opacity, dL_dcolor, dL_dopacity, dL_dextra, gm, gc, go, geom, binning, img)


def check_forward(o, act, rs, extras=None, colors=None, cov3D=None, strict=False):
    W, H = rs.image_width, rs.image_height
    P = act['means3D'].shape[0]
    ref = oracle_forward(o, act, rs, extras, colors, cov3D)
    fwd = hip_forward(act, rs, extras, colors, cov3D)
    R, color, opacity, radii, geom, binning, img, out_extra = fwd
    assert R == ref['num_rendered']
    np.testing.assert_array_equal(to_np(radii), ref['radii'])
    bufs = _C().unpack_buffers(W, H, P, geom, binning, img)
    vis = ref['radii'] > 0
    recs = to_np(bufs['recs'])
    # per-Gaussian records: bit-exact for visible Gaussians
    np.testing.assert_array_equal(recs[vis, 0:2], ref['geom']['means2D'][vis])
    np.testing.assert_array_equal(recs[vis, 2:5], ref['geom']['conic_opacity'][vis, 0:3])
    np.testing.assert_array_equal(recs[vis, 9], ref['geom']['depths'][vis])
    if colors is None:
        np.testing.assert_array_equal(recs[vis, 6:9], ref['geom']['rgb'][vis])
    # tile ranges and sorted lists: bit-exact
    offs = to_np(bufs['tile_offsets']).astype(np.int64)
    ranges = ref['binning']['ranges'].astype(np.int64)
    nonempty = ranges[:, 1] > ranges[:, 0]
    np.testing.assert_array_equal((offs[1:] - offs[:-1])[nonempty], (ranges[:, 1] - ranges[:, 0])[nonempty])
    np.testing.assert_array_equal(offs[:-1][nonempty], ranges[nonempty, 0])
    np.testing.assert_array_equal(to_np(bufs['point_list'])[:R].astype(np.uint32), ref['binning']['point_list'])
    # image
    nc = to_np(bufs['n_contrib']).astype(np.int64)
    if strict:
        np.testing.assert_array_equal(nc, ref['img']['n_contrib'].astype(np.int64))
        np.testing.assert_array_equal(to_np(color), ref['color'])
        np.testing.assert_array_equal(to_np(opacity), ref['opacity'])
        if extras is not None:
            np.testing.assert_array_equal(to_np(out_extra), ref['out_extra'])
    else:
        # every pixel over the tolerance must sit within 2e-6 of a branch of the reference walk (helpers.FlipCensus)
        # (exact census when the census walk -- one pixel per lane -- reproduces this image bit for bit, else every pixel
        # within rounding distance of a branch counts as possibly flipped)
        ccol, cop, cen = _C().render_census(W, H, geom, binning, img)
        exact = torch.equal(ccol, color) and torch.equal(cop, opacity)
        census = FlipCensus(o, ref, W, H, tol=TOL, name=f'P={P} {W}x{H}')
        census.check_image(color, opacity, cen if exact else None)
        ref['census'] = census
        # the last contributor differs only where the walk itself took a different branch
        nc_bad = nc != ref['img']['n_contrib'].astype(np.int64)
        assert not (nc_bad & ~np.asarray(census.flipped, dtype=bool).reshape(nc_bad.shape)).any()
        assert nc_bad.mean() <= max(2e-5, 1.5 / nc_bad.size)
        if extras is not None:
            # (extras blend like colours: a pixel over the tolerance must be one the census names)
            de = np.abs(to_np(out_extra).astype(np.float64) - np.asarray(ref['out_extra'], np.float64))
            de = (de / max(np.abs(ref['out_extra']).max(), 1e-30)).max(0)
            assert not ((de > TOL) & ~np.asarray(census.flipped, dtype=bool)).any(), f'out_extra: max {de.max():.2e}'
            assert de.max() <= FLIP_HARD
    return ref, fwd


GRAD_NAMES = ['dL_dmean2D', 'dL_dcolors', 'dL_dopacity', 'dL_dmeans3D', 'dL_dcov3D', 'dL_dsh', 'dL_dscales',
              'dL_drotations']


@pytest.mark.parametrize('strict', [True, False])
@pytest.mark.parametrize('ppl', [1, 2, 4])
@pytest.mark.parametrize('colmap', [True, False])
@pytest.mark.parametrize('P,W,H,seed,scale_mult', [(5000, 256, 256, 0, 2.0), (3000, 200, 136, 1, 3.0)])
def test_forward_backward_parity(oracle32, colmap, P, W, H, seed, scale_mult, ppl, strict):
    _C().set_pixels_per_lane(ppl)
    _C().set_strict_math(strict)
    oracle32.set_exp_mode(1 if strict else 0)
    try:
        act, rs, cam = scene_inputs(P, W, H, seed=seed, colmap=colmap, scale_mult=scale_mult, device='cuda')
        ref, fwd = check_forward(oracle32, act, rs, strict=strict)
        g = torch.Generator().manual_seed(seed + 7)
        dL_dcolor = torch.randn(3, H, W, generator=g).cuda()
        dL_dopacity = torch.randn(H, W, generator=g).cuda()
        gref = oracle_backward(oracle32, ref, act, rs, dL_dcolor, dL_dopacity)
        got = hip_backward(fwd, act, rs, dL_dcolor, dL_dopacity)
        for name, t in zip(GRAD_NAMES, got[:8]):
            if strict:
                err = rel_err(t, gref[name].reshape(to_np(t).shape))
                assert err <= 1e-5, (name, err)
            else:
                ref['census'].check_rows(t, gref[name], name)
        assert got[8] is None
    finally:
        _C().set_pixels_per_lane(0)
        _C().set_strict_math(False)
        oracle32.set_exp_mode(0)


@pytest.mark.parametrize('E', [1, 4])
def test_extras_in_main_pass(oracle32, E):
    P, W, H = 3000, 160, 160
    act, rs, cam = scene_inputs(P, W, H, seed=3, colmap=True, scale_mult=3.0, device='cuda')
    g = torch.Generator().manual_seed(11)
    extras = torch.randn(P, E, generator=g).cuda()
    ref, fwd = check_forward(oracle32, act, rs, extras=extras)
    dL_dcolor = torch.randn(3, H, W, generator=g).cuda()
    dL_dopacity = torch.randn(H, W, generator=g).cuda()
    dL_dextra = torch.randn(E, H, W, generator=g).cuda()
    gref = oracle_backward(oracle32, ref, act, rs, dL_dcolor, dL_dopacity, extras, dL_dextra)
    got = hip_backward(fwd, act, rs, dL_dcolor, dL_dopacity, extras, dL_dextra)
    assert_close_robust(got[8], gref['dL_dextras'], TOL, 1e-3, name='dL_dextras')
    assert_close_robust(got[0], gref['dL_dmean2D'], TOL, 1e-3, name='dL_dmean2D')
    assert_close_robust(got[5], gref['dL_dsh'], TOL, 1e-3, name='dL_dsh')


def test_precomputed_colors_and_cov(oracle32):
    P, W, H = 2000, 128, 128
    act, rs, cam = scene_inputs(P, W, H, seed=4, colmap=True, scale_mult=3.0, device='cuda')
    g = torch.Generator().manual_seed(5)
    colors = torch.rand(P, 3, generator=g).cuda()
    # covariance from the oracle's own forward (scale/rot path) reused as cov3D_precomp
    ref0 = oracle_forward(oracle32, act, rs)
    cov3D = torch.from_numpy(ref0['geom']['cov3D']).cuda()
    ref, fwd = check_forward(oracle32, act, rs, colors=colors, cov3D=cov3D)
    dL_dcolor = torch.randn(3, H, W, generator=g).cuda()
    dL_dopacity = torch.randn(H, W, generator=g).cuda()
    gref = oracle_backward(oracle32, ref, act, rs, dL_dcolor, dL_dopacity, colors=colors, cov3D=cov3D)
    got = hip_backward(fwd, act, rs, dL_dcolor, dL_dopacity, colors=colors, cov3D=cov3D)
    assert_close_robust(got[1], gref['dL_dcolors'], TOL, 1e-3, name='dL_dcolors')
    assert_close_robust(got[4], gref['dL_dcov3D'], TOL, 1e-3, name='dL_dcov3D')
    assert_close_robust(got[3], gref['dL_dmeans3D'], TOL, 1e-3, name='dL_dmeans3D')
    assert float(got[6].abs().max()) == 0.0 and float(got[7].abs().max()) == 0.0


def test_chained_input_grads(oracle32):
    """grad_means2D / grad_conic / grad_opacity handed in (from extra passes) are added to the results"""
    P, W, H = 1500, 96, 96
    act, rs, cam = scene_inputs(P, W, H, seed=6, colmap=True, scale_mult=3.0, device='cuda')
    ref, fwd = check_forward(oracle32, act, rs)
    g = torch.Generator().manual_seed(9)
    dL_dcolor = torch.randn(3, H, W, generator=g).cuda()
    dL_dopacity = torch.randn(H, W, generator=g).cuda()
    gm = torch.randn(P, 3, generator=g).cuda()
    gc = torch.randn(P, 2, 2, generator=g).cuda()
    go = torch.randn(P, 1, generator=g).cuda()
    gref = oracle_backward(oracle32, ref, act, rs, dL_dcolor, dL_dopacity, grad_means2D=gm, grad_conic=gc,
                           grad_opacity=go)
    got = hip_backward(fwd, act, rs, dL_dcolor, dL_dopacity, gm=gm.clone(), gc=gc.clone(), go=go.clone())
    assert_close_robust(got[0], gref['dL_dmean2D'], TOL, 1e-3, name='dL_dmean2D')
    assert_close_robust(got[2], gref['dL_dopacity'], TOL, 1e-3, name='dL_dopacity')
    assert_close_robust(got[3], gref['dL_dmeans3D'], TOL, 1e-3, name='dL_dmeans3D')
    assert_close_robust(got[6], gref['dL_dscales'], TOL, 1e-3, name='dL_dscales')


def test_other_extras_and_topk(oracle32):
    C = _C()
    P, W, H, E = 2000, 112, 80, 20
    act, rs, cam = scene_inputs(P, W, H, seed=8, colmap=True, scale_mult=3.0, device='cuda')
    ref, fwd = check_forward(oracle32, act, rs)
    R, color, opacity, radii, geom, binning, img, _ = fwd
    g = torch.Generator().manual_seed(13)
    extra = torch.randn(P, E, generator=g).cuda()
    pe = C.gaussian_rasterize_extra_forward(W, H, R, extra, geom, binning, img)
    assert tuple(pe.shape) == (W, H, E)
    pe_ref = oracle32.extra_forward(W, H, ref, to_np(extra))
    assert_close_robust(pe.reshape(H * W, E), pe_ref, TOL, name='pixel_extra')
    gpe = torch.randn(W, H, E, generator=g).cuda()
    ge, gm, gc, go = C.gaussian_rasterize_extra_backward(W, H, R, extra, opacity, gpe, geom, binning, img, None, None,
                                                         None)
    gref = oracle32.extra_backward(W, H, ref, to_np(extra), to_np(gpe))
    assert_close_robust(ge, gref['dL_dextra'], TOL, 1e-3, name='dL_dextra')
    assert_close_robust(gm, gref['dL_dmean2D'], TOL, 1e-3, name='gm')
    assert_close_robust(gc.reshape(P, 4), gref['dL_dconic'], TOL, 1e-3, name='gc')
    assert_close_robust(go, gref['dL_dopacity'], TOL, 1e-3, name='go')
    idx, w = C.gaussian_topk_weights(3, W, H, P, R, geom, binning, img)
    idx_ref, w_ref = oracle32.topk_weights(3, W, H, ref)
    assert_close_robust(w, w_ref, TOL, 1e-3, name='topk w')
    assert (to_np(idx) != idx_ref).mean() <= 1e-3
    vis = C.mark_visible(act['means3D'], rs.viewmatrix, rs.projmatrix, True)
    np.testing.assert_array_equal(to_np(vis), oracle32.mark_visible(to_np(act['means3D']), to_np(rs.viewmatrix), True))


@pytest.mark.parametrize('bucket', [0, 64])
def test_empty_scene_through_the_raw_c_abi(bucket):
    """P == 0 straight through skgs_rasterize_forward (the Python wrappers short-circuit an empty scene, the C ABI
    did not -- the sort launch that publishes the tile ranges / the blend kernels' group order was skipped and the blend
    launch indexed tiles through uninitialised words).  Poisoned buffers in, an all-zero image (plus background) out."""
    import ctypes as ct
    C = _C()
    lib = C.load_library()
    W, H = 200, 136
    act, rs, cam = scene_inputs(16, W, H, seed=0, colmap=True, device='cuda')
    e = torch.Tensor([]).cuda()
    a, keep, P, M, E = C._make_inputs(H, W, rs.tanfovx, rs.tanfovy, 3, 1.0, False, False, True, rs.viewmatrix, rs.projmatrix,
                                      rs.campos, act['means3D'][:0], act['opacity'][:0], act['sh'][:0], act['scales'][:0],
                                      act['rotations'][:0], None, e, e)
    assert P == 0
    bg = torch.tensor([0.25, 0.5, 0.75], device='cuda')
    a.background = bg.data_ptr()
    T = ((W + 15) // 16) * ((H + 15) // 16)
    a.tile_bucket_capacity = bucket
    cap = T * bucket if bucket else 1024
    poison = dict(dtype=torch.uint8, device='cuda')
    geom = torch.full((C._buffer_bytes(lib, 'geom', 0),), 0xCD, **poison)
    geom[:256].zero_()
    img = torch.full((C._buffer_bytes(lib, 'img', W, H),), 0xCD, **poison)
    binning = torch.full((C._buffer_bytes(lib, 'binning', cap),), 0xCD, **poison)
    color = torch.full((3, H, W), float('nan'), device='cuda')
    opac = torch.full((H, W), float('nan'), device='cuda')
    bufs = C._buffers(geom, binning, img)
    for _ in range(2):
        C._check(lib.skgs_rasterize_forward(ct.byref(a), ct.byref(bufs), None, ct.c_void_p(color.data_ptr()),
                                            ct.c_void_p(opac.data_ptr()), None, None, C._stream()))
    torch.cuda.synchronize()
    assert float(opac.abs().max()) == 0.0
    assert torch.equal(color, bg.view(3, 1, 1).expand(3, H, W))


def test_empty_and_culled(oracle32):
    C = _C()
    W, H = 64, 48
    act, rs, cam = scene_inputs(16, W, H, seed=0, colmap=True, device='cuda')
    # P = 0
    act0 = {k: v[:0] for k, v in act.items()}
    out = hip_forward(act0, rs)
    assert out[0] == 0 and float(out[1].abs().max()) == 0 and float(out[2].abs().max()) == 0
    # everything behind the camera
    act_b = dict(act)
    act_b['means3D'] = act['means3D'] + rs.campos * 3
    out = hip_forward(act_b, rs)
    assert out[0] == 0 and int(out[3].abs().max()) == 0 and float(out[1].abs().max()) == 0
    got = hip_backward(out, act_b, rs, torch.randn(3, H, W).cuda(), torch.randn(H, W).cuda())
    for t in got[:8]:
        assert float(t.abs().max()) == 0.0


@pytest.mark.parametrize('P,W,H,scale_mult,longest', [(6000, 64, 64, 6.0, 1024), (20000, 96, 96, 5.0, 4096),
                                                      (48000, 64, 64, 6.0, 8192)])
def test_long_tile_lists_all_sort_paths(oracle32, P, W, H, scale_mult, longest):
    """tile lists of 1k..10k entries: the merge-path instantiations (<= 2048 / 4096 / 8192 keys in LDS) and the
    global-memory network (> 8192 keys) behind the one-wave register sort; sorted lists and the strict image stay bit-exact"""
    _C().set_strict_math(True)
    oracle32.set_exp_mode(1)
    try:
        act, rs, cam = scene_inputs(P, W, H, seed=3, colmap=True, scale_mult=scale_mult, device='cuda')
        ref, fwd = check_forward(oracle32, act, rs, strict=True)
        r = ref['binning']['ranges'].astype(np.int64)
        assert (r[:, 1] - r[:, 0]).max() > longest
    finally:
        _C().set_strict_math(False)
        oracle32.set_exp_mode(0)


def test_async_capacity_overflow_flag():
    C = _C()
    P, W, H = 4000, 128, 128
    act, rs, cam = scene_inputs(P, W, H, seed=2, colmap=True, scale_mult=4.0, device='cuda')
    ref = hip_forward(act, rs)
    R = ref[0]
    C.config.sync_num_rendered = False
    try:
        C._capacity_hint[(P, W, H)] = R + 100
        out = hip_forward(act, rs)
        st = C.read_status(out[4])
        assert st['num_rendered'] == R and st['overflow'] == 0
        assert torch.equal(out[1], ref[1]) or rel_err(out[1], ref[1]) < 1e-6
        C._capacity_hint[(P, W, H)] = R // 2
        out = hip_forward(act, rs)
        st = C.read_status(out[4])
        assert st['num_rendered'] == R and st['overflow'] == 1
    finally:
        C.config.sync_num_rendered = True
        C._capacity_hint.clear()


def test_sync_free_operator_path_with_tile_buckets_matches_the_compact_lists():
    """``update_capacity_hint(..., longest_list)``: the sync-free operator path takes the bucket layout of the tile lists
    (four launches instead of seven to nine) -- same image bit for bit, same gradients up to the atomics' order, the pooled
    all-zero backward scratch is handed back clean, and a bucket that is too small raises the overflow flag"""
    C = _C()
    P, W, H = 20000, 320, 240
    act, rs, cam = scene_inputs(P, W, H, seed=6, colmap=True, scale_mult=2.0, device='cuda')
    g = torch.Generator().manual_seed(8)
    gc, go = torch.randn(3, H, W, generator=g).cuda(), torch.randn(H, W, generator=g).cuda()
    C.config.sync_num_rendered = True
    ref = hip_forward(act, rs)
    gref = hip_backward(ref, act, rs, gc, go)
    R, longest = ref[0], C.read_status(ref[4])['max_tile_count']
    try:
        C.config.sync_num_rendered = False
        C.update_capacity_hint(P, W, H, R, longest)
        assert C._bucket_hint[(P, W, H)] >= longest
        for _ in range(3):  # repeated calls: the pooled scratch must come back all zero every time
            out = hip_forward(act, rs)
            st = C.read_status(out[4])
            assert st['overflow'] == 0
            assert torch.equal(out[1], ref[1]) and torch.equal(out[2], ref[2]) and torch.equal(out[3], ref[3])
            grads = hip_backward(out, act, rs, gc, go)
            for name, a, b in zip(GRAD_NAMES, grads, gref):
                assert rel_err(a, b) <= 2e-5, name
        for ws in C._zero_ws.values():
            assert int(ws.view(torch.int32).abs().max()) == 0
        C._bucket_hint[(P, W, H)] = 64 * max(1, longest // 128)  # too small for the longest list
        out = hip_forward(act, rs)
        assert C.read_status(out[4])['overflow'] == 1
    finally:
        C.config.sync_num_rendered = True
        C._capacity_hint.clear(), C._bucket_hint.clear()


@pytest.mark.parametrize('bucket', [False, True])
def test_compiled_marshalling_of_the_operator_path_equals_the_ctypes_one(bucket):
    """`_skgs_torch.so` (csrc/torch_ops.cpp) and the ctypes code of `_C.py` fill the same structs and make the same C-ABI
    calls: sync-free forward bit-identical (image, opacity, radii, status words), backward identical up to the atomics'
    order; extras and a non-contiguous / double input take the same conversions"""
    C = _C()
    assert C._torch_ops() is not None, 'sk_gs_amd/_skgs_torch.so is not built'
    P, W, H = 12000, 200, 152
    act, rs, cam = scene_inputs(P, W, H, seed=12, colmap=True, scale_mult=2.0, device='cuda')
    g = torch.Generator().manual_seed(4)
    gc, go = torch.randn(3, H, W, generator=g).cuda(), torch.randn(H, W, generator=g).cuda()
    C.config.sync_num_rendered = True
    ref = hip_forward(act, rs)
    R, longest = ref[0], C.read_status(ref[4])['max_tile_count']
    act2 = dict(act)
    act2['means3D'] = act['means3D'].double()                    # converted to float32
    act2['scales'] = act['scales'].t().contiguous().t()        # made contiguous
    try:
        C.config.sync_num_rendered = False
        C.update_capacity_hint(P, W, H, R, longest if bucket else 0)
        results = {}
        for compiled in (False, True):
            C._ops = None if compiled else False
            out = hip_forward(act2, rs)
            assert out[0] == -1
            st = C.read_status(out[4])
            assert st['overflow'] == 0 and (bucket or st['num_rendered'] == R)
            results[compiled] = (out, hip_backward(out, act2, rs, gc, go))
        (fa, ga), (fb, gb) = results[False], results[True]
        for i in (1, 2, 3):
            assert torch.equal(fa[i], fb[i])
        assert torch.equal(fb[1], ref[1])
        for name, a, b in zip(GRAD_NAMES, ga, gb):
            assert a.shape == b.shape and rel_err(b, a) <= 2e-5, name
    finally:
        C._ops = None
        C.config.sync_num_rendered = True
        C._capacity_hint.clear(), C._bucket_hint.clear()


def test_blend_kernels_walk_the_tile_groups_heaviest_first_and_the_order_changes_nothing():
    """the group order written by the sort launch (binning.hip::tile_order_job) is a permutation of the groups of 8 tiles,
    by total list length descending (ties: lower id first); with the order switched off (raster order) the forward is
    bit-identical and the backward differs only by the order of its atomic additions"""
    C = _C()
    P, W, H = 30000, 400, 304
    act, rs, cam = scene_inputs(P, W, H, seed=9, colmap=True, scale_mult=1.5, device='cuda')
    g = torch.Generator().manual_seed(2)
    gc, go = torch.randn(3, H, W, generator=g).cuda(), torch.randn(H, W, generator=g).cuda()
    try:
        C.set_tile_order(1)
        fwd = hip_forward(act, rs)
        bufs = C.unpack_buffers(W, H, P, fwd[4], fwd[5], fwd[6])
        T = ((W + 15) // 16) * ((H + 15) // 16)
        G = (T + 7) // 8
        counts = torch.zeros(G * 8, dtype=torch.long, device='cuda')
        counts[:T] = bufs['tile_counts'].long()
        weight = counts.view(G, 8).sum(1)
        order = bufs['group_order'].long()
        assert sorted(order.tolist()) == list(range(G))
        want = sorted(range(G), key=lambda i: (-int(weight[i]), i))
        assert order.tolist() == want
        grads = hip_backward(fwd, act, rs, gc, go)
        C.set_tile_order(0)
        fwd0 = hip_forward(act, rs)
        assert C.unpack_buffers(W, H, P, fwd0[4], fwd0[5], fwd0[6])['group_order'].tolist() == list(range(G))
        assert torch.equal(fwd0[1], fwd[1]) and torch.equal(fwd0[2], fwd[2]) and torch.equal(fwd0[3], fwd[3])
        grads0 = hip_backward(fwd0, act, rs, gc, go)
        for name, a, b in zip(GRAD_NAMES, grads, grads0):
            assert rel_err(a, b) <= 2e-5, name
    finally:
        C.set_tile_order(1)


def test_single_gaussian_closed_form():
    """one isotropic Gaussian at the image centre: alpha(x) = o * exp(-r^2 / (2 s^2)) with s^2 = (f*sigma/z)^2 + 0.3"""
    import math
    from sk_gs_amd import scene
    W = H = 64
    cam = scene.make_camera(W, H, eye=torch.tensor([0., 0., -4.]))
    rs = scene.raster_settings_from_camera(cam, sh_degree=0, colmap=True, device='cuda')
    sigma, op = 0.05, 0.8
    act = dict(means3D=torch.zeros(1, 3).cuda(), scales=torch.full((1, 3), sigma).cuda(),
               rotations=torch.tensor([[0., 0., 0., 1.]]).cuda(), opacity=torch.tensor([[op]]).cuda(),
               sh=torch.zeros(1, 1, 3).cuda())
    act['sh'][0, 0] = torch.tensor([1.0, 0.5, -0.2])
    out = hip_forward(act, rs)
    opacity = to_np(out[2])
    focal = W / (2 * rs.tanfovx)
    s2 = (focal * sigma / 4.0) ** 2 + 0.3
    ys, xs = np.mgrid[0:H, 0:W]
    cx = ((0 + 1.0) * W - 1.0) * 0.5
    r2 = (xs - cx) ** 2 + (ys - cx) ** 2
    alpha = np.minimum(0.99, op * np.exp(-0.5 * r2 / s2))
    alpha[alpha < 1 / 255] = 0
    assert np.abs(opacity - alpha).max() < 2e-5
    rgb = np.maximum(0.28209479177387814 * np.array([1.0, 0.5, -0.2]) + 0.5, 0)
    assert np.abs(to_np(out[1]) - rgb[:, None, None] * alpha[None]).max() < 2e-5
