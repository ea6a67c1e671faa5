"""ctypes/numpy front-end of the CPU oracle (oracle/skgs_oracle.c).

TEST INFRASTRUCTURE ONLY.  Imported by tests/, by bench.py's ``cpu_baseline`` leg and by
``__graft_entry__.smoke()`` as the checker -- never by the product package ``sk_gs_amd``.

Every method mirrors one reference kernel / host function (file:line in the C source).  Inputs are numpy
arrays (anything ``np.asarray`` accepts); outputs are freshly allocated numpy arrays.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
BLOCK = 16


def build(out_dir: Optional[str] = None, native: bool = False) -> str:
    """Compile the oracle with gcc (idempotent). Returns the directory holding the .so files."""
    out = out_dir or os.path.join(_HERE, '_build')
    args = ['make', '-s', '-C', _HERE, f'OUT={out}']
    if native:
        args.append('native')
    subprocess.run(args, check=True)
    return out


def _lib_path(name: str, out_dir: Optional[str] = None) -> str:
    out = out_dir or os.path.join(_HERE, '_build')
    path = os.path.join(out, name)
    if not os.path.exists(path):
        build(out_dir)
    return path


def _p(a: Optional[np.ndarray]):
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


def freq_encode_forward(inputs: np.ndarray, degree: int) -> np.ndarray:
    """``kernel_freq`` (my_ext/_C/src/nerf/freqencoder.cu:7-31), element by element: column c < D copies the input,
    otherwise ``col = c / D - 1, d = c % D, freq = col / 2`` and the value is ``sin(scalbn(x[d], freq) + (col % 2) pi/2)``
    in fp32 (the CUDA kernel uses the ``__sinf`` intrinsic: bit-level behaviour of the compiled reference is unpinned).
    Test infrastructure only."""
    x = np.ascontiguousarray(inputs, dtype=np.float32)
    B, D = x.shape
    Cn = D + 2 * D * degree
    out = np.empty((B, Cn), np.float32)
    half_pi = np.float32(3.141592653589793 / 2)
    for c in range(Cn):
        if c < D:
            out[:, c] = x[:, c]
        else:
            col, d = c // D - 1, c % D
            arg = np.ldexp(x[:, d], col // 2).astype(np.float32) + np.float32(col % 2) * half_pi
            out[:, c] = np.sin(arg.astype(np.float32)).astype(np.float32)
    return out


def freq_encode_backward(grad: np.ndarray, outputs: np.ndarray, D: int, degree: int) -> np.ndarray:
    """``kernel_freq_backward`` (freqencoder.cu:36-60): ``g_x[d] = g[d] + sum_f 2^f (g[s + d] out[s + D + d] -
    g[s + D + d] out[s + d])`` with ``s = D + 2 f D`` (the saved cos / sin are read back from ``outputs``)."""
    g = np.ascontiguousarray(grad, dtype=np.float32)
    o = np.ascontiguousarray(outputs, dtype=np.float32)
    r = g[:, :D].copy()
    for f in range(degree):
        s = D + 2 * f * D
        r = r + np.float32(2.0 ** f) * (g[:, s:s + D] * o[:, s + D:s + 2 * D] - g[:, s + D:s + 2 * D] * o[:, s:s + D])
    return r.astype(np.float32)


class Oracle:
    """fp32 oracle (``dtype='f32'``) or its fp64 twin (``dtype='f64'``)."""

    def __init__(self, dtype: str = 'f32', lib_path: Optional[str] = None):
        assert dtype in ('f32', 'f64')
        self.dtype = np.float32 if dtype == 'f32' else np.float64
        self.prefix = 'skgs_oracle_' if dtype == 'f32' else 'skgs_oracle64_'
        self.creal = C.c_float if dtype == 'f32' else C.c_double
        name = 'libskgs_oracle.so' if dtype == 'f32' else 'libskgs_oracle64.so'
        self.lib = C.CDLL(lib_path or _lib_path(name))
        self._fn('bin_and_sort').restype = C.c_int64
        self._fn('getHigherMsb').restype = C.c_uint32
        self._fn('num_threads').restype = C.c_int

    # ------------------------------------------------------------------ helpers
    def _fn(self, name):
        return getattr(self.lib, self.prefix + name)

    def r(self, a, shape=None) -> Optional[np.ndarray]:
        """to contiguous REAL array; empty / None -> None (the reference's null-pointer convention)"""
        if a is None:
            return None
        a = np.ascontiguousarray(np.asarray(a), dtype=self.dtype)
        if a.size == 0:
            return None
        if shape is not None:
            a = a.reshape(shape)
        return a

    def num_threads(self) -> int:
        return int(self._fn('num_threads')())

    def set_num_threads(self, n: int):
        """OpenMP threads of the following calls (bench.py times the oracle on all cores and on one)"""
        self._fn('set_num_threads')(C.c_int(int(n)))

    def set_exp_mode(self, mode: int):
        """0: libm exp (literal restatement); 1: reproducible double-arithmetic exp shared with the strict HIP build"""
        self._fn('set_exp_mode')(C.c_int(mode))

    def exp_array(self, x):
        x = self.r(x)
        out = np.zeros_like(x)
        self._fn('exp_array')(C.c_int(x.size), _p(x), _p(out))
        return out

    # ------------------------------------------------------------------ sub-step hooks (pinned by tests/golden)
    def cov3d_array(self, scales, rots, mod=1.0, colmap=True):
        scales, rots = self.r(scales), self.r(rots)
        out = np.zeros((scales.shape[0], 6), self.dtype)
        self._fn('cov3d_array')(C.c_int(scales.shape[0]), _p(scales), self.creal(mod), _p(rots), C.c_int(int(colmap)),
                                _p(out))
        return out

    def cov2d_array(self, means, cov3D, viewmatrix, fx, fy, tanfovx, tanfovy, colmap=True):
        means, cov3D = self.r(means), self.r(cov3D)
        out = np.zeros((means.shape[0], 3), self.dtype)
        self._fn('cov2d_array')(C.c_int(means.shape[0]), _p(means), _p(cov3D), _p(self.r(viewmatrix)), self.creal(fx),
                                self.creal(fy), self.creal(tanfovx), self.creal(tanfovy), C.c_int(int(colmap)), _p(out))
        return out

    def sh_array(self, deg, means, campos, shs):
        means, shs = self.r(means), self.r(shs)
        P, M = shs.shape[0], shs.shape[1]
        rgb = np.zeros((P, 3), self.dtype)
        clamped = np.zeros((P, 3), np.uint8)
        self._fn('sh_array')(C.c_int(P), C.c_int(deg), C.c_int(M), _p(means), _p(self.r(campos)), _p(shs), _p(rgb),
                             _p(clamped))
        return rgb, clamped

    def bone_chain_forward(self, ancestors, root, sk_r, joints, global_T=None):
        """kinematic + skeleton_warp_SE3, literal pointer jumping (sk_gs.py:1069-1107,193-206)"""
        ancestors = np.ascontiguousarray(ancestors, dtype=np.int64)
        M, L = ancestors.shape
        out = np.zeros((M, 7), self.dtype)
        self._fn('bone_chain_forward')(C.c_int(M), C.c_int(L), C.c_int(root), _p(ancestors), _p(self.r(sk_r)),
                                       _p(self.r(joints)), _p(self.r(global_T)), _p(out))
        return out

    # ------------------------------------------------------------------ stages
    def preprocess_forward(self, means3D, scales, rotations, opacities, shs, viewmatrix, projmatrix, campos, W, H,
                           tanfovx, tanfovy, sh_degree, scale_modifier=1.0, colmap=True, cov3D_precomp=None,
                           colors_precomp=None):
        """preprocessCUDA{,_colmap}: gaussian_preprocess{,_colmap}.cu"""
        means3D = self.r(means3D)
        P = means3D.shape[0]
        scales, rotations = self.r(scales), self.r(rotations)
        opacities = self.r(opacities).reshape(-1)
        shs = self.r(shs)
        M = 0 if shs is None else shs.shape[1]
        cov3D_precomp, colors_precomp = self.r(cov3D_precomp), self.r(colors_precomp)
        out = dict(
            radii=np.zeros(P, np.int32), means2D=np.zeros((P, 2), self.dtype), depths=np.zeros(P, self.dtype),
            cov3D=np.zeros((P, 6), self.dtype), rgb=np.zeros((P, 3), self.dtype),
            conic_opacity=np.zeros((P, 4), self.dtype), tiles_touched=np.zeros(P, np.uint32),
            clamped=np.zeros((P, 3), np.uint8))
        self._fn('preprocess_forward')(
            C.c_int(P), C.c_int(sh_degree), C.c_int(M), _p(means3D), _p(scales), self.creal(scale_modifier),
            _p(rotations), _p(opacities), _p(shs), _p(cov3D_precomp), _p(colors_precomp),
            _p(self.r(viewmatrix)), _p(self.r(projmatrix)), _p(self.r(campos)), C.c_int(W), C.c_int(H),
            self.creal(tanfovx), self.creal(tanfovy), C.c_int(int(colmap)),
            _p(out['radii']), _p(out['means2D']), _p(out['depths']), _p(out['cov3D']), _p(out['rgb']),
            _p(out['conic_opacity']), _p(out['tiles_touched']), _p(out['clamped']))
        return out

    def bin_and_sort(self, W, H, geom):
        """InclusiveSum + duplicateWithKeys + SortPairs + identifyTileRanges: gaussian_rasterizer_forward.cu:45-94"""
        P = geom['radii'].shape[0]
        T = ((W + BLOCK - 1) // BLOCK) * ((H + BLOCK - 1) // BLOCK)
        offsets = np.zeros(P, np.uint32)
        ranges = np.zeros((T, 2), np.uint32)
        fn = self._fn('bin_and_sort')
        args = [C.c_int(P), C.c_int(W), C.c_int(H), _p(geom['means2D']), _p(geom['depths']), _p(geom['radii']),
                _p(geom['tiles_touched']), _p(offsets)]
        R = int(fn(*args, C.c_int64(0), None, None, _p(ranges)))
        keys = np.zeros(max(R, 1), np.uint64)
        plist = np.zeros(max(R, 1), np.uint32)
        R2 = int(fn(*args, C.c_int64(R), _p(keys), _p(plist), _p(ranges)))
        assert R2 == R
        return dict(num_rendered=R, point_offsets=offsets, point_list_keys=keys[:R], point_list=plist[:R],
                    ranges=ranges)

    def render_forward(self, W, H, geom, binning, colors=None, extra=None):
        """renderCUDA_forward: gaussian_render.cu:16-112"""
        colors = geom['rgb'] if colors is None else self.r(colors)
        extra = self.r(extra)
        E = 0 if extra is None else extra.shape[1]
        out = dict(color=np.zeros((3, H, W), self.dtype), opacity=np.zeros((H, W), self.dtype),
                   n_contrib=np.zeros((H, W), np.uint32),
                   out_extra=np.zeros((E, H, W), self.dtype) if E else None)
        self._fn('render_forward')(
            C.c_int(W), C.c_int(H), C.c_int(E), _p(binning['ranges']), _p(binning['point_list']),
            _p(geom['means2D']), _p(colors), _p(geom['conic_opacity']), _p(extra), _p(out['n_contrib']),
            _p(out['color']), _p(out['opacity']), _p(out['out_extra']))
        return out

    def render_margins(self, W, H, fwd, with_cause=False):
        """flip census of the forward walk (test infrastructure, see the C source): per-pixel distance [H,W] to the
        nearest data-dependent branch of renderCUDA_forward, in units of the local rounding error"""
        geom, binning = fwd['geom'], fwd['binning']
        margin = np.zeros((H, W), self.dtype)
        cause = np.zeros((H, W), np.uint8) if with_cause else None  # 0: power > 0, 1: alpha < 1/255, 2: T stop
        self._fn('render_margins')(
            C.c_int(W), C.c_int(H), _p(binning['ranges']), _p(binning['point_list']), _p(geom['means2D']),
            _p(geom['conic_opacity']), _p(margin), _p(cause))
        return (margin, cause) if with_cause else margin

    def render_census(self, W, H, fwd):
        """uint32 [H,W,2]: per pixel the number of list entries the reference walk blends and the fingerprint of their
        list positions -- the counterpart of the HIP library's ``skgs_render_census``"""
        geom, binning = fwd['geom'], fwd['binning']
        out = np.zeros((H, W, 2), np.uint32)
        self._fn('render_census')(C.c_int(W), C.c_int(H), _p(binning['ranges']), _p(binning['point_list']),
                                  _p(geom['means2D']), _p(geom['conic_opacity']), _p(out))
        return out

    def render_touching(self, W, H, fwd, pix_mask):
        """bool [P]: the Gaussians that (nearly) contribute to a pixel of `pix_mask` [H,W]"""
        geom, binning = fwd['geom'], fwd['binning']
        flag = np.zeros(geom['radii'].shape[0], np.uint8)
        mask = np.ascontiguousarray(np.asarray(pix_mask).reshape(H, W), dtype=np.uint8)
        self._fn('render_touching')(
            C.c_int(W), C.c_int(H), _p(binning['ranges']), _p(binning['point_list']), _p(geom['means2D']),
            _p(geom['conic_opacity']), _p(mask), _p(flag))
        return flag.astype(bool)

    def render_backward(self, W, H, geom, binning, img, dL_dcolor, dL_dopacity, colors=None, extra=None,
                        dL_dextra=None, grad_means2D=None, grad_conic=None, grad_opacity=None):
        """renderCUDA_backward: gaussian_render.cu:182-341 (accumulates into the optional grad_* inputs)"""
        P = geom['radii'].shape[0]
        colors = geom['rgb'] if colors is None else self.r(colors)
        extra = self.r(extra)
        dL_dextra = self.r(dL_dextra)
        E = extra.shape[1] if (extra is not None and dL_dextra is not None) else 0
        g = dict(
            dL_dmean2D=np.zeros((P, 3), self.dtype) if grad_means2D is None else self.r(grad_means2D).copy(),
            dL_dconic=np.zeros((P, 4), self.dtype) if grad_conic is None else self.r(grad_conic).reshape(P, 4).copy(),
            dL_dopacity=np.zeros((P, 1), self.dtype) if grad_opacity is None else self.r(grad_opacity).reshape(P, 1).copy(),
            dL_dcolors=np.zeros((P, 3), self.dtype),
            dL_dextras=np.zeros((P, E), self.dtype) if E else None)
        self._fn('render_backward')(
            C.c_int(P), C.c_int(W), C.c_int(H), C.c_int(E), _p(binning['ranges']), _p(binning['point_list']),
            _p(geom['means2D']), _p(geom['conic_opacity']), _p(colors), _p(extra), _p(self.r(img['opacity'])),
            _p(img['n_contrib']), _p(self.r(dL_dcolor)), _p(dL_dextra), _p(self.r(dL_dopacity)),
            _p(g['dL_dmean2D']), _p(g['dL_dconic']), _p(g['dL_dopacity']), _p(g['dL_dcolors']), _p(g['dL_dextras']))
        return g

    def preprocess_backward(self, means3D, scales, rotations, shs, viewmatrix, projmatrix, campos, W, H, tanfovx,
                            tanfovy, sh_degree, geom, dL_dmean2D, dL_dconic, dL_dcolors, scale_modifier=1.0,
                            colmap=True, cov3D_precomp=None):
        """computeCov2DCUDA + preprocessCUDA_backward: gaussian_preprocess{,_colmap}.cu"""
        means3D = self.r(means3D)
        P = means3D.shape[0]
        shs = self.r(shs)
        M = 0 if shs is None else shs.shape[1]
        scales, rotations = self.r(scales), self.r(rotations)
        cov3D_precomp = self.r(cov3D_precomp)
        cov3Ds = geom['cov3D'] if cov3D_precomp is None else cov3D_precomp
        g = dict(dL_dmeans3D=np.zeros((P, 3), self.dtype), dL_dcov3D=np.zeros((P, 6), self.dtype),
                 dL_dsh=np.zeros((P, M, 3), self.dtype), dL_dscales=np.zeros((P, 3), self.dtype),
                 dL_drotations=np.zeros((P, 4), self.dtype))
        dL_dcolors = self.r(dL_dcolors).copy()
        self._fn('preprocess_backward')(
            C.c_int(P), C.c_int(sh_degree), C.c_int(M), _p(means3D), _p(geom['radii']), _p(shs), _p(geom['clamped']),
            _p(scales), _p(rotations), self.creal(scale_modifier), _p(cov3Ds), _p(self.r(viewmatrix)),
            _p(self.r(projmatrix)), C.c_int(W), C.c_int(H), self.creal(tanfovx), self.creal(tanfovy),
            _p(self.r(campos)), C.c_int(int(colmap)), _p(self.r(dL_dmean2D)), _p(self.r(dL_dconic).reshape(P, 4)),
            _p(g['dL_dmeans3D']), _p(dL_dcolors), _p(g['dL_dcov3D']), _p(g['dL_dsh']), _p(g['dL_dscales']),
            _p(g['dL_drotations']))
        return g

    # ------------------------------------------------------------------ whole-op (mirrors the pybind entry points)
    def rasterize_forward(self, H, W, tanfovx, tanfovy, sh_degree, scale_modifier, colmap, viewmatrix, projmatrix,
                          campos, means3D, opacity, sh=None, scales=None, rotations=None, extras=None, colors=None,
                          cov3D_precomp=None):
        """RasterizeGaussiansCUDA / Rasterizer::forward: gaussian_rasterizer_forward.cu:157-315"""
        geom = self.preprocess_forward(means3D, scales, rotations, opacity, sh, viewmatrix, projmatrix, campos, W, H,
                                       tanfovx, tanfovy, sh_degree, scale_modifier, colmap, cov3D_precomp, colors)
        binning = self.bin_and_sort(W, H, geom)
        colors_r = self.r(colors)
        img = self.render_forward(W, H, geom, binning, colors_r, extras)
        return dict(num_rendered=binning['num_rendered'], color=img['color'], opacity=img['opacity'],
                    radii=geom['radii'], out_extra=img['out_extra'], geom=geom, binning=binning, img=img)

    def rasterize_backward(self, fwd, H, W, tanfovx, tanfovy, sh_degree, scale_modifier, colmap, viewmatrix,
                           projmatrix, campos, means3D, sh, scales, rotations, dL_dcolor, dL_dopacity, extras=None,
                           dL_dextra=None, colors=None, cov3D_precomp=None, grad_means2D=None, grad_conic=None,
                           grad_opacity=None):
        """RasterizeGaussiansBackwardCUDA / Rasterizer::backward: gaussian_rasterizer_backwrad.cu:148-261"""
        g1 = self.render_backward(W, H, fwd['geom'], fwd['binning'], fwd['img'], dL_dcolor, dL_dopacity, colors, extras,
                                  dL_dextra, grad_means2D, grad_conic, grad_opacity)
        g2 = self.preprocess_backward(means3D, scales, rotations, sh, viewmatrix, projmatrix, campos, W, H, tanfovx,
                                      tanfovy, sh_degree, fwd['geom'], g1['dL_dmean2D'], g1['dL_dconic'],
                                      g1['dL_dcolors'], scale_modifier, colmap, cov3D_precomp)
        out = dict(g1)
        out.update(g2)
        return out

    # ------------------------------------------------------------------ extras / top-k
    def extra_forward(self, W, H, fwd, extra):
        extra = self.r(extra)
        E = extra.shape[1]
        out = np.zeros((H * W, E), self.dtype)
        self._fn('render_extra_forward')(
            C.c_int(W), C.c_int(H), C.c_int(E), _p(fwd['binning']['ranges']), _p(fwd['binning']['point_list']),
            _p(fwd['geom']['means2D']), _p(fwd['geom']['conic_opacity']), _p(fwd['img']['n_contrib']), _p(extra),
            _p(out))
        return out

    def extra_backward(self, W, H, fwd, extra, grad_pixel_extra, grad_means2D=None, grad_conic=None,
                       grad_opacity=None):
        extra = self.r(extra)
        P, E = extra.shape
        g = dict(
            dL_dmean2D=np.zeros((P, 3), self.dtype) if grad_means2D is None else self.r(grad_means2D).copy(),
            dL_dconic=np.zeros((P, 4), self.dtype) if grad_conic is None else self.r(grad_conic).reshape(P, 4).copy(),
            dL_dopacity=np.zeros((P, 1), self.dtype) if grad_opacity is None else self.r(grad_opacity).reshape(P, 1).copy(),
            dL_dextra=np.zeros((P, E), self.dtype))
        self._fn('render_extra_backward')(
            C.c_int(P), C.c_int(W), C.c_int(H), C.c_int(E), _p(fwd['binning']['ranges']),
            _p(fwd['binning']['point_list']), _p(fwd['geom']['means2D']), _p(fwd['geom']['conic_opacity']),
            _p(self.r(fwd['img']['opacity'])), _p(fwd['img']['n_contrib']), _p(extra),
            _p(self.r(grad_pixel_extra).reshape(H * W, E)), _p(g['dL_dmean2D']), _p(g['dL_dconic']),
            _p(g['dL_dopacity']), _p(g['dL_dextra']))
        return g

    def topk_weights(self, topk, W, H, fwd):
        idx = np.full((H, W, topk), -1, np.int32)
        w = np.zeros((H, W, topk), self.dtype)
        self._fn('topk_weights')(
            C.c_int(topk), C.c_int(W), C.c_int(H), _p(fwd['binning']['ranges']), _p(fwd['binning']['point_list']),
            _p(fwd['geom']['means2D']), _p(fwd['geom']['conic_opacity']), _p(fwd['img']['n_contrib']), _p(idx), _p(w))
        return idx, w

    def mark_visible(self, means3D, viewmatrix, colmap=True):
        means3D = self.r(means3D)
        out = np.zeros(means3D.shape[0], np.uint8)
        self._fn('mark_visible')(C.c_int(means3D.shape[0]), _p(means3D), _p(self.r(viewmatrix)), C.c_int(int(colmap)),
                                 _p(out))
        return out.astype(bool)

    # ------------------------------------------------------------------ deform
    def lbs_deform_forward(self, points, weights, indices, bone_T, bone_drot, bone_dscale, xyz, log_scale, rot,
                           opacity_logit):
        """sk_gs.py:1147-1149,1162,1192-1203 + lie.h SE3 act"""
        points = self.r(points)
        P = points.shape[0]
        weights = self.r(weights)
        K = weights.shape[1]
        indices = np.ascontiguousarray(indices, dtype=np.int64)
        bone_T = self.r(bone_T)
        M = bone_T.shape[0]
        o = dict(means=np.zeros((P, 3), self.dtype), scales=np.zeros((P, 3), self.dtype),
                 rotations=np.zeros((P, 4), self.dtype), opacity=np.zeros((P, 1), self.dtype),
                 d_xyz=np.zeros((P, 3), self.dtype), d_rot=np.zeros((P, 4), self.dtype),
                 d_scale=np.zeros((P, 3), self.dtype))
        self._fn('lbs_deform_forward')(
            C.c_int(P), C.c_int(K), C.c_int(M), _p(points), _p(weights), _p(indices), _p(bone_T),
            _p(self.r(bone_drot)), _p(self.r(bone_dscale)), _p(self.r(xyz)), _p(self.r(log_scale)), _p(self.r(rot)),
            _p(self.r(opacity_logit).reshape(-1)), _p(o['means']), _p(o['scales']), _p(o['rotations']),
            _p(o['opacity']), _p(o['d_xyz']), _p(o['d_rot']), _p(o['d_scale']))
        return o

    def lbs_deform_backward(self, points, weights, indices, bone_T, bone_drot, bone_dscale, log_scale, rot,
                            opacity_logit, g_means, g_scales, g_rotations, g_opacity):
        points = self.r(points)
        P = points.shape[0]
        weights = self.r(weights)
        K = weights.shape[1]
        indices = np.ascontiguousarray(indices, dtype=np.int64)
        bone_T = self.r(bone_T)
        M = bone_T.shape[0]
        g = dict(g_weights=np.zeros((P, K), self.dtype), g_bone_T=np.zeros((M, 7), self.dtype),
                 g_bone_drot=np.zeros((M, 4), self.dtype), g_bone_dscale=np.zeros((M, 3), self.dtype),
                 g_xyz=np.zeros((P, 3), self.dtype), g_log_scale=np.zeros((P, 3), self.dtype),
                 g_rot=np.zeros((P, 4), self.dtype), g_opacity_logit=np.zeros((P, 1), self.dtype))
        self._fn('lbs_deform_backward')(
            C.c_int(P), C.c_int(K), C.c_int(M), _p(points), _p(weights), _p(indices), _p(bone_T),
            _p(self.r(bone_drot)), _p(self.r(bone_dscale)), _p(self.r(log_scale)), _p(self.r(rot)),
            _p(self.r(opacity_logit).reshape(-1)), _p(self.r(g_means)), _p(self.r(g_scales)), _p(self.r(g_rotations)),
            _p(self.r(g_opacity).reshape(-1)), _p(g['g_weights']), _p(g['g_bone_T']), _p(g['g_bone_drot']),
            _p(g['g_bone_dscale']), _p(g['g_xyz']), _p(g['g_log_scale']), _p(g['g_rot']), _p(g['g_opacity_logit']))
        return g

    def knn_bones(self, points, joints, K):
        points, joints = self.r(points), self.r(joints)
        P, dim = points.shape
        M = joints.shape[0]
        dist = np.zeros((P, K), self.dtype)
        idx = np.zeros((P, K), np.int64)
        self._fn('knn_bones')(C.c_int(P), C.c_int(M), C.c_int(K), C.c_int(dim), _p(points), _p(joints), _p(dist),
                              _p(idx))
        return dist, idx

    # ------------------------------------------------------------------------------------- per-iteration bookkeeping
    def densify_stats(self, radii, grad_means2D, xyz_gradient_accum, denom, max_radii2D):
        """numpy restatement of the densification statistics of one view (networks/sk_gs.py:1990-1997:
        ``mask = radii > 0; max_radii2D[mask] = max(max_radii2D[mask], radii[mask])``; networks/gaussian_splatting.py:
        503-513: ``xyz_gradient_accum[mask] += norm(grad[mask, :2]); denom[mask] += 1``).  Returns the three updated
        arrays (inputs untouched)."""
        radii = np.asarray(radii)
        g = self.r(grad_means2D).reshape(-1, 3)
        acc, den, mr = (self.r(x).copy() for x in (xyz_gradient_accum, denom, max_radii2D))
        mask = radii > 0
        mr[mask] = np.maximum(mr[mask], radii[mask].astype(self.dtype))
        nrm = np.sqrt(g[:, 0] * g[:, 0] + g[:, 1] * g[:, 1]).astype(self.dtype)
        acc.reshape(-1)[mask] += nrm[mask]
        den.reshape(-1)[mask] += 1
        return acc, den, mr

    def lbs_weights_kernel(self, nn_dist, indices, kernel_radius, kernel_weight=None, g_weights=None):
        """the `weighted_kernel` / `kernel` branch of calc_LBS_weight (networks/sk_gs.py:759-766):
        ``u = exp(-d / (2 r[idx]^2)) [* s[idx]] + 1e-7;  w = u / sum_k u`` with r = kernel_radius (= exp(_sp_radius),
        sk_gs.py:547-549) and s = kernel_weight (= sigmoid(_sp_weight)).  With ``g_weights`` also the analytic gradients
        (what autograd gives the reference): dict(g_dist [P,K], g_radius [M], g_weight [M])."""
        d, idx = self.r(nn_dist).astype(np.float64), np.asarray(indices)
        r, M = self.r(kernel_radius).astype(np.float64), len(kernel_radius)
        e = np.exp(-d / (2 * r[idx] ** 2))
        sk = self.r(kernel_weight).astype(np.float64)[idx] if kernel_weight is not None else np.ones_like(e)
        u = e * sk + 1e-7
        ssum = u.sum(axis=1, keepdims=True)
        w = u / ssum
        if g_weights is None:
            return w.astype(self.dtype)
        g = self.r(g_weights).astype(np.float64)
        g_u = (g - (g * w).sum(axis=1, keepdims=True)) / ssum
        g_e, g_sk = g_u * sk, g_u * e
        g_d = g_e * e * (-1.0 / (2 * r[idx] ** 2))
        g_r_pk = g_e * e * (d / r[idx] ** 3)
        g_radius, g_weight = np.zeros(M), np.zeros(M)
        np.add.at(g_radius, idx.reshape(-1), g_r_pk.reshape(-1))
        np.add.at(g_weight, idx.reshape(-1), g_sk.reshape(-1))
        return w.astype(self.dtype), dict(g_dist=g_d.astype(self.dtype), g_radius=g_radius.astype(self.dtype),
                                          g_weight=g_weight.astype(self.dtype) if kernel_weight is not None else None)

    def lbs_weights_dist(self, nn_dist, temperature=1.0, g_weights=None):
        """the `dist` branch (networks/sk_gs.py:769-770): ``w = softmax_k(-d / temperature)``; with ``g_weights`` also
        g_dist [P,K]"""
        d = self.r(nn_dist).astype(np.float64)
        l = -d / temperature
        e = np.exp(l - l.max(axis=1, keepdims=True))
        w = e / e.sum(axis=1, keepdims=True)
        if g_weights is None:
            return w.astype(self.dtype)
        g = self.r(g_weights).astype(np.float64)
        g_l = w * (g - (g * w).sum(axis=1, keepdims=True))
        return w.astype(self.dtype), (-g_l / temperature).astype(self.dtype)

    def knn_dist_backward(self, points, joints, indices, g_dist):
        """gradient of the squared distances d[n,k] = |p_n - j_idx[n,k]|^2 (pytorch3d.knn_points backward, sk_gs.py:757)
        w.r.t. points [P,dim] and joints [M,dim]"""
        p, j, idx = self.r(points).astype(np.float64), self.r(joints).astype(np.float64), np.asarray(indices)
        diff = p[:, None, :] - j[idx]
        gd = self.r(g_dist).astype(np.float64)[..., None] * 2.0 * diff
        g_j = np.zeros_like(j)
        np.add.at(g_j, idx.reshape(-1), -gd.reshape(-1, j.shape[1]))
        return gd.sum(axis=1).astype(self.dtype), g_j.astype(self.dtype)

    def lbs_weights(self, sp_W, indices):
        """``torch.gather(sp_W, 1, indices).softmax(-1)`` (networks/sk_gs.py:769-770) in numpy"""
        l = np.take_along_axis(self.r(sp_W), np.asarray(indices), axis=1)
        e = np.exp(l - l.max(axis=1, keepdims=True))
        return (e / e.sum(axis=1, keepdims=True)).astype(self.dtype)
