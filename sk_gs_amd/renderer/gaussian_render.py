"""Operator surface of the in-tree rasterizer, call-compatible with the reference module
``networks/renderer/gaussian_render.py`` (names, field order and defaults, ``.apply`` argument order, returned
tuples / dict keys, the ``means2D`` retain-grad trick, empty-tensor conventions, ``debug`` snapshots), backed by the
gfx950 kernels of ``libskgs_hip.so`` through :mod:`sk_gs_amd._C`.

Reference anchors: ``RasterizeBuffer`` :20-31, ``GaussianRasterizationSettings`` :34-48, ``_RasterizeGaussians`` :51-188,
``rasterize_gaussians`` :191-219, ``GaussianRasterizer`` :222-282, ``render`` :285-340, ``topk_weights`` :343-347.
"""
from typing import NamedTuple, Tuple

import torch
from torch import Tensor, nn
from torch.amp import custom_bwd, custom_fwd

from sk_gs_amd._C import get_C_function


def cpu_deep_copy_tuple(input_tuple):
    return tuple(x.cpu().clone() if isinstance(x, torch.Tensor) else x for x in input_tuple)


class RasterizeBuffer(NamedTuple):
    W: int
    """output image width"""
    H: int
    """output image height"""
    P: int
    """number of gaussians"""
    R: int
    """number of rendered tile instances (-1 when the forward ran without a host sync)"""
    geomBuffer: Tensor
    binningBuffer: Tensor
    imgBuffer: Tensor


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    detach_other_extra: bool = False
    """keep the gradients of the other extras away from means2D / conic / opacity"""
    colmap: bool = False


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    @custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, extras,
                raster_settings: GaussianRasterizationSettings, *other_extras):
        s = raster_settings
        args = (
            s.image_height, s.image_width, s.tanfovx, s.tanfovy, s.sh_degree, s.scale_modifier, s.prefiltered, s.debug,
            s.colmap, s.viewmatrix, s.projmatrix, s.campos,
            means3D, opacities, sh, scales, rotations, extras, colors_precomp, cov3Ds_precomp,
        )
        fwd = get_C_function('rasterize_gaussians')
        if s.debug:
            cpu_args = cpu_deep_copy_tuple(args)  # before anything can corrupt them
            try:
                num_rendered, color, opacity, radii, geomBuffer, binningBuffer, imgBuffer, out_extra = fwd(*args)
            except Exception:
                torch.save(cpu_args, 'snapshot_fw.dump')
                print('\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.')
                raise
        else:
            num_rendered, color, opacity, radii, geomBuffer, binningBuffer, imgBuffer, out_extra = fwd(*args)

        buffer = RasterizeBuffer(s.image_width, s.image_height, len(opacities), num_rendered, geomBuffer, binningBuffer,
                                 imgBuffer)
        extra_fwd = get_C_function('gaussian_rasterize_extra_forward') if other_extras else None
        pixel_extras = [extra_fwd(buffer.W, buffer.H, buffer.R, e, geomBuffer, binningBuffer, imgBuffer)
                        for e in other_extras]
        ctx.raster_settings = s
        ctx.num_rendered = num_rendered
        ctx.has_extras = extras is not None
        ctx.save_for_backward(colors_precomp, cov3Ds_precomp, means3D, scales, rotations, sh, extras,
                              geomBuffer, binningBuffer, imgBuffer, radii, opacity, *other_extras)
        ctx.mark_non_differentiable(radii)
        return (color, opacity, out_extra, radii, buffer, *pixel_extras)

    @staticmethod
    @custom_bwd(device_type='cuda')
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out_color, grad_out_opacity, grad_out_extra, *grad_outputs):
        s = ctx.raster_settings
        num_rendered = ctx.num_rendered
        colors_precomp, cov3Ds_precomp, means3D, scales, rotations, sh, extras = ctx.saved_tensors[:7]
        geomBuffer, binningBuffer, imgBuffer = ctx.saved_tensors[7:10]
        radii, opacity = ctx.saved_tensors[10:12]
        other_extras = ctx.saved_tensors[12:]

        if grad_out_color is None:
            grad_out_color = torch.zeros((3, s.image_height, s.image_width), device=means3D.device)
        if grad_out_opacity is None:
            grad_out_opacity = torch.zeros((s.image_height, s.image_width), device=means3D.device)

        grad_means2D = grad_conic = grad_opacity = None
        grad_extras = []
        grad_outputs = grad_outputs[2:]  # skip (radii, buffer)
        assert len(grad_outputs) == len(other_extras)
        for i, extra_i in enumerate(other_extras):
            if grad_outputs[i] is None:
                grad_extras.append(None)
                continue
            grad_extra_i, grad_means2D, grad_conic, grad_opacity = get_C_function('gaussian_rasterize_extra_backward')(
                s.image_width, s.image_height, num_rendered, extra_i, opacity, grad_outputs[i],
                geomBuffer, binningBuffer, imgBuffer, grad_means2D, grad_conic, grad_opacity)
            grad_extras.append(grad_extra_i)
        if s.detach_other_extra:
            grad_means2D = grad_conic = grad_opacity = None
        args = (
            s.scale_modifier, s.tanfovx, s.tanfovy, s.sh_degree, s.debug, s.colmap,
            s.viewmatrix, s.projmatrix, s.campos,
            means3D, colors_precomp, extras if ctx.has_extras else None, scales, rotations, cov3Ds_precomp, sh,
            num_rendered, radii, opacity,
            grad_out_color, grad_out_opacity, grad_out_extra,
            grad_means2D, grad_conic, grad_opacity,
            geomBuffer, binningBuffer, imgBuffer,
        )
        bwd = get_C_function('rasterize_gaussians_backward')
        if s.debug:
            cpu_args = cpu_deep_copy_tuple(args)
            try:
                out = bwd(*args)
            except Exception:
                torch.save(cpu_args, 'snapshot_bw.dump')
                print('\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n')
                raise
        else:
            out = bwd(*args)
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations, grad_extra) = out
        return (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_opacities, grad_scales, grad_rotations,
                grad_cov3Ds_precomp, grad_extra, None, *grad_extras)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, extras,
                        raster_settings: GaussianRasterizationSettings, **kwargs):
    outputs = _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                        cov3Ds_precomp, extras, raster_settings, *kwargs.values())
    color, opacity, out_extra, radii, buffer = outputs[:5]
    output_extras = {k: v for k, v in zip(kwargs.keys(), outputs[5:])}
    return color, opacity, out_extra, radii, buffer, output_extras


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """boolean mask of the points in front of the near plane"""
        with torch.no_grad():
            s = self.raster_settings
            return get_C_function('mark_visible')(positions, s.viewmatrix, s.projmatrix, s.colmap)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, extras=None, **kwargs):
        if (shs is None) == (colors_precomp is None):
            raise Exception('colour input: pass either shs or colors_precomp, not both and not neither')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
                (scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('shape input: pass either (scales, rotations) or cov3D_precomp, not both and not neither')
        empty = torch.Tensor([])
        shs = empty if shs is None else shs
        colors_precomp = empty if colors_precomp is None else colors_precomp
        scales = empty if scales is None else scales
        rotations = empty if rotations is None else rotations
        cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   extras, self.raster_settings, **kwargs)


def render(points: Tensor, opacity: Tensor, raster_settings: GaussianRasterizationSettings, scales: Tensor = None,
           rotations: Tensor = None, covariance: Tensor = None, sh_features: Tensor = None, colors=None, extras=None,
           **kwargs):
    """Render the scene; same dict as the reference's ``render`` (gaussian_render.py:285-340)."""
    # zero tensor through which autograd hands back the gradient of the 2D (screen-space) means: a NON-LEAF of zeros with
    # retain_grad() (gaussian_render.py:304-308).  A view of a fresh leaf is such a tensor for one fill launch (the
    # reference's "+ 0" costs a second elementwise launch per render)
    screenspace_points = torch.zeros_like(points, requires_grad=True).view(points.shape)
    try:
        screenspace_points.retain_grad()
    except Exception:  # noqa
        pass
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)
    rendered_image, rendered_opacity, rendered_extra, radii, buffer, outputs_extras = rasterizer(
        means3D=points, means2D=screenspace_points, shs=sh_features, colors_precomp=colors, opacities=opacity,
        scales=scales, rotations=rotations, cov3D_precomp=covariance, extras=extras, **kwargs)
    return {
        'images': rendered_image,
        'opacity': rendered_opacity,
        'viewspace_points': screenspace_points,
        'visibility_filter': radii > 0,
        'radii': radii,
        'extras': rendered_extra,
        'buffer': buffer,
        **outputs_extras,
    }


def topk_weights(topk, buffer: RasterizeBuffer) -> Tuple[Tensor, Tensor]:
    """top-k (alpha * T) Gaussian ids and weights per pixel"""
    return get_C_function('gaussian_topk_weights')(topk, buffer.W, buffer.H, buffer.P, buffer.R, buffer.geomBuffer,
                                                   buffer.binningBuffer, buffer.imgBuffer)


def debug_backward(path='snapshot_bw.dump'):
    dump = list(torch.load(path, map_location='cuda'))
    dump[4] = True
    get_C_function('rasterize_gaussians_backward')(*dump)
    print('No Error')
