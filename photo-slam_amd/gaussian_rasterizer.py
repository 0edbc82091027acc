"""GaussianRasterizationSettings / GaussianRasterizerFunction / GaussianRasterizer.

Mirror of include/gaussian_rasterizer.h:25-127 and src/gaussian_rasterizer.cpp:18-234: same
member names, argument order, saved tensors, gradient order and std::runtime_error texts.
"""
from dataclasses import dataclass

import torch

from . import rasterize_points as rp


@dataclass
class GaussianRasterizationSettings:
    """include/gaussian_rasterizer.h:25-55"""
    image_height_: int
    image_width_: int
    tanfovx_: float
    tanfovy_: float
    bg_: torch.Tensor
    scale_modifier_: float
    viewmatrix_: torch.Tensor
    projmatrix_: torch.Tensor
    sh_degree_: int
    campos_: torch.Tensor
    prefiltered_: bool
    raw_params_: int = 0   # extension: GSR_RAW_* mask, activations fused into the rasterizer (include/gsr.h)
    # extension: a [P,3] tensor that receives the clamp-masked colour gradient in backward; the SH gradient is then left
    # to the view-factored exchange (trainer.ViewFactoredExchange) and autograd gets None for sh
    sh_grad_view_: torch.Tensor = None
    # extension, optimizer-in-backward for the SH tensor: dict(exp_avg, exp_avg_sq, lr, lr_tail, beta1, beta2, eps, step) --
    # backward applies this Adam step to sh in place instead of returning its gradient (gsr_backward_args.sh_adam)
    sh_adam_: dict = None
    # extension: (xyz_gradient_accum, denom, max_radii2D) -- backward adds this view's densification statistics itself
    view_stats_: tuple = None
    # extension, optimizer-in-backward for xyz / opacity / scaling / rotation: dict(tensors=[(param, exp_avg, exp_avg_sq, lr,
    # step)] x 4, beta1, beta2, eps) -- backward applies their Adam steps in place (gsr_backward_args.geom_adam) and autograd
    # gets None for the four; training_outputs_only_: the viewspace gradient and dL_dcov3D are not written either
    geom_adam_: dict = None
    training_outputs_only_: bool = False
    # extension (GSR_CULL_EMPTY_TILES, include/gsr.h): instances of tiles in which no pixel can blend the Gaussian are dropped
    # in front of the tile sort -- same image, same gradients, shorter internal lists
    cull_empty_tiles_: bool = False
    # extension: rasterize_points.RasterWorkspace -- the caller's persistent scratch buffers (None = fresh buffers per call, as the
    # reference)
    workspace_: object = None


class GaussianRasterizerFunction(torch.autograd.Function):
    """src/gaussian_rasterizer.cpp:28-180"""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
        s = raster_settings
        num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer = rp.RasterizeGaussiansCUDA(
            s.bg_, means3D, colors_precomp, opacities, scales, rotations, s.scale_modifier_, cov3Ds_precomp,
            s.viewmatrix_, s.projmatrix_, s.tanfovx_, s.tanfovy_, s.image_height_, s.image_width_, sh, s.sh_degree_,
            s.campos_, s.prefiltered_, s.raw_params_ | (8 if s.cull_empty_tiles_ else 0), s.sh_adam_,   # sh_adam_: lazy mode brings visible rows up to date first
            s.workspace_)
        ctx.set_materialize_grads(False)   # (no zero tensor for the unused gradient of `radii`)
        ctx.num_rendered = num_rendered
        ctx.raster_settings = s
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer,
                              binningBuffer, imgBuffer)
        ctx.mark_non_differentiable(radii)
        return color, radii

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii):
        if grad_out_color is None:   # the image took no part in the loss (set_materialize_grads(False)): no gradients
            return (None,) * 9
        s = ctx.raster_settings
        colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer, imgBuffer = \
            ctx.saved_tensors
        (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales,
         dL_drotations) = rp.RasterizeGaussiansBackwardCUDA(
            s.bg_, means3D, radii, colors_precomp, scales, rotations, s.scale_modifier_, cov3Ds_precomp, s.viewmatrix_,
            s.projmatrix_, s.tanfovx_, s.tanfovy_, grad_out_color, sh, s.sh_degree_, s.campos_, geomBuffer,
            ctx.num_rendered, binningBuffer, imgBuffer, s.raw_params_, s.sh_grad_view_,
            # view-factored mode: the SH step follows the exchange (gsr_sh_adam_from_views); sh_adam_ -- its lazy form -- served
            # the forward pass (rows this view sees caught up) and lets backward run this step's slice of the rotating
            # catch-up next to the blend kernel
            s.sh_adam_ if (s.sh_grad_view_ is None or (s.sh_adam_ or {}).get("row_step") is not None) else None, s.view_stats_,
            s.geom_adam_, s.training_outputs_only_)
        # order of src/gaussian_rasterizer.cpp:159-179
        def g(t, like):   # (None where an extension took the gradient's place)
            return t if like.numel() and t is not None else None
        return (dL_dmeans3D, dL_dmeans2D, g(dL_dsh, sh), g(dL_dcolors, colors_precomp), dL_dopacity,
                g(dL_dscales, scales), g(dL_drotations, rotations), g(dL_dcov3D, cov3Ds_precomp), None)


def rasterizeGaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
    """include/gaussian_rasterizer.h:78-100"""
    return GaussianRasterizerFunction.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                            cov3Ds_precomp, raster_settings)


class GaussianRasterizer(torch.nn.Module):
    """include/gaussian_rasterizer.h:102-127, src/gaussian_rasterizer.cpp:18-26,182-234"""

    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings_ = raster_settings

    def markVisibleGaussians(self, positions):
        with torch.no_grad():
            s = self.raster_settings_
            return rp.markVisible(positions, s.viewmatrix_, s.projmatrix_)

    def forward(self, means3D, means2D, opacities, has_shs, has_colors_precomp, has_scales, has_rotations,
                has_cov3D_precomp, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None):
        if (not has_shs and not has_colors_precomp) or (has_shs and has_colors_precomp):
            raise RuntimeError("Please provide excatly one of either SHs or precomputed colors!")
        if ((not has_scales or not has_rotations) and not has_cov3D_precomp) or \
                ((has_scales or has_rotations) and has_cov3D_precomp):
            raise RuntimeError("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        empty = torch.empty(0, device=means3D.device)
        shs = shs if has_shs else empty
        colors_precomp = colors_precomp if has_colors_precomp else empty
        scales = scales if has_scales else empty
        rotations = rotations if has_rotations else empty
        cov3D_precomp = cov3D_precomp if has_cov3D_precomp else empty
        color, radii = rasterizeGaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                          cov3D_precomp, self.raster_settings_)
        return color, radii
