"""GaussianRenderer::render (include/gaussian_renderer.h:29-42, src/gaussian_renderer.cpp:23-149)."""
import os
from dataclasses import dataclass

import torch

from .gaussian_rasterizer import GaussianRasterizationSettings, GaussianRasterizer


@dataclass
class GaussianPipelineParams:
    """include/gaussian_parameters.h (pipeline flags)"""
    convert_SHs_: bool = False
    compute_cov3D_: bool = False


@dataclass
class GaussianKeyframe:
    """The per-view tensors render() consumes (include/gaussian_keyframe.h): transposed view
    matrix, transposed full projection, camera centre, FoV -- see scene.make_camera."""
    image_height_: int
    image_width_: int
    tanfovx_: float   # the reference stores FoVx_/FoVy_ and takes tan(FoV/2) in render()
    tanfovy_: float
    world_view_transform_: torch.Tensor
    full_proj_transform_: torch.Tensor
    camera_center_: torch.Tensor

    @classmethod
    def from_camera(cls, cam, device):
        t = lambda a: torch.from_numpy(a).to(device).contiguous()
        return cls(cam.H, cam.W, cam.tanfovx, cam.tanfovy, t(cam.viewmatrix), t(cam.projmatrix), t(cam.campos))


class GaussianRenderer:
    @staticmethod
    def render(viewpoint_camera, image_height, image_width, pc, pipe, bg_color, override_color=None,
               scaling_modifier=1.0, use_override_color=False, fuse_activations=True, sh_grad_view=None, sh_adam=None, view_stats=None,
               geom_adam=None, training_outputs_only=False, cull_empty_tiles=False, workspace=None):
        """returns (render, viewspace_points, visibility_filter, radii)

        fuse_activations (extension; False = the reference data flow): hand the raw opacity / scaling / rotation
        leaves to the rasterizer, which applies sigmoid / exp / normalize in preprocess and their chain rule in the
        backward preprocess -- same result, ~13 fewer elementwise launches and 3 fewer [P,*] temporaries per step.

        sh_grad_view, sh_adam, view_stats, geom_adam (extensions; None = the reference data flow): see
        GaussianRasterizationSettings.  training_outputs_only: the viewspace gradient and dL_dcov3D are not computed (for a caller
        whose densification statistics are fused: view_stats); implied by geom_adam.  cull_empty_tiles: instances of tiles in which
        no pixel can blend the Gaussian leave the list (same image and gradients; off by default -- measured a wash, DESIGN.md
        section 10); the environment variable GSR_CULL_EMPTY_TILES=0/1, when set, overrides the argument (an A/B handle)."""
        env = os.environ.get("GSR_CULL_EMPTY_TILES")
        if env:
            cull_empty_tiles = env == "1"
        # (with the fused geometry step nobody reads its gradient and the rasterizer never reads its values: no zero fill then)
        slim = geom_adam is not None or training_outputs_only
        screenspace_points = (torch.empty_like if slim else torch.zeros_like)(pc.getXYZ(), requires_grad=True)
        try:
            screenspace_points.retain_grad()
        except Exception:
            pass
        # SH evaluated in torch (convert_SHs_) or colours given: the rasterizer sees no SH tensor, so the SH extensions are off
        sh_in_rasterizer = not use_override_color and not pipe.convert_SHs_
        raw = 7 if fuse_activations and not pipe.compute_cov3D_ else 0
        raster_settings = GaussianRasterizationSettings(
            image_height, image_width, viewpoint_camera.tanfovx_, viewpoint_camera.tanfovy_, bg_color, scaling_modifier,
            viewpoint_camera.world_view_transform_, viewpoint_camera.full_proj_transform_, pc.active_sh_degree_,
            viewpoint_camera.camera_center_, False, raw,
            sh_grad_view if sh_in_rasterizer else None, sh_adam if sh_in_rasterizer else None, view_stats,
            geom_adam if raw == 7 else None, bool((geom_adam is not None or training_outputs_only) and raw == 7),
            cull_empty_tiles_=bool(cull_empty_tiles), workspace_=workspace)
        rasterizer = GaussianRasterizer(raster_settings)
        means3D = pc.getXYZ()
        means2D = screenspace_points
        opacity = pc.opacity_ if raw else pc.getOpacityActivation()
        scales = rotations = cov3D_precomp = None
        if pipe.compute_cov3D_:                                        # src/gaussian_renderer.cpp:78-86
            cov3D_precomp = pc.getCovarianceActivation()
        else:
            scales = pc.scaling_ if raw else pc.getScalingActivation()
            rotations = pc.rotation_ if raw else pc.getRotationActivation()
        has_shs = has_color_precomp = False
        shs = colors_precomp = None
        if use_override_color:
            colors_precomp, has_color_precomp = override_color, True
        elif pipe.convert_SHs_:                                        # :106-113: SH -> RGB in torch
            from . import sh_utils
            K = (pc.max_sh_degree_ + 1) ** 2
            shs_view = pc.getFeatures().transpose(1, 2).reshape(-1, 3, K)
            dir_pp = pc.getXYZ() - viewpoint_camera.camera_center_.reshape(1, 3)
            dir_pp_normalized = dir_pp / torch.norm(dir_pp, dim=1, keepdim=True)
            sh2rgb = sh_utils.eval_sh(pc.active_sh_degree_, shs_view, dir_pp_normalized)
            colors_precomp, has_color_precomp = torch.clamp_min(sh2rgb + 0.5, 0.0), True
        else:
            shs, has_shs = pc.getFeatures(), True
        has_sr = not pipe.compute_cov3D_
        rendered_image, radii = rasterizer(means3D, means2D, opacity, has_shs, has_color_precomp, has_sr, has_sr,
                                           pipe.compute_cov3D_, shs, colors_precomp, scales, rotations, cov3D_precomp)
        # (visibility_filter is one more launch: with the fused geometry step its consumers are fused too -- None then)
        return rendered_image, screenspace_points, (None if slim else radii > 0), radii
