"""GaussianRenderer::render (include/gaussian_renderer.h:29-42, src/gaussian_renderer.cpp:23-149)."""
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
               scaling_modifier=1.0, use_override_color=False, fuse_activations=True, sh_grad_view=None, sh_adam=None, view_stats=None):
        """returns (render, viewspace_points, visibility_filter, radii)

        fuse_activations (extension; False = the reference data flow): hand the raw opacity / scaling / rotation
        leaves to the rasterizer, which applies sigmoid / exp / normalize in preprocess and their chain rule in the
        backward preprocess -- same result, ~13 fewer elementwise launches and 3 fewer [P,*] temporaries per step.

        sh_grad_view, sh_adam, view_stats (extensions; None = the reference data flow): see GaussianRasterizationSettings."""
        screenspace_points = torch.zeros_like(pc.getXYZ(), requires_grad=True)
        try:
            screenspace_points.retain_grad()
        except Exception:
            pass
        raster_settings = GaussianRasterizationSettings(
            image_height, image_width, viewpoint_camera.tanfovx_, viewpoint_camera.tanfovy_, bg_color, scaling_modifier,
            viewpoint_camera.world_view_transform_, viewpoint_camera.full_proj_transform_, pc.active_sh_degree_,
            viewpoint_camera.camera_center_, False, 7 if fuse_activations else 0,
            None if use_override_color else sh_grad_view, None if use_override_color else sh_adam, view_stats)
        rasterizer = GaussianRasterizer(raster_settings)
        means3D = pc.getXYZ()
        means2D = screenspace_points
        if pipe.compute_cov3D_:
            raise NotImplementedError("compute_cov3D: pass cov3D_precomp to GaussianRasterizer.forward directly")
        if fuse_activations:
            opacity, scales, rotations = pc.opacity_, pc.scaling_, pc.rotation_
        else:
            opacity = pc.getOpacityActivation()
            scales = pc.getScalingActivation()
            rotations = pc.getRotationActivation()
        has_shs = has_color_precomp = False
        shs = colors_precomp = None
        if use_override_color:
            colors_precomp, has_color_precomp = override_color, True
        else:
            shs, has_shs = pc.getFeatures(), True
        rendered_image, radii = rasterizer(means3D, means2D, opacity, has_shs, has_color_precomp, True, True, False,
                                           shs, colors_precomp, scales, rotations, None)
        return rendered_image, screenspace_points, radii > 0, radii
