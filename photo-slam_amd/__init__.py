"""photo-slam_amd -- MI355X-native Gaussian-splatting rasterizer hot path behind Photo-SLAM's API.

Directory name has a hyphen (repo convention), so the package is loaded under the module name
`photo_slam_amd` by __graft_entry__.load_package().  Contents:

  csrc/                 hand-written HIP (gfx950) kernels + the C-ABI (include/gsr.h)
  capi.py               ctypes binding of the C-ABI (no CPU fallback)
  rasterize_points.py   RasterizeGaussiansCUDA / ...BackwardCUDA / markVisible / distCUDA2
  gaussian_rasterizer.py GaussianRasterizationSettings / GaussianRasterizerFunction / GaussianRasterizer
  gaussian_renderer.py  GaussianRenderer.render
  gaussian_model.py / loss_utils.py / trainer.py   the measured train step (trainForOneIteration)
  scene.py              seeded synthetic clouds + reference camera conventions
"""
from . import capi, scene  # noqa: F401
