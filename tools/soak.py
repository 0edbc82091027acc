"""Soak run on the GPU: N train iterations of the Python host with densification / pruning / opacity reset in the loop
(src/gaussian_mapper.cpp:614-774 order), checking that the loss stays finite and the model stays consistent."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as entry
entry.load_package()
from photo_slam_amd import scene
from photo_slam_amd.gaussian_model import GaussianModel, GaussianOptimizationParams
from photo_slam_amd.gaussian_renderer import GaussianKeyframe, GaussianPipelineParams, GaussianRenderer
from photo_slam_amd.trainer import TrainStep

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
# third argument: lazy SH Adam window (32 = default of the hosts, 0 = eager); fourth: 0 = the four geometry tensors step in
# separate passes
window = int(sys.argv[3]) if len(sys.argv) > 3 else 32
fused_geom = (int(sys.argv[4]) if len(sys.argv) > 4 else 1) != 0
dev = torch.device("cuda", 0)
cl = scene.make_config(cfg, seed=0, n_views=4)
g = GaussianModel.from_cloud(cl, device=dev)
opt = GaussianOptimizationParams()
opt.densification_interval_, opt.densify_from_iter_, opt.opacity_reset_interval_ = 100, 100, 500
g.trainingSetup(opt)
bg = torch.zeros(3, device=dev)
kfs = [GaussianKeyframe.from_camera(c, dev) for c in cl.cameras]
# ground truth = renders of a perturbed copy of the scene, so that there is something to learn
with torch.no_grad():
    gts = [GaussianRenderer.render(k, k.image_height_, k.image_width_, g, GaussianPipelineParams(), bg)[0].clone() for k in kfs]
    g.xyz_.add_(0.01 * torch.randn_like(g.xyz_))
    g.features_.add_(0.05 * torch.randn_like(g.features_))
mask = torch.ones_like(gts[0])
torch.manual_seed(0)
ts = TrainStep(g, opt, GaussianPipelineParams(), bg, cameras_extent=cl.extent, densify=True, lazy_sh_adam_window=window,
               fused_geom_adam=fused_geom)
t0 = time.time(); losses = []
for it in range(iters):
    k = it % len(kfs)
    loss = ts.trainForOneIteration(kfs[k], gts[k], mask)
    if it % 100 == 0 or it == iters - 1:
        l = float(loss)
        losses.append(l)
        assert l == l and abs(l) < 1e3, f"loss {l} at iteration {it}"
        P = g.xyz_.shape[0]
        for name in ("features_", "opacity_", "scaling_", "rotation_"):
            assert getattr(g, name).shape[0] == P, name
        assert torch.isfinite(g.xyz_).all() and torch.isfinite(g.features_).all()
        print(f"it {it:5d} loss {l:.5f} P {P}")
torch.cuda.synchronize()
print(f"soak ok (lazy window {window}, fused geometry step {fused_geom}): {iters} iterations in {time.time()-t0:.1f} s, "
      f"loss {losses[0]:.5f} -> {losses[-1]:.5f}, P {g.xyz_.shape[0]}, |features| {float(g.features_.abs().mean()):.6f}")
