"""Times GaussianModel.densifyAndPrune on the GPU (C3 by default) and lists its most expensive ops."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as entry
entry.load_package()
from photo_slam_amd import scene
from photo_slam_amd.gaussian_model import GaussianModel, GaussianOptimizationParams
from photo_slam_amd.gaussian_renderer import GaussianKeyframe, GaussianPipelineParams
from photo_slam_amd.trainer import TrainStep

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
dev = torch.device("cuda", 0)
cl = scene.make_config(cfg, seed=0)
cam = cl.cameras[0]
g = GaussianModel.from_cloud(cl, device=dev)
opt = GaussianOptimizationParams()
g.trainingSetup(opt)
kf = GaussianKeyframe.from_camera(cam, dev)
bg = torch.zeros(3, device=dev)
gt = torch.rand(3, cam.H, cam.W, device=dev)
mask = torch.ones(3, cam.H, cam.W, device=dev)
ts = TrainStep(g, opt, GaussianPipelineParams(), bg, cameras_extent=cl.extent)
for _ in range(5):
    ts.trainForOneIteration(kf, gt, mask)
for it in range(4):
    for _ in range(3):
        ts.trainForOneIteration(kf, gt, mask)
    torch.cuda.synchronize(); t0 = time.time()
    info = g.densifyAndPrune(opt.densify_grad_threshold_, 0.005, cl.extent, 0)
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"densifyAndPrune #{it}: {dt*1e3:.1f} ms  {info}")
from torch.profiler import profile, ProfilerActivity
for _ in range(3):
    ts.trainForOneIteration(kf, gt, mask)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    g.densifyAndPrune(opt.densify_grad_threshold_, 0.005, cl.extent, 0)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=60))
