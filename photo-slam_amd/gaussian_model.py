"""GaussianModel: the six learnable tensors, their activations, the 6-group Adam and the
densification statistics -- the parts of src/gaussian_model.cpp the measured train step uses
(activations :48-101, trainingSetup :477-510, addDensificationStats :817-831,
exponLrFunc :1118-1131).  ATen ops on the HIP device; fused HIP versions are a "next" row
(SURVEY.md 8f)."""
import math
from dataclasses import dataclass

import torch

from . import capi
from . import rasterize_points as rp


class FusedAdam:
    """torch.optim.Adam semantics (the reference's torch::optim::Adam with eps 1e-15,
    src/gaussian_model.cpp:477-510) as one HIP streaming pass per tensor (csrc/train_ops.hip).
    A group may carry `period`/`split`/`lr_tail`: elements [split, period) of every row use lr_tail
    (features_dc | features_rest share one [P,16,3] buffer with lr and lr/20)."""

    def __init__(self, groups, betas=(0.9, 0.999), eps=1e-15):
        self.param_groups = groups
        self.betas, self.eps = betas, eps
        self.state = {}   # id(param) -> exp_avg, exp_avg_sq, step (per parameter, as torch::optim::AdamParamState)

    def step(self):
        self.begin_step()
        for i in range(len(self.param_groups)):
            self.step_group(i)

    def begin_step(self):
        """(kept for the per-group drivers: the step counters are per parameter and advance in step_group)"""

    def step_group(self, i):
        """This step's update of parameter group i alone: a data-parallel driver updates a tensor as soon as ITS gradient
        reduction has landed, while the larger reductions are still in flight.  A parameter without a gradient (the
        iteration that densified) is skipped and its step counter does not advance -- torch::optim::Adam semantics."""
        lib = rp._lib()
        grp = self.param_groups[i]
        with torch.no_grad():
            for p in grp["params"]:
                if p.grad is None:
                    continue
                st = self.state.get(id(p))
                if st is None:
                    st = self.state[id(p)] = dict(exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p), step=0)
                st["step"] += 1
                g = p.grad.contiguous()
                capi.check(lib, lib.gsr_adam_step(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(),
                                                  st["exp_avg_sq"].data_ptr(), p.numel(), float(grp["lr"]),
                                                  self.betas[0], self.betas[1], self.eps, st["step"],
                                                  int(grp.get("period", 0)), int(grp.get("split", 0)),
                                                  float(grp.get("lr_tail", grp["lr"])), rp._stream_ptr(p)),
                           "gsr_adam_step")

    def begin_fused_step(self, i):
        """Arguments of the fused update of single-tensor group i (GaussianRasterizationSettings.sh_adam_): advances the
        parameter's step counter -- the update itself happens inside the rasterizer's backward, and step_group(i) then finds
        no gradient and does nothing."""
        grp = self.param_groups[i]
        (p,) = grp["params"]
        st = self.state.get(id(p))
        if st is None:
            st = self.state[id(p)] = dict(exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p), step=0)
        st["step"] += 1
        return dict(exp_avg=st["exp_avg"], exp_avg_sq=st["exp_avg_sq"], lr=float(grp["lr"]),
                    lr_tail=float(grp.get("lr_tail", grp["lr"])), beta1=self.betas[0], beta2=self.betas[1], eps=self.eps,
                    step=st["step"])

    def replace_param(self, old, new, exp_avg=None, exp_avg_sq=None):
        """Swap a parameter tensor (densify / prune / opacity reset): the Adam moments are replaced by the given
        tensors, or by zeros; the step counter carries over (replaceTensorToOptimizer, src/gaussian_model.cpp:567-586)."""
        for grp in self.param_groups:
            grp["params"] = [new if p is old else p for p in grp["params"]]
        prev = self.state.pop(id(old), None)
        self.state[id(new)] = dict(exp_avg=torch.zeros_like(new) if exp_avg is None else exp_avg,
                                   exp_avg_sq=torch.zeros_like(new) if exp_avg_sq is None else exp_avg_sq,
                                   step=prev["step"] if prev else 0)

    def moments(self, p):
        st = self.state.get(id(p))
        if st is None:
            return torch.zeros_like(p), torch.zeros_like(p)
        return st["exp_avg"], st["exp_avg_sq"]

    def zero_grad(self, set_to_none=True):
        for grp in self.param_groups:
            for p in grp["params"]:
                if set_to_none:
                    p.grad = None
                elif p.grad is not None:
                    p.grad.zero_()


@dataclass
class GaussianOptimizationParams:
    """include/gaussian_parameters.h:61-96 defaults (cfg/gaussian_mapper/RGB-D/Replica/replica_rgbd.yaml)."""
    iterations_: int = 30000
    position_lr_init_: float = 0.00016
    position_lr_final_: float = 0.0000016
    position_lr_delay_mult_: float = 0.01
    position_lr_max_steps_: int = 30000
    feature_lr_: float = 0.0025
    opacity_lr_: float = 0.05
    scaling_lr_: float = 0.005
    rotation_lr_: float = 0.001
    percent_dense_: float = 0.01
    lambda_dssim_: float = 0.2
    densification_interval_: int = 100
    opacity_reset_interval_: int = 3000
    densify_from_iter_: int = 500
    densify_until_iter_: int = 15000
    densify_grad_threshold_: float = 0.0002


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


class GaussianModel:
    def __init__(self, sh_degree=3, device="cuda"):
        self.max_sh_degree_ = sh_degree
        self.active_sh_degree_ = 0
        self.device_ = torch.device(device)
        self.spatial_lr_scale_ = 1.0
        self.optimizer_ = None

    # ---- construction
    @classmethod
    def from_cloud(cls, cloud, device="cuda", sh_degree=3):
        """Load a scene.Cloud (raw parameters) -- stands in for createFromPcd/loadPly in benchmarks."""
        m = cls(sh_degree, device)
        t = lambda a: torch.from_numpy(a).to(m.device_).contiguous().requires_grad_(True)
        m.xyz_ = t(cloud.xyz)
        import numpy as _np
        m.features_ = t(_np.concatenate([cloud.features_dc, cloud.features_rest], axis=1))
        m.scaling_ = t(cloud.scaling)
        m.rotation_ = t(cloud.rotation)
        m.opacity_ = t(cloud.opacity)
        m.active_sh_degree_ = sh_degree
        m.spatial_lr_scale_ = cloud.extent
        P = m.xyz_.shape[0]
        m.max_radii2D_ = torch.zeros(P, device=m.device_)
        m.xyz_gradient_accum_ = torch.zeros((P, 1), device=m.device_)
        m.denom_ = torch.zeros((P, 1), device=m.device_)
        return m

    def createFromPcd(self, points, colors, spatial_lr_scale):
        """src/gaussian_model.cpp:114-191: scales from the simple-knn distance (distCUDA2)."""
        self.spatial_lr_scale_ = spatial_lr_scale
        pts = points.to(self.device_).float().contiguous()
        n = pts.shape[0]
        C0 = 0.28209479177387814
        fused_color = (colors.to(self.device_).float() - 0.5) / C0  # RGB2SH, include/sh_utils.h:138
        M = (self.max_sh_degree_ + 1) ** 2
        features = torch.zeros((n, 3, M), device=self.device_)
        features[:, :3, 0] = fused_color
        dist2 = torch.clamp_min(rp.distCUDA2(pts), 0.0000001)
        scales = torch.log(torch.sqrt(dist2))[:, None].repeat(1, 3)
        rots = torch.zeros((n, 4), device=self.device_)
        rots[:, 0] = 1
        opacities = inverse_sigmoid(0.1 * torch.ones((n, 1), device=self.device_))
        req = lambda a: a.contiguous().requires_grad_(True)
        self.xyz_ = req(pts)
        self.features_ = req(features.transpose(1, 2))
        self.scaling_ = req(scales)
        self.rotation_ = req(rots)
        self.opacity_ = req(opacities)
        self.max_radii2D_ = torch.zeros(n, device=self.device_)
        self.xyz_gradient_accum_ = torch.zeros((n, 1), device=self.device_)
        self.denom_ = torch.zeros((n, 1), device=self.device_)

    # ---- checkpoint interchange, src/gaussian_model.cpp:838-1047
    def savePly(self, path):
        from . import ply_io
        n = lambda t: t.detach().cpu().numpy()
        ply_io.save_ply(path, n(self.xyz_), n(self.features_), n(self.opacity_), n(self.scaling_), n(self.rotation_))

    @classmethod
    def loadPly(cls, path, device="cuda", sh_degree=3):
        from . import ply_io
        d = ply_io.load_ply(path, sh_degree)
        m = cls(sh_degree, device)
        t = lambda a: torch.from_numpy(a).to(m.device_).contiguous().requires_grad_(True)
        m.xyz_, m.features_, m.opacity_, m.scaling_, m.rotation_ = (t(d["xyz"]), t(d["features"]), t(d["opacity"]),
                                                                   t(d["scaling"]), t(d["rotation"]))
        m.active_sh_degree_ = sh_degree   # loadPly sets the active degree to the maximum (:953)
        P = m.xyz_.shape[0]
        m.max_radii2D_ = torch.zeros(P, device=m.device_)
        m.xyz_gradient_accum_ = torch.zeros((P, 1), device=m.device_)
        m.denom_ = torch.zeros((P, 1), device=m.device_)
        return m

    # ---- activations, src/gaussian_model.cpp:48-71
    def getScalingActivation(self):
        return torch.exp(self.scaling_)

    def getRotationActivation(self):
        return torch.nn.functional.normalize(self.rotation_)

    def getXYZ(self):
        return self.xyz_

    # The reference keeps features_dc [P,1,3] and features_rest [P,15,3] as two leaves and rebuilds
    # [P,16,3] with cat + 2 clones every step (src/gaussian_model.cpp:63-66: ~3 x 384 MB at 2 M Gaussians).
    # Here ONE [P,16,3] leaf is the storage; dc / rest are views (for PLY I/O) and the two learning
    # rates are applied per coefficient inside the fused Adam.
    @property
    def features_dc_(self):
        return self.features_[:, 0:1, :]

    @property
    def features_rest_(self):
        return self.features_[:, 1:, :]

    def getFeatures(self):
        return self.features_

    def getOpacityActivation(self):
        return torch.sigmoid(self.opacity_)

    def setShDegree(self, sh):
        self.active_sh_degree_ = min(max(sh, 0), self.max_sh_degree_)

    def params(self):
        return [self.xyz_, self.features_, self.opacity_, self.scaling_, self.rotation_]

    # ---- optimizer, src/gaussian_model.cpp:477-510
    def trainingSetup(self, opt):
        self.percent_dense_ = opt.percent_dense_
        self.opt_ = opt
        # GaussianOptimizationParams holds C++ floats and set_lr() widens them to double (:489-502): the same values here
        import numpy as _np
        f32 = lambda x: float(_np.float32(x))
        groups = [
            dict(params=[self.xyz_], lr=float(_np.float32(opt.position_lr_init_) * _np.float32(self.spatial_lr_scale_)), name="xyz"),
            dict(params=[self.features_], lr=f32(opt.feature_lr_), name="f_dc+f_rest", period=3 * (self.max_sh_degree_ + 1) ** 2,
                 split=3, lr_tail=f32(opt.feature_lr_) / 20.0),
            dict(params=[self.opacity_], lr=f32(opt.opacity_lr_), name="opacity"),
            dict(params=[self.scaling_], lr=f32(opt.scaling_lr_), name="scaling"),
            dict(params=[self.rotation_], lr=f32(opt.rotation_lr_), name="rotation"),
        ]
        self.optimizer_ = FusedAdam(groups, eps=1e-15)

    def exponLrFunc(self, step):
        """src/gaussian_model.cpp:1118-1131"""
        import numpy as _np
        f32 = _np.float32
        o = self.opt_
        lr_init = f32(o.position_lr_init_) * f32(self.spatial_lr_scale_)       # float arithmetic throughout, as the reference
        lr_final = f32(o.position_lr_final_) * f32(self.spatial_lr_scale_)
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        t = min(max(f32(step) / f32(o.position_lr_max_steps_), f32(0.0)), f32(1.0))   # lr_delay_steps_ == 0: delay_rate 1
        return float(_np.exp(_np.log(lr_init) * (f32(1) - t) + _np.log(lr_final) * t))

    def updateLearningRate(self, step):
        lr = self.exponLrFunc(step)
        self.optimizer_.param_groups[0]["lr"] = lr
        return lr

    # ---- densification statistics, src/gaussian_model.cpp:817-831
    def addDensificationStats(self, viewspace_point_tensor, update_filter):
        g = viewspace_point_tensor.grad
        self.xyz_gradient_accum_[update_filter] += torch.norm(g[update_filter][:, :2], dim=-1, keepdim=True)
        self.denom_[update_filter] += 1

    def addViewStats(self, viewspace_point_tensor, radii):
        """max_radii2D / xyz_gradient_accum / denom update of one view (gaussian_mapper.cpp:714-719) in one
        HIP pass (gsr_densify_stats) instead of boolean-mask gathers and scatters."""
        lib = rp._lib()
        g = viewspace_point_tensor.grad.contiguous()
        r = radii.contiguous()
        capi.check(lib, lib.gsr_densify_stats(self.xyz_.shape[0], g.data_ptr(), r.data_ptr(), self.xyz_gradient_accum_.data_ptr(),
                                              self.denom_.data_ptr(), self.max_radii2D_.data_ptr(), rp._stream_ptr(g)),
                   "gsr_densify_stats")


    # ------------------------------------------------------------------ densification (amortised 1/interval)
    # src/gaussian_model.cpp:588-815.  Same selection rules, same resulting order
    # [originals that were not split | clones | split children] and the same Adam-state surgery, but each
    # tensor is rebuilt ONCE (the reference copies every tensor 4-6 times through clone -> cat -> split -> cat
    # -> prune -> prune and then drops the allocator cache).
    _PARAM_NAMES = ("xyz_", "features_", "opacity_", "scaling_", "rotation_")

    def _rebuild(self, index, overrides=None):
        """Gather all parameters / Adam moments with `index` (long tensor into the current arrays; -1 = new
        entry whose moments are zero); overrides: {name: (positions, values)} applied after the gather."""
        new_rows = index < 0
        safe = index.clamp_min(0)
        for name in self._PARAM_NAMES:
            old = getattr(self, name)
            m, v = self.optimizer_.moments(old) if self.optimizer_ is not None else (None, None)
            with torch.no_grad():
                new = old.detach()[safe].clone()
                if overrides and name in overrides:
                    pos, val = overrides[name]
                    new[pos] = val
                new.requires_grad_(True)
                if m is not None:
                    m2, v2 = m[safe].clone(), v[safe].clone()
                    m2[new_rows] = 0
                    v2[new_rows] = 0
            setattr(self, name, new)
            if self.optimizer_ is not None:
                self.optimizer_.replace_param(old, new, m2, v2)
        n = index.shape[0]
        dev = self.xyz_.device
        return n, dev

    def prunePoints(self, mask):
        """:588-642"""
        keep = torch.nonzero(~mask).squeeze(1)
        self._rebuild(keep)
        self.xyz_gradient_accum_ = self.xyz_gradient_accum_[keep]
        self.denom_ = self.denom_[keep]
        self.max_radii2D_ = self.max_radii2D_[keep]

    def densifyAndPrune(self, max_grad, min_opacity, extent, max_screen_size, generator=None, N=2):
        """:795-815 (densifyAndClone :763-793, densifyAndSplit :716-761, prunePoints :588-642) in one rebuild."""
        with torch.no_grad():
            grads = self.xyz_gradient_accum_ / self.denom_
            grads[grads.isnan()] = 0.0
            g = grads.squeeze(-1)
            scal = self.getScalingActivation()
            smax = scal.max(dim=1).values
            # the reference's thresholds are C++ float products / float arguments: evaluate them in fp32 too, or a Gaussian
            # sitting exactly on a threshold is classified differently
            import numpy as _np
            f32 = _np.float32
            max_grad, min_opacity = float(f32(max_grad)), float(f32(min_opacity))
            big = smax > float(f32(self.percent_dense_) * f32(extent))
            clone_mask = (g.abs() >= max_grad) & ~big          # frobenius_norm over the last dim of [P,1]
            split_mask = (g >= max_grad) & big
            P = self.xyz_.shape[0]
            ar = torch.arange(P, device=self.xyz_.device)
            keep_idx, clone_idx, split_idx = ar[~split_mask], ar[clone_mask], ar[split_mask]
            rep = split_idx.repeat(N)
            # children: position sampled from the parent Gaussian, scale / (0.8 N)
            stds = scal[rep]
            samples = torch.normal(torch.zeros_like(stds), stds, generator=generator)
            q = self.rotation_.detach()[rep]
            q = q / q.norm(dim=1, keepdim=True)
            r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
            R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                             2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                             2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
            child_xyz = torch.bmm(R, samples.unsqueeze(-1)).squeeze(-1) + self.xyz_.detach()[rep]
            child_scaling = torch.log(stds / (0.8 * N))
            # new entries are marked by (-1 - source) so their Adam moments come out zero
            index = torch.cat([keep_idx, clone_idx, rep])
            is_new = torch.cat([torch.zeros_like(keep_idx, dtype=torch.bool), torch.ones_like(clone_idx, dtype=torch.bool),
                                torch.ones_like(rep, dtype=torch.bool)])
            # final prune (:805-813) evaluated on the would-be tensors
            opac = torch.sigmoid(self.opacity_.detach()[index]).squeeze(-1)
            prune = opac < min_opacity
            if max_screen_size:
                # max_radii2D is reset by densificationPostfix before the prune, so big_points_vs is always false
                new_smax = torch.cat([smax[keep_idx], smax[clone_idx], torch.exp(child_scaling).max(dim=1).values])
                prune = prune | (new_smax > float(f32(0.1) * f32(extent)))   # 0.1f * extent
            sel = ~prune
            child_pos_all = torch.arange(keep_idx.shape[0] + clone_idx.shape[0], index.shape[0], device=index.device)
            new_pos = torch.cumsum(sel.to(torch.int64), 0) - 1
            child_sel = sel[child_pos_all]
            child_pos = new_pos[child_pos_all][child_sel]
            index_f = index[sel]
            is_new_f = is_new[sel]
        gather_index = torch.where(is_new_f, -1 - index_f, index_f)
        self._rebuild_with_sources(gather_index, {"xyz_": (child_pos, child_xyz[child_sel]),
                                                  "scaling_": (child_pos, child_scaling[child_sel])})
        n = gather_index.shape[0]
        dev = self.xyz_.device
        self.xyz_gradient_accum_ = torch.zeros((n, 1), device=dev)
        self.denom_ = torch.zeros((n, 1), device=dev)
        self.max_radii2D_ = torch.zeros(n, device=dev)
        return dict(cloned=int(clone_idx.shape[0]), split=int(split_idx.shape[0]), pruned=int(prune.sum()), points=n,
                    children_kept=int(child_sel.sum()))

    def _rebuild_with_sources(self, gather_index, overrides):
        """gather_index >= 0: existing row (keeps its Adam moments); < 0: copy of row (-1 - value) with zero moments."""
        new_rows = gather_index < 0
        src = torch.where(new_rows, -1 - gather_index, gather_index)
        for name in self._PARAM_NAMES:
            old = getattr(self, name)
            m, v = self.optimizer_.moments(old) if self.optimizer_ is not None else (None, None)
            with torch.no_grad():
                new = old.detach()[src].clone()
                if name in overrides:
                    pos, val = overrides[name]
                    new[pos] = val
                new.requires_grad_(True)
                if m is not None:
                    m2, v2 = m[src].clone(), v[src].clone()
                    m2[new_rows] = 0
                    v2[new_rows] = 0
            setattr(self, name, new)
            if self.optimizer_ is not None:
                self.optimizer_.replace_param(old, new, m2, v2)

    def resetOpacity(self, clamp_to=None):
        """src/gaussian_model.cpp:556-565 exactly as shipped: opacities_new = inverse_sigmoid(min(sigmoid(o), ones_like(sigmoid(o)
        * 0.01))) -- the 0.01 sits INSIDE ones_like, so the clamp is against 1 and never binds: the values survive (up to the
        sigmoid / logit round trip, which sends logits above ~17 to +inf exactly as the reference does) and only the Adam
        moments of the opacity group are zeroed (replaceTensorToOptimizer :567-586).  clamp_to=0.01 gives the reset 3DGS
        intended (opacity <- min(opacity, 0.01)), a deliberate deviation a caller has to ask for."""
        with torch.no_grad():
            act = self.getOpacityActivation()
            bound = torch.ones_like(act * 0.01) if clamp_to is None else torch.ones_like(act) * clamp_to
            new = inverse_sigmoid(torch.min(act, bound))
        new = new.clone().requires_grad_(True)
        old = self.opacity_
        self.opacity_ = new
        if self.optimizer_ is not None:
            self.optimizer_.replace_param(old, new)
