"""GaussianModel: the six learnable tensors, their activations, the 6-group Adam and the
densification statistics -- the parts of src/gaussian_model.cpp the measured train step uses
(activations :48-101, trainingSetup :477-510, addDensificationStats :817-831,
exponLrFunc :1118-1131).  ATen ops on the HIP device; fused HIP versions are a "next" row
(SURVEY.md 8f)."""
import ctypes as C
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
        # multiplies every learning rate where it is consumed: 1 = training, 0 = a STATIONARY workload for throughput
        # measurements (every kernel runs, the moments update, the parameters do not move; bench.py)
        self.lr_scale = 1.0
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
                self.sync_lazy(p)   # a dense step: every row must be up to date first
                st = self.state.get(id(p))
                if st is None:
                    st = self.state[id(p)] = dict(exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p), step=0)
                st["step"] += 1
                g = p.grad.contiguous()
                capi.check(lib, lib.gsr_adam_step(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(),
                                                  st["exp_avg_sq"].data_ptr(), p.numel(), float(grp["lr"]) * self.lr_scale,
                                                  self.betas[0], self.betas[1], self.eps, st["step"],
                                                  int(grp.get("period", 0)), int(grp.get("split", 0)),
                                                  float(grp.get("lr_tail", grp["lr"])) * self.lr_scale, rp._stream_ptr(p)),
                           "gsr_adam_step")

    def step_groups(self, indices, grad_scale=1.0):
        """step_group(i) for several single-tensor groups with uniform learning rates in ONE launch (gsr_adam_step_multi: the
        four small per-Gaussian tensors of a data-parallel step, whose gradients arrive together from one all-reduce).
        grad_scale: the gradients are multiplied by it as they are read (1/N of a batch mean whose all-reduce summed)."""
        entries = []
        with torch.no_grad():
            for i in indices:
                grp = self.param_groups[i]
                (p,) = grp["params"]
                if p.grad is None:
                    continue
                if grp.get("period", 0) or not p.is_contiguous() or not p.grad.is_contiguous():
                    if grad_scale != 1.0:
                        p.grad.mul_(grad_scale)
                    self.step_group(i)
                    continue
                st = self.state.get(id(p))
                if st is None:
                    st = self.state[id(p)] = dict(exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p), step=0)
                st["step"] += 1
                entries.append((p.detach(), p.grad, st["exp_avg"], st["exp_avg_sq"], float(grp["lr"]) * self.lr_scale, st["step"]))
            rp.adamStepMulti(entries, self.betas[0], self.betas[1], self.eps, grad_scale)

    def begin_fused_step(self, i, lazy_window=0):
        """Arguments of the fused update of single-tensor group i (GaussianRasterizationSettings.sh_adam_): advances the
        parameter's step counter -- the update itself happens inside the rasterizer's backward, and step_group(i) then finds
        no gradient and does nothing.
        lazy_window >= 2: the zero-gradient steps of the culled Gaussians' rows are taken lazily (gsr_sh_adam_lazy,
        include/gsr.h): the dict then carries row_step / window / the past learning rates, goes to the rasterizer's forward
        AND backward, and end_fused_step() must follow the backward pass; sync_lazy() brings every row up to date."""
        grp = self.param_groups[i]
        (p,) = grp["params"]
        st = self.state.get(id(p))
        if st is None:
            st = self.state[id(p)] = dict(exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p), step=0)
        lazy = lazy_window >= 2 and p.dim() == 3 and p.size(1) == 16 and p.is_contiguous()
        if not lazy or st.get("window", lazy_window) != lazy_window:
            self.sync_lazy(p)
        if lazy and "row_step" not in st:   # every row has taken the st["step"] steps so far
            st["row_step"] = torch.full((p.size(0),), st["step"], dtype=torch.int32, device=p.device)
            st["lr_hist"], st["window"] = [], int(lazy_window)
        st["step"] += 1
        d = dict(exp_avg=st["exp_avg"], exp_avg_sq=st["exp_avg_sq"], lr=float(grp["lr"]) * self.lr_scale,
                 lr_tail=float(grp.get("lr_tail", grp["lr"])) * self.lr_scale, beta1=self.betas[0], beta2=self.betas[1], eps=self.eps,
                 step=st["step"])
        if lazy:
            d.update(row_step=st["row_step"], window=st["window"], lr_past=[a for a, _ in st["lr_hist"]],
                     lr_tail_past=[b for _, b in st["lr_hist"]])
        return d

    def end_fused_step(self, i, d):
        """After the backward pass of a lazy fused step: its learning rates join the history the later catch-ups need."""
        if d is None or d.get("row_step") is None:
            return
        (p,) = self.param_groups[i]["params"]
        st = self.state[id(p)]
        st["lr_hist"].insert(0, (d["lr"], d["lr_tail"]))
        del st["lr_hist"][st["window"]:]

    def is_lazy(self, p):
        st = self.state.get(id(p))
        return st is not None and "row_step" in st

    def sync_lazy(self, p):
        """Lazy mode: every row of p takes the zero-gradient steps it is behind (gsr_sh_adam_flush: the same arithmetic,
        bit-identical to the eager update) and the lazy state is dropped.  No-op otherwise."""
        st = self.state.get(id(p))
        if st is None or "row_step" not in st:
            return
        row_step, hist, window = st.pop("row_step"), st.pop("lr_hist"), st.pop("window")
        if hist:
            rp.shAdamFlush(p.detach(), dict(exp_avg=st["exp_avg"], exp_avg_sq=st["exp_avg_sq"], lr=hist[0][0], lr_tail=hist[0][1],
                                            beta1=self.betas[0], beta2=self.betas[1], eps=self.eps, step=st["step"],
                                            row_step=row_step, window=window, lr_past=[a for a, _ in hist[1:]],
                                            lr_tail_past=[b for _, b in hist[1:]]))

    def replace_param(self, old, new, exp_avg=None, exp_avg_sq=None):
        """Swap a parameter tensor (densify / prune / opacity reset): the Adam moments are replaced by the given
        tensors, or by zeros; the step counter carries over (replaceTensorToOptimizer, src/gaussian_model.cpp:567-586)."""
        self.sync_lazy(old)
        for grp in self.param_groups:
            grp["params"] = [new if p is old else p for p in grp["params"]]
        prev = self.state.pop(id(old), None)
        self.state[id(new)] = dict(exp_avg=torch.zeros_like(new) if exp_avg is None else exp_avg,
                                   exp_avg_sq=torch.zeros_like(new) if exp_avg_sq is None else exp_avg_sq,
                                   step=prev["step"] if prev else 0)

    def moments(self, p):
        self.sync_lazy(p)
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
        self._features = None
        self._in_lazy_step = False   # set by the train step around its own render call
        # include/gaussian_model.h:169,182-183: the iteration each Gaussian has existed since (read by the loop-closure
        # transform, src/gaussian_model.cpp:433), and the sparse points handed to increasePcd so far (saveSparsePointsPly :1049)
        self.exist_since_iter_ = None
        self.sparse_points_xyz_ = None
        self.sparse_points_color_ = None

    # features_: the [P,16,3] SH leaf.  With lazy Adam steps for the rows of culled Gaussians (FusedAdam.begin_fused_step,
    # gsr_sh_adam_lazy) rows of it may be behind between train steps: every read from outside the fused step brings them up
    # to date first.
    @property
    def features_(self):
        if not self._in_lazy_step:
            self.sync_features()
        return self._features

    @features_.setter
    def features_(self, value):
        self._features = value

    def sync_features(self):
        if self.optimizer_ is not None and self._features is not None:
            self.optimizer_.sync_lazy(self._features)

    # ---- construction
    @classmethod
    def from_cloud(cls, cloud, device="cuda", sh_degree=3):
        """Load a scene.Cloud (raw parameters) -- stands in for createFromPcd/loadPly in benchmarks."""
        m = cls(sh_degree, device)
        # (a copy also on the host: the fused optimizer steps update the leaves in place, and from_numpy() alone would alias
        # the cloud's arrays there)
        t = lambda a: torch.from_numpy(a).to(m.device_, copy=True).contiguous().requires_grad_(True)
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
        m.exist_since_iter_ = torch.zeros(P, dtype=torch.int32, device=m.device_)
        return m

    def createFromPcd(self, points, colors, spatial_lr_scale):
        """src/gaussian_model.cpp:114-191: scales from the simple-knn distance (distCUDA2)."""
        self.spatial_lr_scale_ = spatial_lr_scale
        pts = points.to(self.device_).float().contiguous()
        n = pts.shape[0]
        # RGB2SH, include/sh_utils.h:138: (rgb - 0.5f) / C0 with the FLOAT constant C0 as a host scalar, exactly the reference's
        # expression: ATen divides on the host and multiplies by the reciprocal of THAT float on the GPU (the double
        # 0.28209479177387814 has another reciprocal in float: one ulp off in nearly every row, found on the GPU box)
        C0 = float(torch.tensor(0.28209479177387814, dtype=torch.float32))
        fused_color = (colors.to(self.device_).float() - 0.5) / C0
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
        self.exist_since_iter_ = torch.zeros(n, dtype=torch.int32, device=self.device_)   # :167-169

    def increasePcd(self, points, colors, iteration):
        """src/gaussian_model.cpp:193-376, both overloads (std::vector<float> x 2 / torch::Tensor& x 2): new SLAM map points
        join the model -- colours -> SH DC term, scales from the simple-knn distance AMONG THE NEW POINTS (distCUDA2 of the
        new cloud alone, :238,325), identity rotations, opacity 0.1, exist_since_iter = iteration -- through
        densificationPostfix (:644-712): appended behind the existing rows, zero Adam moments for the new rows, the step
        counters carried, and -- as the reference does -- all three statistics arrays reset to zero.
        Here the append is O(new points) while the arena has room (the reference re-cats every tensor and moment)."""
        dev = self.xyz_.device
        as_rows = lambda a: (a if torch.is_tensor(a) else torch.tensor(list(a), dtype=torch.float32)).to(dev, torch.float32).reshape(-1, 3)
        pts, cols = as_rows(points), as_rows(colors)
        assert pts.shape == cols.shape
        n = pts.shape[0]
        if n == 0:
            return 0
        with torch.no_grad():
            self.sparse_points_xyz_ = pts if self.sparse_points_xyz_ is None else torch.cat([self.sparse_points_xyz_, pts], 0)
            self.sparse_points_color_ = cols if self.sparse_points_color_ is None else torch.cat([self.sparse_points_color_, cols], 0)
            C0 = float(torch.tensor(0.28209479177387814, dtype=torch.float32))   # (the FLOAT constant: see createFromPcd)
            M = (self.max_sh_degree_ + 1) ** 2
            features = torch.zeros((n, M, 3), device=dev)
            features[:, 0, :] = (cols - 0.5) / C0                                   # RGB2SH, include/sh_utils.h:138
            dist2 = torch.clamp_min(rp.distCUDA2(pts.clone().contiguous()), 0.0000001)
            scales = torch.log(torch.sqrt(dist2))[:, None].repeat(1, 3)
            rots = torch.zeros((n, 4), device=dev)
            rots[:, 0] = 1
            opacities = inverse_sigmoid(0.1 * torch.ones((n, 1), device=dev))
            self._append_rows(dict(xyz_=pts, features_=features, opacity_=opacities, scaling_=scales, rotation_=rots), int(iteration))
        return n

    def _append_rows(self, new_rows, iteration):
        """densificationPostfix (:644-712) as an append into the arena of the rebuilds."""
        _ = self.features_           # lazy SH Adam: every row up to date before the tensor is re-seated
        P, n = self.xyz_.shape[0], new_rows["xyz_"].shape[0]
        dev = self.xyz_.device
        arena = getattr(self, "_arena", None)
        live = arena is not None and arena["capacity"] >= P + n and all(
            getattr(self, name).data_ptr() == arena["sets"][arena["cur"]]["params"][name][0].data_ptr() for name in self._PARAM_NAMES)
        if self.exist_since_iter_ is None:
            self.exist_since_iter_ = torch.zeros(P, dtype=torch.int32, device=dev)
        old_exist = self.exist_since_iter_
        if not live:
            self.reserve(int((P + n) * 1.25) + 64)
            arena = self._arena
        cur = arena["sets"][arena["cur"]]
        for name in self._PARAM_NAMES:
            old = getattr(self, name)
            have = self.optimizer_ is not None and id(old) in self.optimizer_.state
            m, v = self.optimizer_.moments(old) if have else (None, None)
            bufs = [b.narrow(0, 0, P + n) for b in cur["params"][name]]
            if not live:
                bufs[0][:P].copy_(old.detach())
                if have:
                    bufs[1][:P].copy_(m)
                    bufs[2][:P].copy_(v)
            bufs[0][P:].copy_(new_rows[name])
            bufs[1][P:].zero_()
            bufs[2][P:].zero_()
            if not have:
                bufs[1][:P].zero_()
                bufs[2][:P].zero_()
            new = bufs[0].detach().requires_grad_(True)
            setattr(self, name, new)
            if self.optimizer_ is not None:
                self.optimizer_.replace_param(old, new, bufs[1], bufs[2])
        exist = cur["exist"].narrow(0, 0, P + n)
        if not live or old_exist.data_ptr() != exist.data_ptr():
            exist[:P].copy_(old_exist)
        exist[P:].fill_(iteration)
        self.exist_since_iter_ = exist
        stats = [b.narrow(0, 0, P + n) for b in cur["stats"]]
        for t in stats:
            t.zero_()                                                            # :709-711
        self.xyz_gradient_accum_, self.denom_, self.max_radii2D_ = stats

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
        m.exist_since_iter_ = torch.zeros(P, dtype=torch.int32, device=m.device_)
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

    def getCovarianceActivation(self, scaling_modifier=1):
        """src/gaussian_model.cpp:73-96: Sigma = (R S)(R S)^T, R = build_rotation(rotation_) (the quaternion is normalised
        there, include/general_utils.h:33-57), S = diag(scaling_modifier * exp(scaling_)); entries xx xy xz yy yz zz."""
        q = self.rotation_ / torch.sqrt((self.rotation_ * self.rotation_).sum(1, keepdim=True))
        r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                         2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                         2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
        L = R * (scaling_modifier * self.getScalingActivation()).unsqueeze(1)
        cov = torch.bmm(L, L.transpose(1, 2))
        return torch.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], 1)

    def setShDegree(self, sh):
        self.active_sh_degree_ = min(max(sh, 0), self.max_sh_degree_)

    def params(self):
        return [self.xyz_, self.features_, self.opacity_, self.scaling_, self.rotation_]

    def params_raw(self):
        """The five leaves as they are (no catch-up of lazily stepped SH rows): for code that only needs the tensors' .grad or
        shape inside a train step."""
        return [self.xyz_, self._features, self.opacity_, self.scaling_, self.rotation_]

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
    # src/gaussian_model.cpp:588-815 as stream compaction (csrc/densify.hip, include/gsr.h): gsr_densify_select turns the
    # per-Gaussian decisions into a gather plan on the device, the host reads the counts ONCE (it has to size the new
    # tensors), gsr_densify_gather rebuilds the five parameter tensors, their ten Adam moments and the statistics in one
    # launch.  Same selection rules, same resulting order [originals that were not split | clones | first children | second
    # children] and the same Adam-state surgery as the reference, which copies every tensor 4-6 times through clone -> cat ->
    # split -> cat -> prune -> prune and then drops the allocator cache (:814).  Pinned to the reference's own functions by
    # tests/test_densify_reference.py.
    _PARAM_NAMES = ("xyz_", "features_", "opacity_", "scaling_", "rotation_")

    def reserve(self, capacity):
        """Arena for the rebuilds: two sets of [capacity, row] buffers for the five parameters, their moments and the three
        statistics arrays.  densifyAndPrune / prunePoints gather from the live tensors into the idle set and swap, so a
        growing map allocates nothing until it outgrows the capacity (then the arena grows by 1.5x).  Without this call the
        arena is created on the first rebuild with 25 % headroom."""
        rows = dict(xyz_=(3,), features_=tuple(self.features_.shape[1:]), opacity_=(1,), scaling_=(3,), rotation_=(4,))
        dev = self.xyz_.device
        mk = lambda shape: torch.empty((capacity,) + shape, device=dev, dtype=torch.float32)
        self._arena = dict(capacity=capacity, cur=0,
                           sets=[dict(params={n: [mk(r), mk(r), mk(r)] for n, r in rows.items()},
                                      stats=[mk((1,)), mk((1,)), mk(())],
                                      exist=torch.empty((capacity,), device=dev, dtype=torch.int32)) for _ in range(2)])

    def release_arena(self):
        """The live tensors move into allocations of their own and both arena sets are freed (GaussianModel::releaseArena of the
        C++ host; what the reference's emptyCache() after densification achieves, src/gaussian_model.cpp:814).  LIFETIME
        CONTRACT of the arena: after a rebuild xyz_, features_, ..., the Adam moments, the statistics and exist_since_iter_ are
        narrow() views into one of two persistent sets -- a tensor obtained BEFORE rebuild N is overwritten by rebuild N + 2;
        clone() what has to outlive a rebuild, or call this when densification ends."""
        with torch.no_grad():
            for name in self._PARAM_NAMES:
                old = getattr(self, name)      # (features_: brings lazily stepped rows up to date first)
                new = old.detach().clone().requires_grad_(True)
                setattr(self, name, new)
                if self.optimizer_ is not None:
                    m, v = self.optimizer_.moments(old)
                    self.optimizer_.replace_param(old, new, m.clone(), v.clone())
            self.xyz_gradient_accum_, self.denom_ = self.xyz_gradient_accum_.clone(), self.denom_.clone()
            self.max_radii2D_ = self.max_radii2D_.clone()
            if self.exist_since_iter_ is not None:
                self.exist_since_iter_ = self.exist_since_iter_.clone()
        self._arena = None
        self._densify_scratch = None

    def _compact(self, select, generator=None, morton_reindex=False):
        """select: fills a capi.DensifySelectArgs.  Returns the counts [kept, clones, child parents, split, clone-selected,
        rows of the new set]."""
        lib = rp._lib()
        P = self.xyz_.shape[0]
        dev = self.xyz_.device
        stream = rp._stream_ptr(self.xyz_)
        with torch.no_grad():
            a = capi.DensifySelectArgs()
            a.P = P
            keep = select(a)
            nbytes = lib.gsr_densify_scratch_bytes(P)
            scratch = getattr(self, "_densify_scratch", None)
            if scratch is None or scratch.numel() < nbytes or scratch.device != dev:
                scratch = self._densify_scratch = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=dev)
            counts = torch.empty(8, dtype=torch.int32, device=dev)
            capi.check(lib, lib.gsr_densify_select(C.byref(a), scratch.data_ptr(), counts.data_ptr(), stream), "gsr_densify_select")
            n_keep, n_clone, n_child, n_split, n_clone_sel, n_new = counts.tolist()[:6]   # the one host read of the rebuild
            del keep
            arena = getattr(self, "_arena", None)
            if arena is None or arena["capacity"] < n_new or arena["sets"][0]["params"]["xyz_"][0].device != dev:
                grow = max(n_new, P)
                self.reserve(int(grow * (1.25 if arena is None else 1.5)) + 64)
                arena = self._arena
            arena["cur"] ^= 1
            dst = arena["sets"][arena["cur"]]
            # at::normal(zeros, stds) of the reference (:731-734) is randn(2k,3) * stds: the same draws, the scale applied in-kernel
            samples = torch.empty((2 * n_split, 3), device=dev).normal_(generator=generator) if n_split else None
            g = capi.DensifyGatherArgs()
            g.P, g.n_new, g.n_keep, g.n_clone, g.n_child, g.n_split = P, n_new, n_keep, n_clone, n_child, n_split
            g.features_row_floats = int(self.features_.shape[1] * self.features_.shape[2])
            olds, news = [], []
            for i, name in enumerate(self._PARAM_NAMES):
                old = getattr(self, name)
                m, v = self.optimizer_.moments(old) if self.optimizer_ is not None and id(old) in self.optimizer_.state else (None, None)
                outs = [b.narrow(0, 0, n_new) for b in dst["params"][name]]
                src = old.detach()
                assert src.is_contiguous()
                g.param_in[i], g.param_out[i] = src.data_ptr(), outs[0].data_ptr()
                if m is not None:
                    g.exp_avg_in[i], g.exp_avg_sq_in[i] = m.data_ptr(), v.data_ptr()
                    g.exp_avg_out[i], g.exp_avg_sq_out[i] = outs[1].data_ptr(), outs[2].data_ptr()
                olds.append((old, src, m, v))
                news.append(outs)
            g.samples = samples.data_ptr() if samples is not None else None
            stats = [b.narrow(0, 0, n_new) for b in dst["stats"]]
            for k in range(3):
                g.stats_out[k] = stats[k].data_ptr()
            exist_old = self.exist_since_iter_
            exist_new = dst["exist"].narrow(0, 0, n_new)
            if exist_old is not None:   # every new row inherits its source's value (:636, :744, :782)
                assert exist_old.is_contiguous() and exist_old.dtype == torch.int32 and exist_old.numel() == P
                g.exist_since_iter_in, g.exist_since_iter_out = exist_old.data_ptr(), exist_new.data_ptr()
            morton = None
            if morton_reindex and n_new:   # (densifyAndPrune only: prunePoints copies its statistics by the mask's order)
                morton = torch.empty(int(lib.gsr_densify_morton_scratch_bytes(n_new)) + 256, dtype=torch.uint8, device=dev)
                g.morton_scratch = morton.data_ptr()
            if n_new:
                capi.check(lib, lib.gsr_densify_gather(C.byref(g), scratch.data_ptr(), stream), "gsr_densify_gather")
        for name, (old, _, m, _), outs in zip(self._PARAM_NAMES, olds, news):
            new = outs[0].detach().requires_grad_(True)
            setattr(self, name, new)
            if self.optimizer_ is not None:
                if m is not None:
                    self.optimizer_.replace_param(old, new, outs[1], outs[2])
                else:
                    self.optimizer_.replace_param(old, new, outs[1].zero_(), outs[2].zero_())
        self.xyz_gradient_accum_, self.denom_, self.max_radii2D_ = stats
        if exist_old is not None:
            self.exist_since_iter_ = exist_new
        return n_keep, n_clone, n_child, n_split, n_clone_sel, n_new

    def prunePoints(self, mask):
        """:588-642: keep = ~mask for the six tensors, their moments and the three statistics arrays (which, unlike in
        densifyAndPrune, keep their values)."""
        P = self.xyz_.shape[0]
        mask_u8 = mask.to(torch.uint8).contiguous()
        old_stats = (self.xyz_gradient_accum_, self.denom_, self.max_radii2D_)

        def select(a):
            a.prune_mask = mask_u8.data_ptr()
            return mask_u8
        self._compact(select)
        keep = ~mask.bool()
        self.xyz_gradient_accum_.copy_(old_stats[0][keep])
        self.denom_.copy_(old_stats[1][keep])
        self.max_radii2D_.copy_(old_stats[2][keep])

    def reorderAlongZCurve(self):
        """The whole model -- parameters, Adam moments, statistics, exist_since_iter_ -- laid out along a Z-order curve of the positions
        (what morton_reindex_ does inside densifyAndPrune, on request: a map that no longer densifies keeps growing at its end through
        increasePcd).  The same Gaussians with the same values; returns perm with new row r = old row perm[r].  The permutation is
        read off the gather itself: exist_since_iter_ travels through it as the row number."""
        P = self.xyz_.shape[0]
        dev = self.xyz_.device
        if P == 0:
            return torch.empty(0, dtype=torch.long, device=dev)
        old_stats = (self.xyz_gradient_accum_, self.denom_, self.max_radii2D_)
        old_exist = self.exist_since_iter_
        self.exist_since_iter_ = torch.arange(P, dtype=torch.int32, device=dev)
        none = torch.zeros(P, dtype=torch.uint8, device=dev)

        def select(a):
            a.prune_mask = none.data_ptr()
            return none
        self._compact(select, morton_reindex=True)
        perm = self.exist_since_iter_.long()   # (a copy: exist_since_iter_ is a view into the arena)
        self.xyz_gradient_accum_.copy_(old_stats[0][perm])
        self.denom_.copy_(old_stats[1][perm])
        self.max_radii2D_.copy_(old_stats[2][perm])
        if old_exist is not None:
            self.exist_since_iter_.copy_(old_exist[perm])
        else:
            self.exist_since_iter_ = None
        return perm

    def densifyAndPrune(self, max_grad, min_opacity, extent, max_screen_size, generator=None):
        """:795-815 (densifyAndClone :763-793, densifyAndSplit :716-761 with N = 2, prunePoints :588-642) in one rebuild."""
        P = self.xyz_.shape[0]
        tensors = (self.xyz_gradient_accum_.contiguous(), self.denom_.contiguous(), self.scaling_.detach().contiguous(),
                   self.opacity_.detach().contiguous())

        def select(a):
            a.xyz_gradient_accum, a.denom, a.scaling, a.opacity = (t.data_ptr() for t in tensors)
            a.percent_dense, a.max_grad, a.min_opacity, a.extent = self.percent_dense_, max_grad, min_opacity, extent
            a.max_screen_size = int(max_screen_size)
            return tensors
        # morton_reindex_ (off by default; GaussianModel::morton_reindex_ of the C++ host): the new set laid out along a Z-order curve
        # of the Gaussians' positions (include/gsr.h: gsr_densify_gather_args.morton_scratch) -- the same Gaussians in another order
        n_keep, n_clone, n_child, n_split, n_clone_sel, n_new = self._compact(select, generator,
                                                                              morton_reindex=getattr(self, "morton_reindex_", False))
        pruned = (P - n_split - n_keep) + (n_clone_sel - n_clone) + 2 * (n_split - n_child)
        return dict(cloned=n_clone_sel, split=n_split, pruned=pruned, points=n_new, children_kept=2 * n_child)

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
        self._replace_leaf("opacity_", new)

    # ---- loop closure (src/gaussian_model.cpp:379-475) ------------------------------------------------------------------------
    def _replace_leaf(self, name, values):
        """replaceTensorToOptimizer (:567-586) with a tensor of the same shape: a fresh leaf, zero Adam moments, the step counter
        carried.  While the leaf is a view of the arena the values are written into its rows and the moment rows zeroed in place,
        so that the next increasePcd / densifyAndPrune still finds the tensor where it expects it (re-seated outside, the next
        append rebuilt the whole arena: +25 ms after a resetOpacity at 4 M Gaussians in the C++ host's mapper-loop leg)."""
        old = getattr(self, name)
        arena = getattr(self, "_arena", None)
        slot = arena["sets"][arena["cur"]]["params"][name] if arena is not None else None
        if (slot is not None and old.data_ptr() == slot[0].data_ptr() and values.shape == old.shape and values.device == old.device
                and values.data_ptr() != old.data_ptr()):
            P = old.shape[0]
            with torch.no_grad():
                bufs = [b.narrow(0, 0, P) for b in slot]
                bufs[0].copy_(values.detach())
                bufs[1].zero_()
                bufs[2].zero_()
            new = bufs[0].detach().requires_grad_(True)
            setattr(self, name, new)
            if self.optimizer_ is not None:
                self.optimizer_.replace_param(old, new, bufs[1], bufs[2])
            return
        new = values.detach().contiguous().clone().requires_grad_(True)
        setattr(self, name, new)
        if self.optimizer_ is not None:
            self.optimizer_.replace_param(old, new)

    def applyScaledTransformation(self, s, T):
        """every point p <- s R p + t (transformPoints on the scaled positions, :385-388), then -- as shipped -- `scaling_ *= s` on
        the LOG-scales (:395: mirrored, not fixed); xyz and scaling become fresh leaves with zero moments
        (scaledTransformationPostfix :398-411).  T: the 4x4 matrix [R t; 0 1], row-major."""
        from . import operate_points as op
        with torch.no_grad():
            pts = (self.xyz_.detach() * s).contiguous()
            T_tensor = torch.as_tensor(T, dtype=torch.float32, device=pts.device).transpose(0, 1)
            pts = op.transformPoints(pts, T_tensor)
            scl = self.scaling_.detach() * s
        self._replace_leaf("xyz_", pts)
        self._replace_leaf("scaling_", scl)

    def scaledTransformVisiblePointsOfKeyframe(self, point_not_transformed_flags, diff_pose, kf_world_view_transform,
                                               kf_full_proj_transform, kf_creation_iter, stable_num_iter_existence, num_transformed=0,
                                               scale=1.0):
        """:408-475.  In place on point_not_transformed_flags; returns the new num_transformed.  rotation_ becomes the NORMALISED
        (and, for the moved points, rotated) quaternions; xyz and rotation get zero moments."""
        from . import operate_points as op
        with torch.no_grad():
            points = self.xyz_.detach().clone()
            rots = self.getRotationActivation().detach().clone()
            unstable = (self.exist_since_iter_ - kf_creation_iter).abs() < stable_num_iter_existence
            num_transformed = op.scaleAndTransformThenMarkVisiblePoints(points, rots, point_not_transformed_flags, unstable, diff_pose,
                                                                        kf_world_view_transform, kf_full_proj_transform,
                                                                        num_transformed, scale)
        self._replace_leaf("xyz_", points)
        self._replace_leaf("rotation_", rots)
        return num_transformed
