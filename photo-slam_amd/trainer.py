"""The measured train step: GaussianMapper::trainForOneIteration (src/gaussian_mapper.cpp:614-774)
without the SLAM keyframe scheduling -- render -> mask -> L1 + lambda*(1-SSIM) -> backward ->
densification statistics -> Adam -- plus the keyframe-batch data parallelism of SURVEY.md 8(e):
one keyframe per rank, mean of the five leaf gradients over the ranks (ViewFactoredExchange, or the plain
all-reduce of GradientReduction), SUM/MAX of the statistics."""
import torch
import torch.distributed as dist

from . import loss_utils
from .gaussian_renderer import GaussianRenderer


FEATURES_GROUP = 1   # position of features_ in GaussianModel.params() and in the optimizer's groups


def _one_buffer(tensors):
    """A flat view over the storage the given contiguous tensors tile exactly and without gaps (the rasterizer's backward
    hands the gradients of the four small parameter tensors out as slices of one buffer), or None."""
    if len(tensors) < 2 or any(t is None or not t.is_contiguous() or t.dtype != torch.float32 for t in tensors):
        return None
    base = tensors[0].untyped_storage()
    if any(t.untyped_storage().data_ptr() != base.data_ptr() for t in tensors):
        return None
    spans = sorted((t.storage_offset(), t.numel()) for t in tensors)
    end = spans[0][0]
    for off, n in spans:
        if off != end:
            return None
        end = off + n
    if spans[0][0] != 0 or end * 4 != base.nbytes():
        return None
    return torch.empty(0, dtype=torch.float32, device=tensors[0].device).set_(base, 0, (end,))


class GradientReduction:
    """Mean of the per-view gradients over the ranks, overlapped with the optimizer: every tensor's all-reduce is issued
    asynchronously right after backward, largest first (the [P,16,3] SH gradient is 81 % of the 472 MB), and wait(i) blocks
    only the compute stream, only for tensor i -- so Adam on the SH tensor runs while the small reductions are still on the
    links, and their Adam follows.  Tensors that tile one buffer (the four small gradients, see _one_buffer) travel as ONE
    collective.  RCCL averages inside the collective (ncclAvg); gloo (the CPU test path) has no AVG: it sums, and wait()
    scales."""

    def __init__(self, tensors, world_size, sum_only=False):
        """sum_only: the collective SUMS and nothing scales -- the consumer multiplies by grad_scale() = 1/N as it reads the
        gradient (FusedAdam.step_groups): no averaging pass over the buffer (ncclAvg = pre-multiply + sum: with ONE rank a
        whole extra kernel)"""
        self.tensors_, self.world_size_ = tensors, world_size
        self.sum_only_ = sum_only
        self.avg_ = not sum_only and dist.get_backend() == "nccl"
        op = dist.ReduceOp.AVG if self.avg_ else dist.ReduceOp.SUM
        self.order_ = sorted(range(len(tensors)), key=lambda i: -tensors[i].numel())
        # members of one buffer share a single collective: group -> [tensor to reduce, work, scaled?]
        self.group_of_, self.groups_ = {}, []
        by_storage = {}
        for i in self.order_:
            by_storage.setdefault(tensors[i].untyped_storage().data_ptr(), []).append(i)
        for members in by_storage.values():
            flat = _one_buffer([tensors[i] for i in members])
            if flat is not None:
                for i in members:
                    self.group_of_[i] = len(self.groups_)
                self.groups_.append([flat, None, False])
        for i in self.order_:
            if i not in self.group_of_:
                self.group_of_[i] = len(self.groups_)
                self.groups_.append([tensors[i], None, False])
        issued = set()
        for i in self.order_:   # issue in size order; a shared buffer goes out when its first member comes up
            k = self.group_of_[i]
            if k not in issued:
                issued.add(k)
                self.groups_[k][1] = dist.all_reduce(self.groups_[k][0], op=op, async_op=True)

    def collectives(self):
        return len(self.groups_)

    def order(self):
        return self.order_

    def wait(self, i):
        grp = self.groups_[self.group_of_[i]]
        grp[1].wait()
        if not self.avg_ and not self.sum_only_ and not grp[2]:
            grp[0].mul_(1.0 / self.world_size_)
            grp[2] = True

    def grad_scale(self):
        return 1.0 / self.world_size_ if self.sum_only_ else 1.0

    def scale_now(self):
        """sum_only mode after all: average in place (a consumer that cannot apply the scale itself)"""
        if self.sum_only_:
            self.wait_all()
            for grp in self.groups_:
                if not grp[2]:
                    grp[0].mul_(1.0 / self.world_size_)
                    grp[2] = True
            self.sum_only_ = False

    def wait_all(self):
        for i in self.order_:
            self.wait(i)


class ViewFactoredExchange:
    """The same batch mean with 2.6x (8 ranks) to 4.2x (2 ranks) fewer bytes on the links.  81 % of
    the gradient is the [P,16,3] SH tensor, and one view's SH gradient is rank one per Gaussian: basis(dir) x dL_dcolor, with dir known to every rank.
    So the ranks ALL-GATHER the 3-float colour gradients (rasterizer backward with sh_grad_view_; in PARTS row ranges) and
    the camera centres, each rebuilds the mean SH gradient locally (gsr_sh_grad_from_views), and only the other four tensors
    (11 floats per Gaussian) are all-reduced.  Per Gaussian a rank sends (N-1) * 12 + 2 (N-1)/N * 44 B instead of 2 (N-1)/N * 236 B.

    Usage: send, color_view = ViewFactoredExchange.send_buffer(P, device) before the render; color_view ([P,3], rows 0..P-1 of
    send) is handed to the backward (row P receives the camera centre: one gather carries both); others =
    [(index, grad), ...] of the remaining parameters (one all-reduce when they share a buffer, GradientReduction); after
    construction everything is in flight.  sh_gradient() / sh_adam_step() wait for the gather; they read means3D, so call
    them BEFORE Adam moves the positions."""

    @staticmethod
    def send_buffer(P, device):
        send = torch.empty((P + 1, 3), dtype=torch.float32, device=device)
        return send, send[:P]

    PARTS = 1   # ONE all-gather: the colour gradients with the camera centre as row P (every collective costs a launch on
                # RCCL's stream and two cross-stream hand-offs: at one rank four collectives per step cost ~0.1 ms more than two)

    def __init__(self, send, camera_center, others, world_size):
        self.world_size_ = world_size
        P = send.size(0) - 1
        dev = send.device
        self.P_ = P
        self.nccl_ = dist.get_backend() == "nccl"
        send[P].copy_(camera_center.detach().reshape(3).to(torch.float32))
        gathered = torch.empty((world_size, P + 1, 3), dtype=torch.float32, device=dev)
        if self.nccl_:
            work = dist.all_gather_into_tensor(gathered, send.unsqueeze(0), async_op=True)
        else:
            # gloo (the CPU test path) gathers host tensors
            host = gathered.cpu()
            dist.all_gather_into_tensor(host, send.unsqueeze(0).cpu().contiguous())
            gathered.copy_(host)
            work = None
        self.centre_work_ = None
        self.centres_ = gathered[:, P]            # [N, 3], stride (P + 1) * 3
        self.parts_ = [[0, gathered[:, :P], work]]   # [row0, views [N, P, 3] (strided over the views), work or None]
        self.indices_ = [i for i, _ in others]
        # (summed, not averaged: the Adam launch of the four small tensors multiplies by 1/N as it reads the gradients)
        self.reduction_ = GradientReduction([t for _, t in others], world_size, sum_only=True)

    def gathered_parts(self):
        """Yields (first row, camera centres [N,3], colour gradients [N,rows,3]) part by part, each once ITS all-gather has
        landed (stream-side wait: the host keeps queueing)."""
        if self.centre_work_ is not None:
            self.centre_work_.wait()
            self.centre_work_ = None
        for part in self.parts_:
            if part[2] is not None:
                part[2].wait()
                part[2] = None
            yield part[0], self.centres_, part[1]

    def gathered(self):
        """(camera centres [N,3], colour gradients [N,P,3]) of all ranks in one tensor (a copy when the gather ran in parts)"""
        parts = [v for _, _, v in self.gathered_parts()]
        return self.centres_, parts[0] if len(parts) == 1 else torch.cat(parts, 1)

    def sh_gradient(self, means3D, degree, M, out=None):
        from . import rasterize_points as rp
        centres, views = self.gathered()
        return rp.shGradFromViews(means3D.detach(), centres, views, degree, M, 1.0 / self.world_size_, out)

    def sh_adam_step(self, means3D, degree, sh, sh_adam):
        """The rebuild and the Adam step of the SH tensor in one pass per part (gsr_sh_adam_from_views): the mean gradient
        never reaches HBM.  sh_adam: FusedAdam.begin_fused_step(FEATURES_GROUP)."""
        from . import rasterize_points as rp
        m3, shd = means3D.detach(), sh.detach()
        lazy = sh_adam.get("row_step") is not None
        if lazy and int(sh_adam["window"]) < 3:
            raise RuntimeError("lazy rows in the view-factored step need a window of at least 3")
        for row0, centres, views in self.gathered_parts():
            n = views.size(1)
            part = dict(sh_adam, exp_avg=sh_adam["exp_avg"][row0:row0 + n], exp_avg_sq=sh_adam["exp_avg_sq"][row0:row0 + n])
            if lazy:
                part["row_step"] = sh_adam["row_step"][row0:row0 + n]
            rp.shAdamFromViews(m3[row0:row0 + n], centres, views, degree, 1.0 / self.world_size_, shd[row0:row0 + n], part)
        # (lazy rows, gsr_sh_adam_lazy: the rows no view of the batch lights were left alone above; the rotating catch-up that
        # bounds their lag ran inside the rasterizer's backward, next to the blend kernel -- gsr_backward_args.sh_adam together
        # with dL_dcolor_view)

    def order(self):
        """parameter indices of the all-reduced tensors in completion order"""
        return [self.indices_[j] for j in self.reduction_.order()]

    def wait(self, index):
        self.reduction_.wait(self.indices_.index(index))

    def wait_all(self):
        for _ in self.gathered_parts():
            pass
        self.reduction_.wait_all()


def allreduce_mean(tensors, world_size):
    """In-place mean over the ranks, all reductions in flight together."""
    GradientReduction(tensors, world_size).wait_all()


class TrainStep:
    def __init__(self, gaussians, opt, pipe, background, world_size=1, cameras_extent=None, densify=False,
                 densify_min_opacity=0.005, prune_big_point_after_iter=30000, seed=0, factored_exchange=True, fused_sh_adam=True,
                 lazy_sh_adam_window=32, fused_geom_adam=True, cull_empty_tiles=False):
        # the rasterizer's scratch buffers, kept across iterations and grown with headroom (rasterize_points.RasterWorkspace);
        # persistent_workspace_ = False: fresh buffers per call, as the reference's resizeFunctional
        from . import rasterize_points as rp
        self.persistent_workspace_ = True
        self.workspace_ = rp.RasterWorkspace()
        self.cull_empty_tiles_ = bool(cull_empty_tiles)   # option of this object; GSR_CULL_EMPTY_TILES only overrides (gaussian_renderer.py)
        self.gaussians_, self.opt_, self.pipe_, self.background_ = gaussians, opt, pipe, background
        self.cameras_extent_ = cameras_extent if cameras_extent is not None else gaussians.spatial_lr_scale_
        self.densify_, self.densify_min_opacity_ = densify, densify_min_opacity
        self.prune_big_point_after_iter_ = prune_big_point_after_iter
        # every rank draws the same split samples: identical replicas without a broadcast
        self.generator_ = torch.Generator(device=gaussians.xyz_.device).manual_seed(seed)
        self.last_densify_ = None
        self.iteration_ = 0
        self.world_size_ = world_size
        # world_size > 1: ViewFactoredExchange (default) or the plain all-reduce of all five gradients
        self.factored_exchange_ = factored_exchange
        # world_size == 1: the Adam step of the SH tensor runs inside the rasterizer's backward (its 192 B/Gaussian gradient
        # row never reaches HBM); same arithmetic, same result as the separate pass
        self.fused_sh_adam_ = fused_sh_adam
        # ... and the zero-gradient steps of the culled Gaussians' rows taken lazily, at most this many at a time (0 = every
        # row at every step): TrainStep::lazy_sh_adam_window_ of the C++ host
        self.lazy_sh_adam_window_ = lazy_sh_adam_window
        # ... and the Adam steps of xyz / opacity / scaling / rotation inside the backward kernels that hold their gradients
        # (TrainStep::fused_geom_adam_ of the C++ host), on the same iterations
        self.fused_geom_adam_ = fused_geom_adam
        self.ema_loss_for_log_ = 0.0

    def _effective_mask(self, mask):
        """rendered * mask with a mask of ones is the identity (src/gaussian_mapper.cpp:692-693; most keyframes carry a full
        mask): such a mask is recognised ONCE per mask tensor (one reduction + host read when a keyframe's mask is first seen)
        and the loss kernels then skip its 2 x 25 MB of reads at 1080p.  Remembered per tensor OBJECT through a weak reference
        (an address or id alone could be reused by another tensor after this one is freed) together with its version counter
        (in-place writes invalidate the entry)."""
        if mask is None:
            return None
        import weakref
        cache = self.__dict__.setdefault("_mask_is_ones", {})
        entry = cache.get(id(mask))
        if entry is None or entry[0]() is not mask or entry[1] != mask._version:
            if len(cache) >= 64:
                cache.clear()
            with torch.no_grad():
                entry = cache[id(mask)] = (weakref.ref(mask), mask._version, bool((mask == 1).all().item()))
        return None if entry[2] else mask

    def trainForOneIteration(self, viewpoint_cam, gt_image, mask, sync_loss=True, position_lr_step=None):
        """position_lr_step: the step of the position learning-rate schedule -- None = the iteration (src/gaussian_mapper.cpp:672-674,
        the COLMAP flavour); a SLAM session passes the keyframe's use count (:663-671, capped at position_lr_max_steps_)."""
        g, opt = self.gaussians_, self.opt_
        self.iteration_ += 1
        it = self.iteration_
        g.updateLearningRate(it if position_lr_step is None else min(int(position_lr_step), opt.position_lr_max_steps_))   # :661-674
        sh_send = sh_view = sh_adam = geom_adam = sh_adam_views = None
        if self.world_size_ > 1 and self.factored_exchange_:
            sh_send, sh_view = ViewFactoredExchange.send_buffer(g.xyz_.size(0), g.xyz_.device)
        # this iteration densifies (src/gaussian_mapper.cpp:720-721; the fresh leaves have no gradient, so the reference's
        # optimizer step skips every group)
        rebuilds = bool(self.densify_ and it < opt.densify_until_iter_ and it > opt.densify_from_iter_ and
                        opt.densification_interval_ and it % opt.densification_interval_ == 0)
        # The fused optimizer steps advance their step counters HERE, so they are taken only when render() will really hand
        # them to the rasterizer: with convert_SHs_ the rasterizer sees colours, not the SH tensor (no SH step to fuse), and
        # with compute_cov3D_ it sees a covariance, not the raw scaling / rotation leaves (no geometry step to fuse) --
        # autograd then leaves dense gradients and the ordinary optimizer step below takes those groups.
        if self.world_size_ == 1 and self.fused_sh_adam_ and it < opt.iterations_ and not rebuilds and \
                g._features.size(1) == 16 and g.optimizer_ is not None and not self.pipe_.convert_SHs_:
            sh_adam = g.optimizer_.begin_fused_step(FEATURES_GROUP, self.lazy_sh_adam_window_)
            # (an iteration that resets the opacity replaces that leaf AFTER backward: the reference's optimizer step then
            # skips it -- no gradient -- while a step fused into backward would already have been taken: src/gaussian_mapper.cpp:732-735)
            resets = bool(self.densify_ and it < opt.densify_until_iter_ and opt.opacity_reset_interval_ and
                          it % opt.opacity_reset_interval_ == 0)
            if self.fused_geom_adam_ and len(g.optimizer_.param_groups) == 5 and not self.pipe_.compute_cov3D_ and not resets:
                # xyz, opacity, scaling, rotation = groups 0, 2, 3, 4 (GaussianModel.trainingSetup)
                tensors = []
                for gi in (0, 2, 3, 4):
                    d = g.optimizer_.begin_fused_step(gi)
                    tensors.append((g.optimizer_.param_groups[gi]["params"][0].detach(), d["exp_avg"], d["exp_avg_sq"], d["lr"], d["step"]))
                geom_adam = dict(tensors=tensors, beta1=sh_adam["beta1"], beta2=sh_adam["beta2"], eps=sh_adam["eps"])
        # Data-parallel step with the view-factored exchange: the SH rows step AFTER the exchange (gsr_sh_adam_from_views), and
        # lazily there too -- a row no view of the batch lights takes a zero-gradient step, i.e. it may take it later; the
        # forward pass gets the same struct so that the rows THIS view sees are up to date before they are evaluated.
        if sh_view is not None and self.lazy_sh_adam_window_ >= 3 and it < opt.iterations_ and not rebuilds and \
                g._features.size(1) == 16 and g.optimizer_ is not None and not self.pipe_.convert_SHs_ and \
                g._features.is_contiguous():
            sh_adam_views = g.optimizer_.begin_fused_step(FEATURES_GROUP, self.lazy_sh_adam_window_)
        # this view's densification statistics (:714-719) are added by the backward kernel that holds dL_dmean2D.  With
        # several ranks they accumulate PER RANK and are reduced only when densification consumes them (below): SUM and MAX
        # commute with the accumulation over iterations, so nothing crosses the links for them on the other 99 of 100 steps
        view_stats = (g.xyz_gradient_accum_, g.denom_, g.max_radii2D_) if it < opt.densify_until_iter_ else None
        fwd_adam = sh_adam if sh_adam is not None else sh_adam_views
        g._in_lazy_step = fwd_adam is not None and fwd_adam.get("row_step") is not None
        try:
            rendered_image, viewspace_point_tensor, visibility_filter, radii = GaussianRenderer.render(
                viewpoint_cam, viewpoint_cam.image_height_, viewpoint_cam.image_width_, g, self.pipe_, self.background_,
                sh_grad_view=sh_view, sh_adam=fwd_adam, view_stats=view_stats, geom_adam=geom_adam,
                training_outputs_only=True,   # the statistics are fused (or over): nobody reads the viewspace gradient
                cull_empty_tiles=self.cull_empty_tiles_, workspace=self.workspace_ if self.persistent_workspace_ else None)
        finally:
            g._in_lazy_step = False
        # :692-698  masked L1 + lambda * (1 - SSIM), fused with its gradient (csrc/train_ops.hip)
        loss = loss_utils.fused_l1_ssim_loss(rendered_image, gt_image, self._effective_mask(mask), opt.lambda_dssim_, is_root=True)
        # :699 (the root gradient: a cached 1 instead of the ones_like fill autograd launches per backward())
        if getattr(self, "_root_grad", None) is None or self._root_grad.device != loss.device:
            self._root_grad = torch.ones_like(loss).detach()
        loss.backward(self._root_grad)
        if sh_adam is not None:
            g.optimizer_.end_fused_step(FEATURES_GROUP, sh_adam)
        with torch.no_grad():
            reduction = None
            if self.world_size_ > 1:
                # keyframe-batch data parallelism: mean of the per-view gradients over RCCL, in flight from here on
                if sh_view is not None:
                    reduction = ViewFactoredExchange(sh_send, viewpoint_cam.camera_center_,
                                                     [(i, p.grad) for i, p in enumerate(g.params_raw()) if i != FEATURES_GROUP],
                                                     self.world_size_)
                else:
                    reduction = GradientReduction([p.grad for p in g.params()], self.world_size_)
            if it < opt.densify_until_iter_:
                if self.densify_:
                    if reduction is not None and rebuilds:
                        # every tensor is about to be rebuilt and this step's update is skipped: the exchange only has to
                        # finish.  (An opacity reset alone replaces ONE leaf: the other groups keep their gradients and
                        # step below through the reduction, the reset opacity has no gradient and is skipped -- :732-735.)
                        reduction.wait_all()
                        reduction = None
                    if rebuilds:                                                                    # :721-730
                        if self.world_size_ > 1:
                            # the batch's statistics since the last densification: norm sums and counts SUM, radii MAX
                            dist.all_reduce(g.xyz_gradient_accum_, op=dist.ReduceOp.SUM)
                            dist.all_reduce(g.denom_, op=dist.ReduceOp.SUM)
                            dist.all_reduce(g.max_radii2D_, op=dist.ReduceOp.MAX)
                        size_threshold = 20 if it > self.prune_big_point_after_iter_ else 0   # :723
                        g.optimizer_.zero_grad(set_to_none=True)   # shapes change; this step's update is skipped
                        self.last_densify_ = g.densifyAndPrune(opt.densify_grad_threshold_, self.densify_min_opacity_,
                                                               self.cameras_extent_, size_threshold,
                                                               generator=self.generator_)
                    if opt.opacity_reset_interval_ and it % opt.opacity_reset_interval_ == 0:      # :732-735
                        g.resetOpacity()
            if it < opt.iterations_:                                       # :769-772
                if reduction is None:
                    g.optimizer_.step()
                else:
                    # each tensor is updated as soon as ITS reduction has landed (largest first): Adam on the SH tensor
                    # overlaps the four small reductions still on the links
                    g.optimizer_.begin_step()
                    if sh_view is not None:
                        # the SH gradient is rebuilt from the gathered views (reads xyz_: before ITS update) and applied
                        # while the all-reduces of the other four tensors are on the links
                        if sh_adam_views is not None:   # rebuild + Adam in one pass, lazy rows
                            g._in_lazy_step = True
                            try:
                                reduction.sh_adam_step(g.xyz_, g.active_sh_degree_, g._features, sh_adam_views)
                            finally:
                                g._in_lazy_step = False
                            g.optimizer_.end_fused_step(FEATURES_GROUP, sh_adam_views)
                            sh_adam_views = None
                        elif g.features_.size(1) == 16:   # rebuild + Adam in one pass
                            reduction.sh_adam_step(g.xyz_, g.active_sh_degree_, g.features_,
                                                   g.optimizer_.begin_fused_step(FEATURES_GROUP))
                        else:
                            g.features_.grad = reduction.sh_gradient(g.xyz_, g.active_sh_degree_, g.features_.size(1))
                            g.optimizer_.step_group(FEATURES_GROUP)
                    small = [i for i in reduction.order() if i != FEATURES_GROUP]
                    if sh_view is not None:
                        # the small gradients arrived SUMMED (in ONE all-reduce when they share a buffer): one Adam launch for
                        # the four tensors, which applies the 1/N as it reads them
                        for i in small:
                            reduction.wait(i)
                        g.optimizer_.step_groups(small, reduction.reduction_.grad_scale())
                    else:
                        for i in reduction.order():
                            reduction.wait(i)
                            g.optimizer_.step_group(i)
                g.optimizer_.zero_grad(set_to_none=True)
            elif reduction is not None:
                reduction.wait_all()
            if sync_loss:
                # :705 (the reference's per-iteration host sync for the loss EMA) -- moved behind the optimizer launches, so
                # that Adam is already queued behind backward while the host waits
                self.ema_loss_for_log_ = 0.4 * loss.item() + 0.6 * self.ema_loss_for_log_
        return loss
