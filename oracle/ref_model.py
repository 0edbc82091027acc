"""Python access to oracle/_ref/libref_densify{,_cuda}.so: the reference's own GaussianModel map-maintenance and optimizer
code (src/gaussian_model.cpp:477-510, 553-831) behind torch ops -- see oracle/ref_densify.cpp.  TEST INFRASTRUCTURE ONLY."""
import torch

from . import build_ref

PARAMS = ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation")   # the reference's param-group order
_ops = {}


def load(kind="cpu"):
    """kind: "cpu" (this container and any host) or "cuda" (a GPU box: tensors on the HIP device).  None if the library
    was never built (the reference tree is absent and no prebuilt .so travelled)."""
    if kind not in _ops:
        path = build_ref.build_densify().get(kind)
        if path is None:
            _ops[kind] = None
        else:
            torch.ops.load_library(path)
            _ops[kind] = torch.ops.photoslam_reference_model if kind == "cpu" else torch.ops.photoslam_reference_model_cuda
    return _ops[kind]


class State:
    """A GaussianModel + Adam state as plain tensors: params / exp_avg / exp_avg_sq (six each, the reference's group order),
    steps (six ints), xyz_gradient_accum [P,1], denom [P,1], max_radii2D [P], exist_since_iter [P] int32."""

    def __init__(self, params, exp_avg, exp_avg_sq, steps, accum, denom, max_radii2D, exist_since_iter):
        self.params, self.exp_avg, self.exp_avg_sq, self.steps = list(params), list(exp_avg), list(exp_avg_sq), list(steps)
        self.accum, self.denom, self.max_radii2D, self.exist_since_iter = accum, denom, max_radii2D, exist_since_iter

    def args(self):
        return (self.params, self.exp_avg, self.exp_avg_sq, self.steps, self.accum, self.denom, self.max_radii2D,
                self.exist_since_iter)

    @staticmethod
    def from_dump(out):
        return State(out[0:6], out[6:12], out[12:18], [int(s) for s in out[22].tolist()], out[18], out[19], out[20], out[21])

    @property
    def features(self):
        """[P,16,3]: cat(features_dc, features_rest), the single SH leaf of this repository's hosts"""
        return torch.cat([self.params[1], self.params[2]], 1)


def densify_and_prune(ops, st, percent_dense, max_grad, min_opacity, extent, max_screen_size):
    return State.from_dump(ops.densify_and_prune(*st.args(), percent_dense, max_grad, min_opacity, extent, max_screen_size))


def increase_pcd(ops, st, points, colors, iteration, vector_overload=False):
    """the reference's GaussianModel::increasePcd (src/gaussian_model.cpp:188-376), either overload; distCUDA2 is the CPU
    oracle's kNN (pinned to simple_knn.cu), see oracle/ref_densify.cpp"""
    return State.from_dump(ops.increase_pcd(*st.args(), points, colors, int(iteration), bool(vector_overload)))


def reset_opacity(ops, st):
    return State.from_dump(ops.reset_opacity(*st.args()))


def prune_points(ops, st, mask):
    return State.from_dump(ops.prune_points(*st.args(), mask))


def adam_step(ops, st, grads, spatial_lr_scale=1.0, xyz_lr=-1.0):
    return State.from_dump(ops.adam_step(st.params, list(grads), st.exp_avg, st.exp_avg_sq, st.steps, spatial_lr_scale, xyz_lr))


def save_ply(ops, params, path):
    """the reference's GaussianModel::savePly (tinyply) on the six tensors (reference order: xyz, f_dc, f_rest, opacity, scaling, rotation)"""
    ops.save_ply(list(params), str(path))


def load_ply(ops, path, max_sh_degree=3):
    """the reference's GaussianModel::loadPly -> (xyz, f_dc [P,1,3], f_rest [P,M-1,3], opacity, scaling, rotation, active degree)"""
    out = ops.load_ply(str(path), max_sh_degree)
    return out[:6], int(out[6])
