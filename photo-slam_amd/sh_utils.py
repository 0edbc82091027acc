"""include/sh_utils.h:64-146 of the reference: eval_sh (degrees 0..3), RGB2SH, SH2RGB -- as a basis matrix times the
coefficients (colour[n][c] = sum_k basis_k(dir[n]) * sh[n][c][k]); constants and basis functions are those of
computeColorFromSH (cuda_rasterizer/forward.cu:20-71), which the in-kernel evaluation uses too."""
import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658, 1.445305721320277,
      -0.5900435899266435)


def basis(deg, dirs):
    """[N, (deg+1)^2] real SH basis at the unit directions dirs [N,3]"""
    assert 0 <= deg <= 3
    x, y, z = dirs[..., 0], dirs[..., 1], dirs[..., 2]
    b = [torch.full_like(x, C0)]
    if deg > 0:
        b += [-C1 * y, C1 * z, -C1 * x]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        b += [C2[0] * xy, C2[1] * yz, C2[2] * (2.0 * zz - xx - yy), C2[3] * xz, C2[4] * (xx - yy)]
        if deg > 2:
            b += [C3[0] * y * (3.0 * xx - yy), C3[1] * xy * z, C3[2] * y * (4.0 * zz - xx - yy),
                  C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy), C3[4] * x * (4.0 * zz - xx - yy), C3[5] * z * (xx - yy),
                  C3[6] * x * (xx - 3.0 * yy)]
    return torch.stack(b, -1)


def eval_sh(deg, sh, dirs):
    """sh [N, C, >= (deg+1)^2], dirs [N,3] (unit) -> [N, C]"""
    K = (deg + 1) ** 2
    assert sh.shape[-1] >= K
    return (sh[..., :K] * basis(deg, dirs).unsqueeze(-2)).sum(-1)


def RGB2SH(rgb):
    return (rgb - 0.5) / C0


def SH2RGB(sh):
    return sh * C0 + 0.5
