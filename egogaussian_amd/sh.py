"""Real spherical-harmonics colour evaluation in PyTorch (degrees 0-3) for the `convert_SHs_python` branch of
render() (/root/reference/gaussian_renderer/__init__.py:78-83).  Same basis, constants and coefficient order as
/root/reference/utils/sh_utils.py:57-118, which is also what the HIP preprocess kernel evaluates in-kernel."""
import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435)


def sh_basis(deg, dirs):
    """dirs [..., 3] unit vectors -> basis [..., (deg+1)^2]."""
    if not 0 <= deg <= 3:
        raise ValueError("SH degree must be in 0..3")
    x, y, z = dirs[..., 0], dirs[..., 1], dirs[..., 2]
    b = [torch.full_like(x, C0)]
    if deg > 0:
        b += [-C1 * y, C1 * z, -C1 * x]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        b += [C2[0] * xy, C2[1] * yz, C2[2] * (2 * zz - xx - yy), C2[3] * xz, C2[4] * (xx - yy)]
    if deg > 2:
        b += [C3[0] * y * (3 * xx - yy), C3[1] * xy * z, C3[2] * y * (4 * zz - xx - yy),
              C3[3] * z * (2 * zz - 3 * xx - 3 * yy), C3[4] * x * (4 * zz - xx - yy), C3[5] * z * (xx - yy),
              C3[6] * x * (xx - 3 * yy)]
    return torch.stack(b, dim=-1)


def eval_sh(deg, sh, dirs):
    """sh [..., C, >=(deg+1)^2], dirs [..., 3] -> [..., C]."""
    n = (deg + 1) ** 2
    return (sh[..., :n] * sh_basis(deg, dirs)[..., None, :]).sum(-1)


def RGB2SH(rgb):
    return (rgb - 0.5) / C0


def SH2RGB(sh):
    return sh * C0 + 0.5
