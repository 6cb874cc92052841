"""TEST INFRASTRUCTURE ONLY (see oracle/raster_oracle.c): plain-PyTorch restatement of the reference's densification
bookkeeping, used as the checker of egogaussian_amd/densify.py (row f-4).  Pinned by tests/golden/densify.npz, which was
captured from the reference's own GaussianModel (tests/golden/make_golden_densify.py).

Follows /root/reference/scene/gaussian_model.py:
  add_densification_stats :735-740 (+ trainers/train_static.py:125 for max_radii2D)
  densify_and_clone :642-676, densify_and_split :588-640, densify_and_prune :678-709, prune_points :536-563,
  densification_postfix :565-586, reset_opacity :484-490
including two behaviours of that code worth naming: densification_postfix zeroes max_radii2D, so after a clone or split step
the screen-size criterion never fires; and densify_and_prune hands curr_gen to densify_and_split positionally into its `N`
slot, so split children always inherit their parent's generation (and split_prev_gen=False raises).

State = dict of tensors: the seven parameters, "<name>_exp_avg", "<name>_exp_avg_sq", generation [N,1] int, is_object [N,1]
int, xyz_gradient_accum [N,1], denom [N,1], max_radii2D [N].
"""
import torch

PARAMS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "label")


def add_densification_stats(st, viewspace_grad, radii):
    vis = radii > 0
    st["max_radii2D"][vis] = torch.maximum(st["max_radii2D"][vis], radii[vis].to(st["max_radii2D"].dtype))
    st["xyz_gradient_accum"][vis] += viewspace_grad[vis, :2].norm(dim=-1, keepdim=True)
    st["denom"][vis] += 1
    return st


def _rotmat(q):
    q = q / q.norm(dim=1, keepdim=True)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).view(-1, 3, 3)


def _append(st, new, n_new):
    for k in PARAMS:
        st[k] = torch.cat([st[k], new[k]], 0)
        for m in ("_exp_avg", "_exp_avg_sq"):
            st[k + m] = torch.cat([st[k + m], torch.zeros_like(new[k])], 0)
    st["generation"] = torch.cat([st["generation"], new["generation"]], 0)
    st["is_object"] = torch.cat([st["is_object"], new["is_object"]], 0)
    n = st["xyz"].shape[0]
    st["xyz_gradient_accum"] = torch.zeros((n, 1)); st["denom"] = torch.zeros((n, 1)); st["max_radii2D"] = torch.zeros(n)


def _keep(st, keep):
    for k in list(st):
        st[k] = st[k][keep]


def densify_and_prune(st, max_grad, min_opacity, extent, max_screen_size, percent_dense=0.01, clone=True, split=True,
                      curr_gen=None, prune_prev_gen=True, which_object=None, z=None):
    st = {k: v.clone() for k, v in st.items()}
    grads = st["xyz_gradient_accum"] / st["denom"]
    grads[grads.isnan()] = 0.0
    thr = percent_dense * extent
    if clone:
        sel = (grads.norm(dim=-1) >= max_grad) & (torch.exp(st["scaling"]).max(dim=1).values <= thr)
        if which_object is not None:
            sel &= (st["is_object"] == which_object).squeeze(1)
        new = {k: st[k][sel] for k in PARAMS}
        new["generation"] = torch.full_like(st["generation"][sel], curr_gen) if curr_gen is not None else st["generation"][sel]
        new["is_object"] = st["is_object"][sel]
        _append(st, new, int(sel.sum()))
    if split:
        n = st["xyz"].shape[0]
        padded = torch.zeros(n)
        padded[:grads.shape[0]] = grads.squeeze(1)
        sel = (padded >= max_grad) & (torch.exp(st["scaling"]).max(dim=1).values > thr)
        if which_object is not None:
            sel &= (st["is_object"] == which_object).squeeze(1)
        scale = torch.exp(st["scaling"][sel]).repeat(2, 1)
        if callable(z):
            z = z(scale.shape[0])                                           # z(rows) -> [rows, 3] draws, as egogaussian_amd.densify accepts
        samples = scale * z                                               # torch.normal(mean=0, std=scale) with the recorded draw
        R = _rotmat(st["rotation"][sel]).repeat(2, 1, 1)
        new = {k: st[k][sel].repeat(2, *([1] * (st[k].dim() - 1))) for k in PARAMS}
        new["xyz"] = torch.bmm(R, samples.unsqueeze(-1)).squeeze(-1) + st["xyz"][sel].repeat(2, 1)
        new["scaling"] = torch.log(scale / (0.8 * 2))
        new["generation"] = st["generation"][sel].repeat(2, 1)             # parent's generation (see the module docstring)
        new["is_object"] = st["is_object"][sel].repeat(2, 1)
        _append(st, new, 2 * int(sel.sum()))
        _keep(st, ~torch.cat([sel, torch.zeros(2 * int(sel.sum()), dtype=torch.bool)]))
    prune = (torch.sigmoid(st["opacity"]) < min_opacity).squeeze(1)
    if max_screen_size:
        prune = prune | (st["max_radii2D"] > max_screen_size) | (torch.exp(st["scaling"]).max(dim=1).values > 0.1 * extent)
    if not prune_prev_gen:
        prune = prune & (st["generation"] == curr_gen).squeeze(1)
    _keep(st, ~prune)
    return st


def reset_opacity(st):
    st = {k: v.clone() for k, v in st.items()}
    o = torch.minimum(torch.sigmoid(st["opacity"]), torch.full_like(st["opacity"], 0.01))
    st["opacity"] = torch.log(o / (1 - o))
    st["opacity_exp_avg"] = torch.zeros_like(st["opacity"]); st["opacity_exp_avg_sq"] = torch.zeros_like(st["opacity"])
    return st
