"""Where an activated tensor came from -- so that the reference's OWN render(), unchanged, reaches the rasterizer's raw-parameter path.

/root/reference/gaussian_renderer/__init__.py:56-98 calls the model's getters and hands their results to the rasterizer:
    opacities     = pc.get_opacity                                   = opacity_activation(_opacity)                     (sigmoid)
    cov3D_precomp = pc.get_covariance(m)                             = covariance_activation(get_scaling, m, _rotation) (exp, normalize, R S S R^T)
                    pc.get_rotated_covariance(accum_R, k, training, m)   (fine_all.py:93: object Gaussians moved by accum_R)
    shs           = pc.get_features                                  = cat(_features_dc, _features_rest)
Each of those is a handful of element-wise launches forward, as many backward, and an autograd node or three -- per iteration, around a
rasterizer that can do all of it inside its preprocess kernel and hand the gradients straight to the leaves (include/egs_raster.h EGS_ACT_*,
split spherical harmonics, egs_object_rotation).  The trainer's loop must not change, so the connection is made through the tensors
themselves: adapter.attach() installs activations that leave `_egs_origin` on the tensor they return, and GaussianRasterizer.forward asks
`substitute()` whether the three tensors it was handed are such results, untouched, of raw parameters untouched since.  Only then are the raw
parameters used; in every other case the call proceeds with the tensors as given.  Values: the in-kernel activations equal torch's to the
last place or two (tests/test_gpu_provenance.py holds images and gradients of both routes together).

A tag is a plain Python attribute: it lives and dies with the tensor object, needs no registry, and is not copied by .detach(), .clone(),
slicing or any other op -- a tensor that is not the getter's own result is never mistaken for one."""
import torch

substitutions = 0          # how often GaussianRasterizer.forward took the raw parameters (tests and bench spy on it)


class Origin:
    __slots__ = ("kind", "raws", "versions", "own_version", "extra")

    def __init__(self, kind, raws, own, extra=None):
        self.kind, self.raws, self.extra = kind, raws, extra
        self.versions = tuple(r._version for r in raws)
        self.own_version = own._version


def tag(result, kind, raws, extra=None):
    result._egs_origin = Origin(kind, tuple(raws), result, extra)
    return result


def origin(t, kind):
    """The Origin of tensor `t` if it is an untouched getter result of that kind whose raw parameters are untouched since, else None."""
    o = getattr(t, "_egs_origin", None)
    if o is None or o.kind != kind or t._version != o.own_version:
        return None
    for r, v in zip(o.raws, o.versions):
        if r._version != v:
            return None
    return o


def tagging_activation(fn, kind):
    """fn (torch.exp, torch.sigmoid, ...) as an activation whose result remembers its argument.  The result of the most recent call is kept
    and returned again while the argument is unchanged (same object, same version, same grad mode): the reference calls get_scaling /
    get_opacity several times per iteration (render, the label render's rendervar, densification), each a launch."""
    last = {}

    def activation(x):
        key = (id(x), x._version, torch.is_grad_enabled())
        hit = last.get("entry")
        if hit is not None and hit[0] == key and hit[1] is x and hit[2]._version == hit[3]:
            return hit[2]
        y = fn(x)
        if torch.is_tensor(x) and x.is_leaf:
            tag(y, kind, (x,))
            last["entry"] = (key, x, y, y._version)
        return y
    activation.__wrapped__ = fn
    return activation


def tag_covariance(cov, scaling, scaling_modifier, rotation, object_rotation=None):
    """cov = covariance_activation(scaling, modifier, rotation): remembered as coming from (raw log-scales, raw quaternions) when `scaling`
    is itself a tagged exp() of a leaf and `rotation` is a leaf (the reference passes self._rotation, gaussian_model.py:167-171)."""
    o = origin(scaling, "scaling")
    if o is None or not (torch.is_tensor(rotation) and rotation.is_leaf):
        return cov
    return tag(cov, "covariance", (o.raws[0], rotation), extra=(float(scaling_modifier), object_rotation))


def tag_features(features, dc, rest):
    if dc.is_leaf and rest.is_leaf:
        tag(features, "features", (dc, rest))
    return features


def substitute(opacities, cov3D_precomp, shs, scale_modifier):
    """-> None, or dict(scales=, rotations=, opacities=, shs=, object_rotation=) of RAW parameters to rasterize instead of the activated
    tensors given.  All-or-nothing for the geometry (the rasterizer's raw-parameter mode activates scales, rotations and opacities together);
    the colour coefficients are swapped for their two stored halves when they are a tagged concatenation, else passed on as they are."""
    if cov3D_precomp is None or opacities is None:
        return None
    oc, oo = origin(cov3D_precomp, "covariance"), origin(opacities, "opacity")
    if oc is None or oo is None or oc.extra[0] != float(scale_modifier):
        return None
    raw_s, raw_r = oc.raws
    raw_o = oo.raws[0]
    if not (raw_s.is_cuda and raw_s.dtype == torch.float32 and raw_s.shape[0] == raw_o.shape[0] == raw_r.shape[0]):
        return None
    out = dict(scales=raw_s, rotations=raw_r, opacities=raw_o, shs=shs, object_rotation=oc.extra[1])
    if shs is not None:
        of = origin(shs, "features")
        if of is not None:
            out["shs"] = of.raws                                      # (features_dc, features_rest): the split form
    return out
