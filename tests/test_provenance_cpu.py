"""CPU: egogaussian_amd/provenance.py -- the tags by which the reference's own render() reaches the rasterizer's raw-parameter path.  A tag
must vouch for a tensor only while that very object is untouched AND its raw parameters are untouched; everything else is an ordinary tensor."""
import torch

from egogaussian_amd import provenance as prov
from tests.test_adapter import RefShaped, _scene


def _model():
    import egogaussian_amd
    m = RefShaped(_scene(deg=1), sh_degree=1)
    m.training_setup()
    return egogaussian_amd.attach(m)


def test_getter_results_are_tagged_and_the_tags_expire():
    m = _model()
    s = m.get_scaling
    o = prov.origin(s, "scaling")
    assert o is not None and o.raws[0] is m._scaling and prov.origin(s, "opacity") is None
    assert torch.equal(s, torch.exp(m._scaling))                                   # an ordinary tensor with ordinary values and history
    assert s.grad_fn is not None
    assert m.get_scaling is s                                                      # unchanged parameter: the same result object (no second launch)
    with torch.no_grad():
        assert m.get_scaling is not s                                              # another grad mode: another result
    # copies and views are NOT the getter's result
    assert prov.origin(s.detach(), "scaling") is None and prov.origin(s.clone(), "scaling") is None and prov.origin(s[:10], "scaling") is None
    cov = m.get_covariance(1.0)
    oc = prov.origin(cov, "covariance")
    assert oc is not None and oc.raws[0] is m._scaling and oc.raws[1] is m._rotation and oc.extra == (1.0, None)
    f = m.get_features
    assert prov.origin(f, "features").raws == (m._features_dc, m._features_rest)
    op = m.get_opacity
    assert prov.origin(op, "opacity").raws[0] is m._opacity
    # the raw parameter changes (an optimizer step, in place): every tag that names it expires, the next getter call computes afresh
    with torch.no_grad():
        m._scaling.add_(0.1)
    assert prov.origin(s, "scaling") is None and prov.origin(cov, "covariance") is None and prov.origin(op, "opacity") is not None
    s2 = m.get_scaling
    assert s2 is not s and torch.equal(s2, torch.exp(m._scaling)) and prov.origin(s2, "scaling") is not None
    # the RESULT changes in place: its own tag expires
    with torch.no_grad():
        op.mul_(0.5)
    assert prov.origin(op, "opacity") is None and m.get_opacity is not op
    # a covariance built from something that is not a tagged exp() of a leaf carries no tag
    assert prov.origin(m.covariance_activation(torch.exp(m._scaling) * 1.0, 1.0, m._rotation), "covariance") is None
    # densification replaces the Parameters: new objects, new results
    m._opacity = torch.nn.Parameter(m._opacity.detach().clone())
    assert prov.origin(m.get_opacity, "opacity").raws[0] is m._opacity


def test_substitute_is_all_or_nothing_and_leaves_cpu_tensors_alone():
    m = _model()
    cov, op, f = m.get_covariance(1.0), m.get_opacity, m.get_features
    assert prov.substitute(op, cov, f, 1.0) is None                                # CPU tensors: the rasterizer's raw path is a HIP kernel
    assert prov.substitute(None, cov, f, 1.0) is None and prov.substitute(op, None, f, 1.0) is None
    # (the decision logic itself, device check aside)
    real = torch.Tensor.is_cuda
    try:
        torch.Tensor.is_cuda = property(lambda self: True)
        sub = prov.substitute(op, cov, f, 1.0)
        assert sub["scales"] is m._scaling and sub["rotations"] is m._rotation and sub["opacities"] is m._opacity
        assert sub["shs"] == (m._features_dc, m._features_rest) and sub["object_rotation"] is None
        assert prov.substitute(op, cov, f, 2.0) is None                            # another scale modifier than the covariance was built with
        assert prov.substitute(op.detach(), cov, f, 1.0) is None                   # one untagged member: nothing is substituted
        assert prov.substitute(op, cov, f.clone(), 1.0)["shs"].shape == f.shape    # untagged colours are passed on as given
    finally:
        torch.Tensor.is_cuda = real


def test_fused_adam_step_counts_as_an_in_place_write():
    """optim._touched: the HIP Adam kernels write parameters through raw pointers; the version counter must say so (CPU: the helper alone)."""
    from egogaussian_amd.optim import _touched
    p = torch.nn.Parameter(torch.zeros(4))
    y = prov.tagging_activation(torch.exp, "scaling")(p)
    assert prov.origin(y, "scaling") is not None
    _touched(p)
    assert prov.origin(y, "scaling") is None
