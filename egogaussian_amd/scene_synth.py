"""Seeded synthetic scenes and cameras of the shapes the reference's trainers feed to render().

No dataset is reachable from the build or GPU boxes (HOI4D / EPIC-KITCHENS are Google-Drive
downloads, /root/reference/README.md:8,30), so every benchmark and parity case runs on the
instance family S(N, H, W, seed) defined in SURVEY.md section 8d.

Camera matrices follow the reference's conventions exactly:
  * projection_matrix      <- getProjectionMatrix, /root/reference/utils/graphics_utils.py:51-71
  * world_view_transform   <- W2V transposed,      /root/reference/scene/cameras.py:67
  * full_proj_transform    <- W2V^T @ P^T,          /root/reference/scene/cameras.py:68-69
  * camera_center          <- inverse(W2V^T)[3,:3], /root/reference/scene/cameras.py:70
  * znear = 0.01, zfar = 100                        /root/reference/scene/cameras.py:61-62
"""
import math

import numpy as np
import torch

ZNEAR, ZFAR = 0.01, 100.0
N_FRAMES = 300                      # frames per HOI4D video, /root/reference/README.md:36
SCENE_CENTRE = np.array([0.0, 0.0, 6.0])


def projection_matrix(znear, zfar, fovx, fovy):
    """Perspective matrix P (column-vector form; the renderer consumes P^T)."""
    tx, ty = math.tan(fovx / 2), math.tan(fovy / 2)
    top, right = ty * znear, tx * znear
    bottom, left = -top, -right
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def world_to_view(R, t):
    """4x4 world->view from the reference's (R, T) pair: rotation block is R^T, translation t."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = np.asarray(R).transpose()
    Rt[:3, 3] = np.asarray(t)
    Rt[3, 3] = 1.0
    return np.float32(Rt)


class SynthCamera:
    """Duck-typed stand-in for scene.cameras.Camera: the seven attributes render() reads
    (/root/reference/gaussian_renderer/__init__.py:35-48)."""

    def __init__(self, w2v, H, W, fovx, fovy, device="cpu"):
        self.image_height, self.image_width = int(H), int(W)
        self.FoVx, self.FoVy = float(fovx), float(fovy)
        self.world_view_transform = torch.tensor(np.float32(w2v)).transpose(0, 1).contiguous().to(device)
        self.projection_matrix = projection_matrix(ZNEAR, ZFAR, fovx, fovy).transpose(0, 1).to(device)
        self.full_proj_transform = (self.world_view_transform.unsqueeze(0)
                                    .bmm(self.projection_matrix.unsqueeze(0))).squeeze(0).contiguous()
        self.camera_center = self.world_view_transform.inverse()[3, :3].contiguous()
        # the three tensors render() reads, also as views of one block so that a captured step refreshes them in one copy
        self.packed = torch.cat([self.world_view_transform.reshape(-1), self.full_proj_transform.reshape(-1), self.camera_center])
        self.world_view_transform = self.packed[0:16].view(4, 4)
        self.full_proj_transform = self.packed[16:32].view(4, 4)
        self.camera_center = self.packed[32:35]


def fov_pair(H, W, fovx_deg=60.0):
    fovx = math.radians(fovx_deg)
    return fovx, 2.0 * math.atan(math.tan(fovx / 2) * H / W)


def make_camera(k, H, W, device="cpu", fovx_deg=60.0):
    """Frame k of the synthetic 300-frame orbit: yaw 10 deg*sin, pitch 5 deg*cos about (0,0,6)."""
    fovx, fovy = fov_pair(H, W, fovx_deg)
    ph = 2.0 * math.pi * (k % N_FRAMES) / N_FRAMES
    yaw, pitch = math.radians(10.0) * math.sin(ph), math.radians(5.0) * (math.cos(ph) - 1.0)
    cy, sy, cp, sp = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    Rot = Rx @ Ry
    w2v = np.eye(4)
    w2v[:3, :3] = Rot
    w2v[:3, 3] = SCENE_CENTRE - Rot @ SCENE_CENTRE      # frame 0 (yaw = pitch = 0) is the world frame
    return SynthCamera(w2v, H, W, fovx, fovy, device)


def make_scene(N, H, W, seed=0, sh_degree=0, fovx_deg=60.0):
    """S(N,H,W,seed): raw (pre-activation) parameters as float32 numpy arrays, the layout
    scene.gaussian_model.GaussianModel stores (/root/reference/scene/gaussian_model.py:125-165)."""
    rng = np.random.default_rng(seed)
    fovx, _ = fov_pair(H, W, fovx_deg)
    a = 0.9 * 6.0 * math.tan(fovx / 2)
    xyz = np.stack([rng.uniform(-a, a, N), rng.uniform(-a * H / W, a * H / W, N), rng.uniform(2.0, 10.0, N)], 1)
    log_scale = rng.normal(math.log(0.02), 0.5, (N, 3))
    quat = rng.normal(0.0, 1.0, (N, 4))
    quat /= np.linalg.norm(quat, axis=1, keepdims=True)
    opacity_logit = rng.normal(0.0, 1.5, (N, 1))
    M = (sh_degree + 1) ** 2
    features = rng.normal(0.0, 1.0, (N, M, 3))
    if M > 1:
        features[:, 1:] *= 0.2
    f32 = np.float32
    return dict(xyz=xyz.astype(f32), log_scale=log_scale.astype(f32), quat=quat.astype(f32),
                opacity_logit=opacity_logit.astype(f32), features=features.astype(f32))


def perturb_student(scene, seed=1):
    """Student = teacher with xyz += N(0, 0.01^2), f_dc += N(0, 0.1^2)."""
    rng = np.random.default_rng(1000 + seed)
    out = {k: v.copy() for k, v in scene.items()}
    out["xyz"] += rng.normal(0, 0.01, out["xyz"].shape).astype(np.float32)
    out["features"][:, :1] += rng.normal(0, 0.1, out["features"][:, :1].shape).astype(np.float32)
    return out


class SynthGaussians:
    """Minimal parameter holder with the getters render() reads -- a stand-in for
    scene.gaussian_model.GaussianModel (/root/reference/scene/gaussian_model.py:125-171) over a synthetic scene.
    Raw parameters are leaf tensors; activations match the reference (exp, sigmoid, normalize)."""

    def __init__(self, scene, device="cpu", sh_degree=0, requires_grad=True, fused=True):
        from . import covariance as _cov
        self._cov = _cov
        self.fused = fused            # use the fused HIP covariance producer (row f-1) when on a HIP device
        self.rotate_in_rasterizer = True   # render(rot_cov=True): hand the raw parameters + object rotation to the rasterizer (else: fused producer)
        t = lambda a: torch.tensor(a, device=device).requires_grad_(requires_grad)
        self._xyz = t(scene["xyz"])
        self._features_dc = t(scene["features"][:, :1].copy())
        self._features_rest = t(scene["features"][:, 1:].copy())
        self._scaling = t(scene["log_scale"])
        self._rotation = t(scene["quat"])
        self._opacity = t(scene["opacity_logit"])
        n = scene["xyz"].shape[0]
        self._label = torch.zeros((n, 1), device=device).requires_grad_(requires_grad)
        self._is_object = torch.zeros((n, 1), device=device)
        self._generation = torch.zeros((n, 1), dtype=torch.int, device=device)
        self.max_radii2D = torch.zeros((n,), device=device)
        self.xyz_gradient_accum = torch.zeros((n, 1), device=device)
        self.denom = torch.zeros((n, 1), device=device)
        self.percent_dense = 0.01
        self.optimizer = None
        self.scaling_activation, self.opacity_activation, self.rotation_activation = torch.exp, torch.sigmoid, torch.nn.functional.normalize
        self.active_sh_degree = sh_degree
        self.max_sh_degree = sh_degree
        self.trainable_object_move = None

    def parameters(self):
        return [self._xyz, self._features_dc, self._features_rest, self._scaling, self._rotation, self._opacity]

    def training_setup(self, optimizer_cls=None, percent_dense=0.01, **kw):
        """One named group per parameter with the reference's default learning rates
        (/root/reference/scene/gaussian_model.py:180-198, arguments/__init__.py OptimizationParams)."""
        self.percent_dense = percent_dense
        self._label = self._label.detach().requires_grad_(True)
        groups = [{"params": [self._xyz], "lr": 1.6e-4, "name": "xyz"}, {"params": [self._features_dc], "lr": 2.5e-3, "name": "f_dc"},
                  {"params": [self._features_rest], "lr": 2.5e-3 / 20.0, "name": "f_rest"},
                  {"params": [self._opacity], "lr": 0.05, "name": "opacity"}, {"params": [self._scaling], "lr": 5e-3, "name": "scaling"},
                  {"params": [self._rotation], "lr": 1e-3, "name": "rotation"}, {"params": [self._label], "lr": 0.0, "name": "label"}]
        if optimizer_cls is None:
            from .optim import FusedAdam as optimizer_cls
        self.optimizer = optimizer_cls(groups, lr=0.0, eps=1e-15, **kw)
        return self.optimizer

    @property
    def get_xyz(self): return self._xyz
    # (the activations are attributes, as setup_functions leaves them on the reference's model, gaussian_model.py:36-44: what
    # adapter.attach() replaces)
    @property
    def get_scaling(self): return self.scaling_activation(self._scaling)
    @property
    def get_rotation(self): return self.rotation_activation(self._rotation)
    @property
    def get_opacity(self): return self.opacity_activation(self._opacity)
    @property
    def get_features(self):
        if self._features_rest.shape[1] == 0:                # SH degree 0: nothing to concatenate (torch.cat would still copy)
            return self._features_dc
        f = torch.cat((self._features_dc, self._features_rest), dim=1)
        if getattr(self, "_egs_tag_features", False):        # adapter.attach(provenance=True); patching.install() does this for the reference's class
            from .provenance import tag_features
            tag_features(f, self._features_dc, self._features_rest)
        return f
    @property
    def get_label(self): return self._label
    @property
    def get_is_object(self): return self._is_object

    def get_covariance(self, scaling_modifier=1):
        if self.fused and self._xyz.is_cuda:
            from . import fused
            return fused.covariance_from_log_scaling(self._scaling, scaling_modifier, self._rotation)
        return self._cov.covariance_from_scaling_rotation(self.get_scaling, scaling_modifier, self._rotation)

    def get_features_split(self):
        """Optional hook render() looks for: the colour coefficients as the two stored parameters (HIP devices only)."""
        if self.fused and self._xyz.is_cuda:
            return self._features_dc, self._features_rest
        return None

    def get_raw_parameters(self):
        """Optional hook render() looks for: (log-scales, raw quaternions, opacity logits) for the rasterizer's raw-parameter
        mode (HIP devices only; None otherwise)."""
        if self.fused and self._xyz.is_cuda:
            return self._scaling, self._rotation, self._opacity
        return None

    def get_raw_parameters_rotated(self, accum_R, which_object, during_training):
        """Optional hook render(rot_cov=True) looks for: (log-scales, raw quaternions, opacity logits, (M, selected, row-0 gradient
        multiplier)) for the rasterizer's object_rotation mode -- the `fine_all` call shape with no covariance tensor at all.  None
        when the object rotation is being trained (its gradient needs the covariance path) or off a HIP device."""
        if not (self.fused and self._xyz.is_cuda and self.rotate_in_rasterizer):
            return None
        if during_training and self.trainable_object_move is not None:
            return None
        if accum_R is not None and accum_R.requires_grad:
            return None
        M = torch.eye(3, device=self._xyz.device) if accum_R is None else accum_R.to(self._xyz.device, torch.float32)
        sel, mult = self._object_selection(which_object)
        return self._scaling, self._rotation, self._opacity, (M, sel, mult)

    def get_covariance_and_opacity(self, scaling_modifier=1):
        """Optional hook render() looks for: covariance and activated opacity from one fused launch (HIP devices only)."""
        if self.fused and self._xyz.is_cuda:
            from . import fused
            return fused.covariance_and_opacity(self._scaling, scaling_modifier, self._rotation, self._opacity)
        return self.get_covariance(scaling_modifier), self.get_opacity

    def _object_selection(self, which_object):
        """fused.object_selection for this model, kept until `_is_object` is replaced or edited (it depends on nothing else)."""
        from . import fused
        n_live = getattr(self, "n_active", None)                # capacity-sized model: only the live rows are Gaussians
        key = (which_object, self._is_object.data_ptr(), self._is_object._version, tuple(self._is_object.shape), n_live)
        if getattr(self, "_sel_key", None) != key:
            self._sel_cache = fused.object_selection(self._is_object, which_object, self._xyz.shape[0], n_live)
            self._sel_key = key
        return self._sel_cache

    def get_rotated_covariance_and_opacity(self, accum_R, which_object, during_training, scaling_modifier=1):
        """Optional hook render(rot_cov=True) looks for: object-rotated covariance and activated opacity from the raw parameters in one
        launch each way (HIP devices only; same values as get_rotated_covariance + get_opacity)."""
        tom = self.trainable_object_move if during_training else None
        if self.fused and self._xyz.is_cuda:
            from . import fused
            return fused.rotated_covariance_from_scaling_rotation(
                self._scaling, scaling_modifier, self._rotation, accum_R, self._is_object, which_object,
                None if tom is None else tom.rot_L(torch.eye(3, device=self._xyz.device)), scaling_is_log=True, selection=self._object_selection(which_object),
                opacity_raw=self._opacity)
        return self.get_rotated_covariance(accum_R, which_object, during_training, scaling_modifier), self.get_opacity

    def get_rotated_covariance(self, accum_R, which_object, during_training, scaling_modifier=1):
        tom = self.trainable_object_move if during_training else None
        if self.fused and self._xyz.is_cuda:
            from . import fused
            return fused.rotated_covariance_from_scaling_rotation(
                self.get_scaling, scaling_modifier, self._rotation, accum_R, self._is_object, which_object,
                None if tom is None else tom.rot_L(torch.eye(3, device=self._xyz.device)), selection=self._object_selection(which_object))
        return self._cov.rotated_covariance_from_scaling_rotation(
            self.get_scaling, scaling_modifier, self._rotation, accum_R, self._is_object, which_object,
            None if tom is None else tom.rot_L)


class Pipe:
    """PipelineParams as every stage runs it: /root/reference/arguments/__init__.py:66-68, /root/reference/train.py:49."""
    convert_SHs_python = False
    compute_cov3D_python = True
    debug = False
