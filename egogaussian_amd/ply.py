"""PLY hand-off between training stages, with the reference's three extra columns.

save_ply / load_ply mirror /root/reference/scene/gaussian_model.py:375-480 (column list :340-358): one `vertex` element,
every property `float` (little-endian binary, what plyfile writes by default), in the order
    x y z  nx ny nz  f_dc_*  f_rest_*  opacity  scale_*  rot_*  label  generation  is_object
with the SH coefficients channel-major (the reference transposes [N, K, 3] to [N, 3, K] before flattening).  Files written
by upstream 3DGS (no label / generation / is_object columns) load with the reference's defaults.  numpy only -- `plyfile`
is not needed.
"""
import os

import numpy as np
import torch


def attribute_names(n_dc, n_rest):
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(n_dc)] + [f"f_rest_{i}" for i in range(n_rest)]
    names += ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)]
    return names + ["label", "generation", "is_object"]


def vertex_table(gaussians):
    """-> (names, float32 [N, len(names)]) exactly as the reference assembles it (the live rows of a capacity-sized model)."""
    n = getattr(gaussians, "n_active", None)                      # capacity.CapacityGaussians: rows beyond it are not part of the model
    c = lambda t: t.detach()[:n].cpu().numpy().astype(np.float32)
    xyz = c(gaussians._xyz)
    f_dc = c(gaussians._features_dc.detach()[:n].transpose(1, 2).flatten(start_dim=1).contiguous())
    f_rest = c(gaussians._features_rest.detach()[:n].transpose(1, 2).flatten(start_dim=1).contiguous())
    cols = [xyz, np.zeros_like(xyz), f_dc, f_rest, c(gaussians._opacity), c(gaussians._scaling), c(gaussians._rotation),
            c(gaussians._label), c(gaussians._generation), c(gaussians._is_object)]
    cols = [a.reshape(xyz.shape[0], -1) for a in cols]
    return attribute_names(f_dc.shape[1], f_rest.shape[1]), np.concatenate(cols, axis=1)


def write_table(path, names, table):
    table = np.ascontiguousarray(table, dtype="<f4")
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % table.shape[0]
    header += "".join(f"property float {n}\n" for n in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(table.tobytes())


def save_ply(gaussians, path):
    names, table = vertex_table(gaussians)
    write_table(path, names, table)


_PLY_TYPES = {"float": "f4", "float32": "f4", "double": "f8", "float64": "f8", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4",
              "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2", "char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1"}


def read_table(path):
    """-> dict name -> float64 column of the first element (binary little/big endian or ascii, scalar properties only)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, in_first = None, None, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated header")
            tok = line.decode("ascii").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                if count is None:
                    count, in_first = int(tok[2]), True
                else:
                    in_first = False
            elif tok[0] == "property" and in_first:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties are not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt == "ascii":
            data = np.loadtxt(f, max_rows=count, ndmin=2)
            return {n: data[:, i].astype(np.float64) for i, (n, _) in enumerate(props)}
        order = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(n, order + t) for n, t in props])
        rec = np.frombuffer(f.read(count * dt.itemsize), dtype=dt, count=count)
        return {n: rec[n].astype(np.float64) for n, _ in props}


def load_ply(gaussians, path, train_params=True, is_object=False, force_bg=False, device="cuda"):
    """Fills `gaussians` like GaussianModel.load_ply (:398-480): parameters (nn.Parameter when train_params), label (default 0.01),
    generation (default 0), is_object (default from the flag; zeros with force_bg), max_radii2D zeros, active SH degree = max."""
    cols = read_table(path)
    n = cols["x"].shape[0]
    xyz = np.stack([cols["x"], cols["y"], cols["z"]], 1)
    f_dc = np.stack([cols["f_dc_0"], cols["f_dc_1"], cols["f_dc_2"]], 1)[:, :, None]                 # [N, 3, 1]
    rest_names = sorted([k for k in cols if k.startswith("f_rest_")], key=lambda k: int(k.split("_")[-1]))
    K1 = (gaussians.max_sh_degree + 1) ** 2 - 1
    if len(rest_names) != 3 * K1:
        raise ValueError(f"{path}: {len(rest_names)} f_rest columns, the model's SH degree needs {3 * K1}")
    f_rest = np.stack([cols[k] for k in rest_names], 1).reshape(n, 3, K1) if K1 else np.zeros((n, 3, 0))
    scale = np.stack([cols[k] for k in sorted([k for k in cols if k.startswith("scale_")], key=lambda k: int(k.split("_")[-1]))], 1)
    rot = np.stack([cols[k] for k in sorted([k for k in cols if k.startswith("rot")], key=lambda k: int(k.split("_")[-1]))], 1)
    label = cols["label"][:, None] if "label" in cols else np.full((n, 1), 0.01)
    gen = cols["generation"].astype(np.int64)[:, None] if "generation" in cols else np.zeros((n, 1), np.int64)
    if "is_object" in cols:
        obj = cols["is_object"].astype(np.int64)[:, None]
    else:
        obj = np.ones((n, 1), np.int64) if is_object else np.zeros((n, 1), np.int64)
    if force_bg:
        obj = np.zeros((n, 1), np.int64)
    t = lambda a: torch.tensor(a, dtype=torch.float, device=device)
    vals = {"_xyz": t(xyz), "_features_dc": t(f_dc).transpose(1, 2).contiguous(), "_features_rest": t(f_rest).transpose(1, 2).contiguous(),
            "_opacity": t(cols["opacity"][:, None]), "_scaling": t(scale), "_rotation": t(rot), "_label": t(label)}
    for k, v in vals.items():
        setattr(gaussians, k, torch.nn.Parameter(v.requires_grad_(True)) if train_params else v)
    gaussians._generation = torch.tensor(gen, dtype=torch.int, device=device)
    gaussians._is_object = torch.tensor(obj, dtype=torch.int, device=device)
    gaussians.max_radii2D = torch.zeros((n,), device=device)
    gaussians.active_sh_degree = gaussians.max_sh_degree
    return gaussians
