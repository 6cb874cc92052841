"""Where does float32 lose the recorded parity miss (fuzz seed 777123 draw 3869: Gaussian 13937, a 423 px splat centred 270 px outside a
3 x 105 image, dL/dmeans3D 6.7e-4 of the array's maximum away from the oracle)?  CPU only: the C oracle in float32 and float64, then the
covariance part of the preprocess backward of that ONE Gaussian in numpy with the precision chosen per stage (projection recompute /
conic -> cov2D / everything after).  Result (profiles/r6_chain_precision.txt): the blend's sums are fine (float64 chain on float32 sums:
2.5e-5); the float32 CHAIN on exact sums is 7.0e-4 off; with only the three conic -> cov2D lines in float64 it is 2.4e-6 off.
Usage: python tools/dev/chain_precision.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.common import make_inputs, seeded_grads
from oracle.oracle import Oracle
N,H,W,seed,deg,mode,frame,smul,oshift = (70000, 105, 3, 21, 1, "sh_sr", 195, 4.0, 2.0)
d = make_inputs(N, H, W, seed, deg, mode, frame=frame, scale_mul=smul, opacity_shift=oshift)
o32 = Oracle(np.float32, nthreads=8); o64 = Oracle(np.float64, nthreads=8)
st = o32.forward(**d)
st64 = o64.forward(**{k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in d.items()})
grads = seeded_grads(H, W, 7)
gb = o32.backward(st, *grads); gb64 = o64.backward(st64, *[g.double() for g in grads])
g = 13937
print("radius", st["radii"][g], "xy", st["xy"][g], "conic_op", st["conic_opacity"][g], "tiles", st["tiles_touched"][g])
for k in ("dL_dmean2D","dL_dconic","dL_dopacity","dL_dmeans3D","dL_dscale","dL_drot"):
    a, b = np.asarray(gb[k][g], np.float64), gb64[k][g]
    print(k, "f32", a, "f64", b, "rel-own", np.abs(a-b).max()/np.abs(b).max(), "scale(arr max)", np.abs(gb64[k]).max())
# list lengths
r = st["ranges"]; print("list lengths", (r[:,1]-r[:,0]))
V = st["viewmatrix"]; c6 = st["cov3D"][g]; p = st["means3D"][g]
fx = np.float32(W)/(np.float32(2)*np.float32(st["tanfovx"])); fy = np.float32(H)/(np.float32(2)*np.float32(st["tanfovy"]))
def chain(gA, gB, gC, T_proj, T_conic, T_rest, wform=None):
    # T_proj: dtype of the projection recompute (t, J, m, S, a b c); T_conic: dtype of conic->cov2D; T_rest: the rest
    t_ = T_proj
    Vp = V.astype(t_); pp = p.astype(t_); cc = c6.astype(t_)
    t = np.array([Vp[0]*pp[0]+Vp[4]*pp[1]+Vp[8]*pp[2]+Vp[12], Vp[1]*pp[0]+Vp[5]*pp[1]+Vp[9]*pp[2]+Vp[13], Vp[2]*pp[0]+Vp[6]*pp[1]+Vp[10]*pp[2]+Vp[14]], dtype=t_)
    limx = t_(1.3)*t_(st["tanfovx"]); limy = t_(1.3)*t_(st["tanfovy"])
    txtz, tytz = t[0]/t[2], t[1]/t[2]
    tx = min(limx, max(-limx, txtz))*t[2]; ty = min(limy, max(-limy, tytz))*t[2]
    xm = 0.0 if (txtz < -limx or txtz > limx) else 1.0; ym = 0.0 if (tytz < -limy or tytz > limy) else 1.0
    j00 = t_(fx)/t[2]; j02 = -(t_(fx)*tx)/(t[2]*t[2]); j11 = t_(fy)/t[2]; j12 = -(t_(fy)*ty)/(t[2]*t[2])
    m0 = np.array([j00*Vp[4*k]+j02*Vp[4*k+2] for k in range(3)], dtype=t_); m1 = np.array([j11*Vp[4*k+1]+j12*Vp[4*k+2] for k in range(3)], dtype=t_)
    C3 = np.array([[cc[0],cc[1],cc[2]],[cc[1],cc[3],cc[4]],[cc[2],cc[4],cc[5]]], dtype=t_)
    S0 = C3@m0; S1 = C3@m1
    a = m0@S0 + t_(0.3); b = m0@S1; c = m1@S1 + t_(0.3)
    u = T_conic
    a_, b_, c_ = u(a), u(b), u(c); gA_, gB_, gC_ = u(gA), u(gB), u(gC)
    denom = a_*c_ - b_*b_
    d2inv = u(1)/(denom*denom + u(1e-7))
    if wform is None:
        dL_da = d2inv*(-c_*c_*gA_ + u(2)*b_*c_*gB_ + (denom - a_*c_)*gC_)
        dL_dc = d2inv*(-a_*a_*gC_ + u(2)*a_*b_*gB_ + (denom - a_*c_)*gA_)
        dL_db = d2inv*u(2)*(b_*c_*gA_ - (denom + u(2)*b_*b_)*gB_ + a_*b_*gC_)
    else:
        dL_da, dL_db, dL_dc = (u(x) for x in wform)
    r = T_rest
    dL_da, dL_db, dL_dc = r(dL_da), r(dL_db), r(dL_dc)
    S0r, S1r, Vr = S0.astype(r), S1.astype(r), V.astype(r)
    gm0 = r(2)*S0r*dL_da + S1r*dL_db; gm1 = r(2)*S1r*dL_dc + S0r*dL_db
    gJ00 = sum(Vr[4*k]*gm0[k] for k in range(3)); gJ02 = sum(Vr[4*k+2]*gm0[k] for k in range(3))
    gJ11 = sum(Vr[4*k+1]*gm1[k] for k in range(3)); gJ12 = sum(Vr[4*k+2]*gm1[k] for k in range(3))
    tz = r(1)/r(t[2]); tz2 = tz*tz; tz3 = tz2*tz
    gtx = r(xm)*-r(fx)*tz2*gJ02; gty = r(ym)*-r(fy)*tz2*gJ12
    gtz = -r(fx)*tz2*gJ00 - r(fy)*tz2*gJ11 + (r(2)*r(fx)*r(tx))*tz3*gJ02 + (r(2)*r(fy)*r(ty))*tz3*gJ12
    gmean = np.array([Vr[4*k]*gtx + Vr[4*k+1]*gty + Vr[4*k+2]*gtz for k in range(3)], dtype=r)
    return gmean, (dL_da, dL_db, dL_dc), (a, b, c, denom)
sc = np.abs(gb64["dL_dmeans3D"]).max()
gc64 = gb64["dL_dconic"][g]
f32, f64 = np.float32, np.float64
ref, cov_ref, abc = chain(gc64[0], gc64[1], gc64[3], f64, f64, f64)
print("abc denom f64", abc)
print("f64 cov-part of gmean", ref, "cov2D grads", cov_ref)
for name, Ts in (("all f32", (f32,f32,f32)), ("proj64", (f64,f32,f32)), ("conic64", (f32,f64,f32)), ("rest64", (f32,f32,f64)), ("proj+conic 64", (f64,f64,f32)), ("conic+rest 64", (f32,f64,f64))):
    out, cg, abc2 = chain(f32(gc64[0]), f32(gc64[1]), f32(gc64[3]), *Ts)
    print(f"{name:16s} err/arrmax {np.abs(out.astype(f64)-ref).max()/sc:.2e}  cov2D grads rel {[float(abs(f64(x)-y)/abs(y)) for x,y in zip(cg,cov_ref)]}  denom {abc2[3]}")
