import numpy as np, sys
a=np.load(sys.argv[1])
t0=a[:,0]-a[:,0].min(); t1=a[:,1]-a[:,0].min()
xcc=a[:,2]>>16
out=[]
for x in range(8):
    m=xcc==x
    simd=((a[m,2]>>8)&0xff)*16+((a[m,2]>>4)&3)
    ids,inv=np.unique(simd,return_inverse=True)
    endt=np.zeros(len(ids)); np.maximum.at(endt,inv,t1[m]); vis=np.zeros(len(ids)); np.add.at(vis,inv,a[m,6])
    out.append((round(endt.mean()/100,1), int(vis.mean())))
print(" per-XCD mean SIMD end (us), visits/SIMD:", out)
