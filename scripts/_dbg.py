import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, oracle
from conftest import make_test_image
from visual_slam_amd import _lib
c=_lib.Context(0)
img=make_test_image(72,96,101)
xy=oracle.c.detect(img)
a=c.orient(img,xy); e=oracle.c.orient(img,xy)
bad=np.nonzero(a.view(np.uint32)!=e.view(np.uint32))[0]
print("n",len(xy),"bad",len(bad), bad[:20])
for i in bad[:10]:
    print(i, xy[i], a[i], e[i], oracle.c.moments(img,*xy[i]))
d,v=c.describe(img,xy,e); ed,ev=oracle.c.describe(img,xy,e)
print("desc bad rows", int((d!=ed).any(axis=1).sum()), "of", len(ed), np.array_equal(v,ev))
