import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from tomo_tv_amd.engine import tomoengine
for Nx in (192, 320, 64, 128):
    N,P=64,9
    ang=np.linspace(-68,71,P)
    b=np.random.default_rng(3).standard_normal((Nx,P*N)).astype(np.float32)
    vols={}
    for form in ("list","tile","pixel"):
        t=tomoengine(Nx,N,ang*np.pi/180)
        t.set_option("fp_tile",0); t.set_option("bp_tile",0 if form=="pixel" else 1); t.set_option("bp_list",1 if form=="list" else 0)
        t.set_tilt_series(b); t.SIRT(int(sys.argv[1]) if len(sys.argv)>1 else 2); vols[form]=t.get_volume()
    d=vols["tile"]-vols["pixel"]
    bad=np.argwhere(d!=0)
    print(Nx, "tile==pixel", np.array_equal(vols["tile"],vols["pixel"]), "list==pixel", np.array_equal(vols["list"],vols["pixel"]), len(bad), bad[:5].tolist(), np.abs(d).max())
