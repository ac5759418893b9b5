import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cse168-raytracer_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import miro_amd, pyoracle as po
from miro_amd import binding, scenes
d = scenes.SCENES["teapot"]
sc = miro_amd.Scene(0); scenes.populate(sc, d); sc.build(4)
W=H=128
dr = torch.empty((W*H,8), dtype=torch.float32, device="cuda")
sc.gen_eye_rays(binding.make_camera(d["eye"], d["lookat"], d["up"], d["fov"]), W, H, dr)
got = dr.cpu().numpy()
want = po.eye_rays(po.make_camera(d["eye"], d["lookat"], d["up"], d["fov"]), W, H).view(np.float32).reshape(-1,8)
bad = np.nonzero((got.view(np.uint32) != want.view(np.uint32)).any(1))[0]
print("mismatching rays", len(bad), "of", len(got))
for i in bad[:5]:
    print(i, got[i], want[i], (got[i].view(np.uint32).astype(np.int64) - want[i].view(np.uint32).astype(np.int64)))
print("columns differing:", (got.view(np.uint32) != want.view(np.uint32)).sum(0))
