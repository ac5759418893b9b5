"""A/B check of trace-kernel variants at full bench size: every variant must return bit-identical hit buffers
(primary batch, shadow batch, incoherent batch).  Each variant runs in its own process (the variant is read
once from MIRO_TRACE_VARIANT); a position-weighted 64-bit checksum of the hit bits is compared.

NEEDS A DEVELOPMENT BUILD of the library (make -C cse168-raytracer_amd clean && make -C cse168-raytracer_amd DEV=1): the
shipped library does not read MIRO_TRACE_VARIANT.  tools/ab_modes.py compares the shipped control-flow modes in one process.

usage: python tools/ab_variants.py [--spp 16] [--variants 0,3,7,9,11,q]   (q = the default trace with exact quotients)
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def checksum(t):
    import torch
    v = t.view(torch.int32).reshape(-1).to(torch.int64)
    w = (torch.arange(v.numel(), device=v.device, dtype=torch.int64) % 1000003) + 1
    return int((v * w).sum().item())


def child(a):
    sys.path.insert(0, os.path.join(ROOT, "cse168-raytracer_amd"))
    import numpy as np
    import torch
    import miro_amd
    from miro_amd import frame as mframe, scenes
    out = {}
    # variant "q" = the default trace (exact quotients); every other variant is a form of the product kernel
    flags = 0 if os.environ.get("MIRO_AB_QUOTIENTS") == "1" else miro_amd.MR_MATH_PRODUCT
    for name in a.scenes.split(","):
        sc = miro_amd.Scene(0)
        scenes.populate(sc, name)
        sc.build(4)
        fr = mframe.FrameRenderer(sc, name, a.w, a.h, spp=a.spp, flags=flags)
        fr.generate()
        fr.trace_primary()
        fr.make_shadow_rays()
        fr.trace_shadow()
        torch.cuda.synchronize()
        n_p, n_s = fr.ray_counts()
        out[name + ".primary"] = checksum(fr.d_hits)
        # the compacted order of the shadow batch depends on which workgroup reserved its chunk first (one atomic
        # per chunk): bring the shadow hits back into primary-ray order before comparing
        canon = torch.zeros((n_p, 4), dtype=torch.float32, device="cuda")
        canon[fr.d_src[:n_s].to(torch.int64)] = fr.d_shadow_hits[:n_s]
        out[name + ".shadow"] = checksum(canon)
        v = sc.arrays()[0]
        lo, hi = np.maximum(v.min(0), -20), np.minimum(v.max(0), 20)
        g = torch.Generator(device="cuda").manual_seed(7)
        m = a.incoherent
        r = torch.zeros((m, 8), device="cuda")
        r[:, 0:3] = torch.rand((m, 3), device="cuda", generator=g) * torch.tensor(hi - lo, device="cuda") + torch.tensor(lo, device="cuda")
        dd = torch.randn((m, 3), device="cuda", generator=g)
        r[:, 4:7] = dd / dd.norm(dim=1, keepdim=True)
        r[:, 7] = 1e12
        hh = torch.empty((m, 4), device="cuda")
        sc.trace_device(r, m, hh, flags)
        torch.cuda.synchronize()
        out[name + ".incoherent"] = checksum(hh)
        out[name + ".rays"] = [n_p, n_s, m]
    print("AB_RESULT " + json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--variants", default="0,3,7,9,11,q")
    ap.add_argument("--scenes", default="sponza,bunny,teapot")
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--h", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--incoherent", type=int, default=16000000)
    a = ap.parse_args()
    if a.child:
        return child(a)
    results = {}
    for v in a.variants.split(","):
        env = dict(os.environ, MIRO_TRACE_VARIANT="11", MIRO_AB_QUOTIENTS="1") if v == "q" else dict(os.environ, MIRO_TRACE_VARIANT=v)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--scenes", a.scenes, "--w", str(a.w),
                            "--h", str(a.h), "--spp", str(a.spp), "--incoherent", str(a.incoherent)],
                           env=env, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("AB_RESULT ")]
        if p.returncode != 0 or not line:
            print("variant %s failed:\n%s\n%s" % (v, p.stdout[-2000:], p.stderr[-2000:]))
            sys.exit(1)
        results[v] = json.loads(line[0][len("AB_RESULT "):])
    base = results[a.variants.split(",")[0]]
    ok = True
    for v, r in results.items():
        same = r == base
        ok &= same
        print("variant %s: %s" % (v, "identical" if same else "DIFFERENT: %s" % {k: (r[k], base[k]) for k in r if r[k] != base[k]}))
    print("rays compared per variant:", {k: v for k, v in base.items() if k.endswith(".rays")})
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
