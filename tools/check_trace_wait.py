import os, sys
sys.path.insert(0, "tools")
import _tuning
import torch
from toroidal_ray_tracing_amd import camera
from toroidal_ray_tracing_amd.tracer import Tracer
dev = torch.device("cuda:0"); n = 1 << 20
gen = torch.Generator(device=dev).manual_seed(1)
o = torch.rand(n, 3, device=dev, generator=gen) * 8 - 4
tgt = torch.randn(n, 3, device=dev, generator=gen)
tgt = tgt / tgt.norm(dim=1, keepdim=True) * (torch.rand(n, 1, device=dev, generator=gen) * 1.2)
d = tgt - o; d = d / d.norm(dim=1, keepdim=True)
rays = [o[:, k].contiguous() for k in range(3)] + [d[:, k].contiguous() for k in range(3)]
res = {}
for sc_name, sc in (("single", camera.single_torus_scene()), ("nested", camera.nested_tori_scene())):
    for th in ("1", "16", "40"):
        os.environ["TRT_TRACE_VARIANT"] = th
        tr = Tracer(0)
        out = {k: torch.empty(n, device=dev) for k in ("t", "px", "py", "pz", "nx", "ny", "nz")}
        tr.trace_dev(sc, [a.data_ptr() for a in rays], n, {k: v.data_ptr() for k, v in out.items()})
        torch.cuda.synchronize()
        res[(sc_name, th)] = torch.stack([out[k] for k in out]).view(torch.int32).clone()
        tr.close()
    print(sc_name, "bit-identical across thresholds:", all(torch.equal(res[(sc_name, "1")], res[(sc_name, th)]) for th in ("16", "40")))
