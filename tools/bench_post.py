import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _tuning  # noqa: E402  (loads the -DTRT_TUNING build, see _tuning.py)
import torch
from toroidal_ray_tracing_amd.tracer import Tracer
dev = torch.device("cuda:0"); tr = Tracer(0); s = torch.cuda.current_stream()
n = 4096 * 4096
img = torch.rand(n, 4, device=dev); img[:, 3] = 1.0
o8 = torch.empty(n, 4, dtype=torch.uint8, device=dev); of = torch.empty(n, 4, device=dev)
def t(fn, reps=20, rounds=5):
    out = []
    for r in range(rounds + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps): fn()
        e1.record(s); torch.cuda.synchronize()
        if r: out.append(e0.elapsed_time(e1) / reps)
    return statistics.median(out)
for b in ([int(x) for x in sys.argv[1:]] or (4, 8, 16, 32, 64, 128, 256)):   # 0 = the launcher's default
    os.environ["TRT_POST_BLOCKS_PER_CU"] = str(b); _tuning.reload(tr)
    a = t(lambda: tr.post_dev(img.data_ptr(), n, 0, o8.data_ptr(), stream=s.cuda_stream))
    c = t(lambda: tr.post_dev(img.data_ptr(), n, of.data_ptr(), 0, stream=s.cuda_stream))
    print(f"blocks/CU {b:4d}: ->u8 {a:.4f} ms ({20*n/a/1e6:.0f} GB/s)   ->f32 {c:.4f} ms ({32*n/c/1e6:.0f} GB/s)")
cp = t(lambda: of.copy_(img))
print(f"torch copy 268MB->268MB {cp:.4f} ms ({32*n/cp/1e6:.0f} GB/s)")
