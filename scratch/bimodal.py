"""Is the run-to-run spread of k_mf_fused_tv (slab 4096 x 512 x 16: 1.11 or 1.27 ms; full size 8.25 or 8.7 ms for one binary on one box) tied to
where the buffers land? One process, several handles, dummy allocations of different sizes in between; kernel time of each."""
import math, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ma_amd import engine as E, synth

m, n, k = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 512, 16
dev = torch.device("cuda", 0)
imgs = synth.dynamic_test_image(m, n, k, device=dev).transpose(1, 2).contiguous().cpu().numpy().reshape(-1)
u0 = synth.smooth_displacement(m, n, k, amp=2.0, device=dev).cpu().numpy().reshape(-1)
x0 = np.concatenate([u0, np.zeros(k * m * n)])
torch.cuda.empty_cache()
pads = [0, 0, 1 << 20, 3 << 20, 64 << 10, 0, 17 << 20, 2 << 30, 0, 0]
fpads = [int(v) for v in os.environ.get("FIELD_PADS", "").split(",") if v]
if fpads:
    pads = [0] * len(fpads)
keep = []
for it, pad in enumerate(pads):
    if fpads:
        os.environ["CP_TUNE_FIELD_PAD"] = str(fpads[it])
    if pad:
        keep.append(torch.empty(pad, dtype=torch.uint8, device=dev))
    eng = E.Engine(E.MODEL_MF_NN_TV, m, n, k=k, bc="neumann", h=(1.0, 1.0), omega=[0.0, float(m), 0.0, float(n)], mu=0.2)
    eng.set_images(imgs)
    eng.set_primal(x0)
    eng.linearize()
    eng.set_nu(0.9 * float(np.sum(eng.singular_values(0))))
    tau = 0.25
    eng.iterate_begin(1.0, tau, tau, 64)
    eng.iterate_steps(4)
    eng.sync()
    eng.kernel_time_ms("dual")
    out = []
    for rep in range(3):
        eng.iterate_steps(20)
        eng.sync()
        ms, cnt = eng.kernel_time_ms("dual")
        out.append(round(ms / cnt, 4))
    eng.iterate_end()
    eng.close()
    print(f"handle {it} pad {pad} field pad {fpads[it] if fpads else 0}: kernel ms {out}", flush=True)
