"""Diagnostic: cost of the cross-sample merge (tjamd_merge_samples) for N samples' histograms on one GPU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tatajuba_amd.build as B
B._SO = os.path.join(ROOT, "tatajuba_amd", os.environ.get("TJ_DIAG_LIB", "libtatajuba_amd.so"))
import tatajuba_amd.capi as capi
capi.library_path = lambda: B._SO
import numpy as np, torch
import tatajuba_amd as tj
from tatajuba_amd.dist import device_bytes_tensor, merge_histograms_device
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
parts, cnts = [], []
c = tj.Counter(10)
c.set_stream(torch.cuda.current_stream().cuda_stream)
for r in range(N):
    s = tj.synth_stream(3_000_000, 150, 5_000_000, seed_reads=0x7A7A1000 + r, variant_seed=r, n_threads=16)
    d = torch.from_numpy(s).cuda()
    c.reset(); c.scan_device(d.data_ptr(), s.size, 3); c.finalise(1, 5)
    n = c.n_kept
    parts.append(device_bytes_tensor(c.kept_device_ptr, n * 24, dev).clone()); cnts.append(n)
rec = torch.cat(parts)
print("records per sample", cnts)
for it in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    keys, mat = merge_histograms_device(c, rec, cnts)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
print("merge of %d samples: %.3f ms, union %d keys" % (N, dt * 1e3, mat.shape[0]))
