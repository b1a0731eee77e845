"""Diagnostic: which tracts does the fast scan kernel lose?  Compares raw records with the located scan (generic list kernel)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import tatajuba_amd as tj
k, m = int(os.environ.get("TJ_K", "10")), int(os.environ.get("TJ_M", "3"))
s = tj.synth_stream(30000, 150, 300000)
c = tj.Counter(k)
if os.environ.get("TJ_DEV"):
    import torch
    t = torch.from_numpy(s.copy()).cuda()
    if os.environ.get("TJ_DEV") == "2": c.set_stream(torch.cuda.current_stream().cuda_stream)
    half = (s.size // 2 // (151 * 16)) * (151 * 16)
    c.scan_device(t.data_ptr(), half, m); c.scan_device(t.data_ptr() + half, s.size - half, m); c.sync()
elif os.environ.get("TJ_TWO"):
    half = (s.size // 2 // (151 * 16)) * (151 * 16)
    c.scan_host(s[:half], m); c.scan_host(s[half:], m)
else:
    c.scan_host(s, m)
got = c.download_raw()
loc = c.scan_host_located(s, m)
print("got", len(got), "located", len(loc), loc.dtype)
def key(a): return list(zip(a["ctx0"].tolist(), a["ctx1"].tolist(), a["meta"].tolist()))
cg = collections.Counter(key(got))
missing = []
for r, kk in zip(loc, key(loc)):
    if cg[kk] > 0: cg[kk] -= 1
    else: missing.append(int(r["pos"]))
extra = sum(v for v in cg.values() if v > 0)
print("missing", len(missing), "extra", extra)
OWN = 16288
mm = np.array(missing)
if len(mm):
    w = (mm % OWN) + 32
    print("window positions of missing tract starts (first 60):", sorted(w.tolist())[:60])
    print("lane:", collections.Counter((w // 32).tolist()).most_common(12))
    print("tile:", collections.Counter((mm // OWN).tolist()).most_common(8))
    for p in missing[:8]:
        print(p, bytes(s[max(0, p - 14):p + 24]).replace(b"\n", b"|"))
