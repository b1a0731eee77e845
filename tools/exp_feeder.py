"""Diagnostic: where new_or_append_hopo_counter_from_file spends its time on a plain FASTQ file, by feeder threads
(TATAJUBA_AMD_FEEDER_THREADS) -- the library's own trace lines (TATAJUBA_AMD_FEEDER_TRACE=1) and the wall time."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import tatajuba_amd as tj
    fq = sys.argv[2]
    opt = tj.Options.defaults(10, 3, 5, True)
    best = 1e9
    for rep in range(3):
        t = time.perf_counter()
        h = tj.HopoCounter.new_or_append_from_file(None, fq, opt)
        t1 = time.perf_counter()
        h.finalise()
        t2 = time.perf_counter()
        best = min(best, t2 - t)
        print("rep %d: read+scan %.1f ms, finalise %.1f ms" % (rep, (t1 - t) * 1e3, (t2 - t1) * 1e3), flush=True)
        h.delete()
    print("threads %s: best %.1f ms = %.1f M reads/s" % (os.environ.get("TATAJUBA_AMD_FEEDER_THREADS", "default"), best * 1e3, int(sys.argv[3]) / best / 1e6))
    sys.exit(0)
import tatajuba_amd as tj
n = 2_000_000
s = tj.synth_stream(n, 150, 5_000_000, n_threads=16)
reads = bytes(s).split(b"\n")[:-1]
tmp = tempfile.mkdtemp(dir=os.environ.get("TMPDIR", "/tmp"))
fq = os.path.join(tmp, "feeder.fq")
with open(fq, "wb") as f:
    for i in range(0, len(reads), 100000):
        f.write(b"".join(b"@r%d\n%s\n+\n%s\n" % (j, reads[j], b"I" * len(reads[j])) for j in range(i, min(len(reads), i + 100000))))
for th in ("8", "16", "32", None):
    env = dict(os.environ, TATAJUBA_AMD_FEEDER_TRACE="1")
    if th: env["TATAJUBA_AMD_FEEDER_THREADS"] = th
    else: env.pop("TATAJUBA_AMD_FEEDER_THREADS", None)
    subprocess.run([sys.executable, os.path.abspath(__file__), "child", fq, str(n)], env=env)
