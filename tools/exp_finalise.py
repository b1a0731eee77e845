"""Diagnostic: time the finalise of library builds (LIBS="a.so b.so") on one workload (bench.py arguments after --)."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
for lib in os.environ.get("LIBS", "libtatajuba_amd.so").split():
    env = dict(os.environ, TJ_DIAG_LIB=lib)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", os.environ.get("TJ_STEPS", "5"), "--warmup", os.environ.get("TJ_WARMUP", "2"), "--no-io-stages", "--no-cpu-baseline"] + args,
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode().strip().splitlines()
    d = json.loads(out[-1])
    print(lib, "scan %.3f ms partition %.3f ms finalise %.3f ms step %.3f ms" % (d["stages"]["scan"]["ms"], d["stages"].get("partition", {}).get("ms", 0.0),
                                                                               d["stages"]["finalise"]["ms"], d["ms_per_step"]))
