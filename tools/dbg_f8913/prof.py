import os, sys, subprocess
sys.path.insert(0, os.getcwd())
if len(sys.argv) > 1:
    from racon_amd.batch import WindowBatch
    from racon_amd.polisher import Polisher
    from racon_amd.engine import HipEngine
    from oracle import oracle_lib
    d = "tools/dbg_f8913"
    p = Polisher(d + "/reads.fastq", d + "/overlaps.paf", d + "/target.fastq", "kF", 500, 10.0, 0.3, True, 3, -5, -4, 8, 1)
    p.initialize(); b = p.windows().select([2])
    o = oracle_lib.consensus(b, 3, -5, -4, True, 0, simd=True)
    g = HipEngine(3, -5, -4, True).consensus(b)
    print("RESULT", sys.argv[1], "same" if g.consensus[0] == o.consensus[0] else "DIFFER")
    sys.exit(0)
for kn in ("", "RCN_FORCE_TIE=3", "RCN_BAND_SCORES=1"):
    env = dict(os.environ); env["RCN_EXPERIMENT"] = "1"; env["RCN_PROF_LAYERS"] = "1"; env["RACON_HIP_LIB"] = os.path.join(os.getcwd(), "racon_amd/csrc/libracon_hip_prof.so")
    if kn: k, v = kn.split("="); env[k] = v
    r = subprocess.run([sys.executable, __file__, kn or "default"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    print("=====", kn or "default")
    for l in r.stdout.decode().splitlines():
        if l.startswith("RESULT") or "tie" in l: print(l[:200])
