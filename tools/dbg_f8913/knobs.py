import os, sys, subprocess, json
sys.path.insert(0, os.getcwd())
if len(sys.argv) > 1:
    # child: one knob set
    from racon_amd.polisher import Polisher
    from racon_amd.engine import HipEngine
    from oracle import oracle_lib
    d = "tools/dbg_f8913"
    p = Polisher(d + "/reads.fastq", d + "/overlaps.paf", d + "/target.fastq", "kF", 500, 10.0, 0.3, True, 3, -5, -4, 8, 1)
    p.initialize(); b = p.windows()
    o = oracle_lib.consensus(b, 3, -5, -4, True, 0, simd=True)
    e = HipEngine(3, -5, -4, True)
    g = e.consensus(b)
    bad = [w for w in range(b.n_windows) if g.consensus[w] != o.consensus[w]]
    st = e.stats()
    print("RESULT", sys.argv[1], "differ", bad, "banded", st.get("n_banded"), "redone", st.get("n_band_redone"), "code_wave", st.get("n_code_wave"))
    sys.exit(0)
for kn in ("", "RCN_FORCE_EXACT=1", "RCN_NO_BAND=1", "RCN_BAND_SCORES=1", "RCN_FORCE_TIE=2", "RCN_FORCE_TIE=3", "RCN_FORCE_BAND_FAIL=1", "RCN_WIDE_ONLY=1", "RCN_NO_CODE_WAVE=1", "RCN_SPLIT=0", "RCN_NO_PTAB=1"):
    env = dict(os.environ); env["RCN_EXPERIMENT"] = "1"
    if kn: k, v = kn.split("="); env[k] = v
    r = subprocess.run([sys.executable, __file__, kn or "default"], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    print([l for l in r.stdout.decode().splitlines() if l.startswith("RESULT")])
