import os, sys, subprocess, hashlib
sys.path.insert(0, os.getcwd())
d = "tools/dbg_f8913"
exe = "racon_amd/host/racon_hip"
ref = open(d + "/oracle.fasta", "rb").read()
def run(flags, env_add):
    env = dict(os.environ); env.pop("RACON_HIP_DEVICE_WINDOWS", None); env.update(env_add)
    r = subprocess.run([exe, "-f", "-t", "8"] + flags + [d + "/reads.fastq", d + "/overlaps.paf", d + "/target.fastq"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    return r.stdout
for label, flags, env in (("device align", ["--cudaaligner-batches", "1"], {}), ("host aligner, device cigars (2)", [], {"RACON_HIP_DEVICE_WINDOWS": "2"}),
                          ("host aligner, device windows (1)", [], {"RACON_HIP_DEVICE_WINDOWS": "1"}), ("host-built (0)", [], {"RACON_HIP_DEVICE_WINDOWS": "0"}),
                          ("device align + verify", ["--cudaaligner-batches", "1"], {"RACON_HIP_VERIFY": "1"})):
    out = run(flags, env)
    print(label, "== oracle:", out == ref, hashlib.md5(out).hexdigest(), len(out))
    if out != ref and out:
        a, b = out.split(b"\n")[1], ref.split(b"\n")[1]
        k = next((i for i in range(min(len(a), len(b))) if a[i] != b[i]), None)
        print("   first difference at base", k, "lengths", len(a), len(b), a[k-10:k+20], b[k-10:k+20])
# window by window: host-built windows through the engine against the oracle
from racon_amd.polisher import Polisher
from racon_amd.engine import HipEngine
from oracle import oracle_lib
p = Polisher(d + "/reads.fastq", d + "/overlaps.paf", d + "/target.fastq", "kF", 500, 10.0, 0.3, True, 3, -5, -4, 8, 1)
p.initialize(); b = p.windows()
o = oracle_lib.consensus(b, 3, -5, -4, True, 0, simd=True)
g = HipEngine(3, -5, -4, True).consensus(b)
bad = [w for w in range(b.n_windows) if g.consensus[w] != o.consensus[w]]
print("host-built windows through the engine: differ", bad)
# the device aligner's CIGARs against the host aligner's
