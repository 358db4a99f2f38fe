import sys, time, subprocess, os
sys.path.insert(0, ".")
from racon_amd.synth import simulate_files
t=time.time(); paths, truth = simulate_files("/tmp/e2e_big", contig_len=1_000_000, coverage=30.0, read_len=10000, n_contigs=1, seed=5); print("simulate %.1fs" % (time.time()-t), flush=True)
for ovl in ("sam", "paf"):
    t=time.time()
    r = subprocess.run(["racon_amd/host/racon_hip", "-t", "32", paths["reads"], paths[ovl], paths["targets"]], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    print(ovl, "wall %.2fs rc %d out %d bytes" % (time.time()-t, r.returncode, len(r.stdout)))
    print(r.stderr.decode()[-900:])
