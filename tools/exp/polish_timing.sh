python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from racon_amd.synth import simulate_window_files
simulate_window_files("/tmp/cfg2files", 1_000_000, 30.0, 10000, seed=20260921, workers=16)
PY
for k in 1 2 3; do RACON_HIP_TIMING=1 racon_amd/host/racon_hip -t 32 /tmp/cfg2files/reads.fastq /tmp/cfg2files/overlaps.sam /tmp/cfg2files/targets.fasta 2>&1 >/dev/null | grep -E "timing|polish\]|racon::Polisher::\]" | cut -c1-260; echo ==; done
