export TMPDIR=/tmp RCN_EXPERIMENT=1
mkdir -p gpurun_out/r06m
timeout 2400 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "handle SIGUSR1 nostop noprint" -ex run -ex "bt 30" -ex "thread apply all bt 14" --args python -m pytest tests -m gpu -q -x > gpurun_out/r06m/gdb_full.log 2>&1
grep -n "SIGSEGV\|received signal\|passed\|failed" gpurun_out/r06m/gdb_full.log | head -10
grep -n "^#" gpurun_out/r06m/gdb_full.log | head -40 | cut -c1-260
