"""tools/sq_summary.py on a counter pass that rocprofv3 SERIALISED (the round-5 finding: the split launch's second dispatch's HIP-event
interval contains the first, and the dispatches run under CU masks) -- the numbers are profiles/r05/n_sq_cfg2_summary.txt's."""
import csv
import json
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
import sq_summary  # noqa: E402

R05 = {"poa_window_kernel2": {"SQ_ACTIVE_INST_VALU": 3.77692e9, "SQ_ACTIVE_INST_SCA": 2.3266e9, "SQ_WAVE_CYCLES": 5.35237e10, "SQ_ACTIVE_INST_ANY": 7.01987e9,
                              "SQ_INSTS_VALU": 3.67312e9, "SQ_INSTS_SALU": 2.32132e9, "SQ_INSTS_LDS": 1.77391e8, "SQ_INSTS_SMEM": 5.28153e6,
                              "SQ_INSTS_BRANCH": 4.37824e8, "SQ_INSTS_VMEM_RD": 6.03584e7, "SQ_INSTS_VMEM_WR": 7.43975e7},
       "poa_window_kernel2_deep": {"SQ_ACTIVE_INST_VALU": 2.44917e8, "SQ_ACTIVE_INST_SCA": 1.66981e8, "SQ_WAVE_CYCLES": 1.02923e9, "SQ_ACTIVE_INST_ANY": 4.73044e8}}


def write_pass(tmp, n, launch_ms, step_ms, trace=None):
    d = os.path.join(tmp, "sq_cfg2_%d" % n)
    os.makedirs(os.path.join(d, "box"))
    with open(os.path.join(d, "box", "1_counter_collection.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
        for kn, cs in R05.items():
            for disp in (1, 2):                                 # two dispatches with the same counts: per-dispatch mean = the count
                for c, v in cs.items():
                    w.writerow(["%s%d" % (kn, disp), "rcn::%s(rcn::KParams)" % kn, c, v])
    if trace:
        with open(os.path.join(d, "box", "1_kernel_trace.csv"), "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["Kernel_Name", "Start_Timestamp", "End_Timestamp"])
            for kn, ms in trace.items():
                w.writerow(["rcn::%s(rcn::KParams)" % kn, 1000, 1000 + int(ms * 1e6)])
    line = {"roofline": {"step_kernel_ms": step_ms, "launches_per_step": 2, "launch_ms": launch_ms,
                         "split_launch": {"deep_cus": 32, "deep_windows": 32, "mid_cus": 0}}}
    open(d + ".json", "w").write("noise\n" + json.dumps(line) + "\n")


def test_serialised_split_pass_is_undone(tmp_path):
    write_pass(str(tmp_path), 1, [16.14, 33.44], 33.44)
    res = sq_summary.summarise(str(tmp_path), "cfg2", log=lambda *_: None)
    k2, deep = res["poa_window_kernel2"], res["poa_window_kernel2_deep"]
    assert k2["dispatch_ms"] == pytest.approx(17.30, abs=0.01) and "serialised" in k2["dispatch_ms_source"]
    assert (k2["cus"], k2["simds"]) == (224, 896) and (deep["cus"], deep["simds"]) == (32, 128)
    # the verdict's hand calculation from the same counters: vector ~0.41 (not 0.18), scalar ~0.25 (not 0.11), ~5.5 cycles per instruction
    assert k2["valu_issue_frac"] == pytest.approx(0.406, abs=0.005)
    assert k2["scalar_issue_frac"] == pytest.approx(0.250, abs=0.005)
    assert k2["cycles_per_instruction_per_simd"] == pytest.approx(5.5, abs=0.15)
    assert k2["waves_resident_avg"] == pytest.approx(5160, rel=0.02)
    assert deep["dispatch_ms"] == pytest.approx(16.14) and deep["valu_issue_frac"] == pytest.approx(0.1976, abs=0.002)


def test_concurrent_split_pass_is_left_alone(tmp_path):
    write_pass(str(tmp_path), 1, [17.8, 17.1], 17.95)           # the launches overlapped: their own HIP-event durations stand
    res = sq_summary.summarise(str(tmp_path), "cfg2", log=lambda *_: None)
    assert res["poa_window_kernel2"]["dispatch_ms"] == pytest.approx(17.1)
    assert "serialised" not in res["poa_window_kernel2"]["dispatch_ms_source"]


def test_kernel_trace_of_the_pass_wins(tmp_path):
    write_pass(str(tmp_path), 1, [16.14, 33.44], 33.44, trace={"poa_window_kernel2": 17.25, "poa_window_kernel2_deep": 16.0})
    res = sq_summary.summarise(str(tmp_path), "cfg2", log=lambda *_: None)
    assert res["poa_window_kernel2"]["dispatch_ms"] == pytest.approx(17.25)
    assert "kernel trace" in res["poa_window_kernel2"]["dispatch_ms_source"]
