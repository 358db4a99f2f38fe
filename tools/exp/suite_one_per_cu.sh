#!/bin/bash
# the GPU parity suites with EVERY window alone on a CU (all banded alignments through the code waves); the tests that
# assert the launch shape itself are left out
export RCN_EXPERIMENT=1   # engine.hip read_knobs: RCN_* switches are ignored without it
RCN_SPLIT=0 RCN_WG_PER_CU=1 timeout 2000 python -m pytest tests -m gpu -q -x \
  --deselect tests/test_gpu_parity.py::test_work_groups_per_cu_do_not_change_results \
  --deselect tests/test_gpu_product_path.py::test_split_launch_on_and_off \
  --deselect tests/test_gpu_product_path.py::test_split_launch_rule \
  --deselect tests/test_gpu_product_path.py::test_refs_form_streamed_and_queued \
  --deselect tests/test_gpu_product_path.py::test_code_waves_of_the_deep_launch \
  --deselect tests/test_gpu_bench_contract.py 2>&1 | tail -15
