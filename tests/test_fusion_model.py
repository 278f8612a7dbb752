"""CPU: the FIR -> range fusion model (tools/proto/fir_range_fusion_model.py) -- its fused transform sequence reproduces
the two-stage result exactly, and its counts at configs[2] are the ones DESIGN.md quotes."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "proto"))


def test_fused_sequence_equals_two_stage_and_counts(capsys):
    import fir_range_fusion_model as M
    g = M.main()   # asserts the algebra to 1e-12 itself
    assert g["range_transforms_per_pulse"] == 11 and abs(g["fir_transforms_per_pulse"] - 9.518) < 1e-3
    assert g["fused_on_the_xs_grid"] == 18 and g["fused_transforms_per_pulse"] == 22
    assert g["predicted_fused_us_per_cpi"] > g["two_stage_us_per_cpi"]   # the fusion loses at one workgroup per CU
