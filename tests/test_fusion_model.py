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


def test_the_kernels_window_form_equals_two_stage():
    """range_fir_kernel's own sequence (window spectra, block spectra, direct edge products), in fp64 against the plain two-stage
    computation: pulses that end inside a block, on a block boundary (an extra block of |delayMin| samples) and just past
    one; pulse 0's zeroed first samples; the last pulse's look-ahead."""
    import numpy as np
    import fir_range_fusion_model as M
    rng = np.random.default_rng(3)
    F, L = 64, 32
    for n_corr, dmin, taps, n_d in ((75, -5, 30, 4), (64, -3, 33, 3), (70, -7, 20, 3), (37, -5, 12, 5), (96, 0, 33, 2)):
        n = n_corr * n_d + max(-dmin, 1) + 3
        x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        y = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        w = (rng.standard_normal(taps) + 1j * rng.standard_normal(taps)) * 0.2
        lags = np.arange(dmin, dmin + L + 1)
        _, R2 = M.two_stage(x, y, w, dmin, n_corr, n_d, lags)
        Rf, count = M.fused_window_form(x, y, w, dmin, n_corr, n_d, lags, F)
        assert np.max(np.abs(Rf - R2)) <= 1e-10 * np.max(np.abs(R2)), (n_corr, dmin)
        SB = -(-(n_corr - dmin) // L)
        assert abs(count - (3 * SB + 2 - 1 / n_d)) < 1e-9   # 3 blocks + 2 transforms a pulse (pulse 0 has no history block)
