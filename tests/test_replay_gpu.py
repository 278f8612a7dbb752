"""GPU: replay of an .rspduo capture through the HIP engine (int16 device path
and the clutter-filter chain), checked against the compiled-reference fixtures."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def test_replay_int16_capture(built_lib, tmp_path):
    import blah2_amd
    from blah2_amd import replay as R
    assert blah2_amd.device_count() > 0
    g = load_golden("medium")
    fs, n, dmin, dmax, fmin, fmax, rh = (int(v) for v in g["params"])
    pfa, ng, nt, md, mdop = g["det_params"][:5]
    path = str(tmp_path / "cap.rspduo")
    # three CPIs: the fixture, its negation (same map magnitudes), the fixture again
    np.concatenate([g["iq"], -g["iq"], g["iq"]]).tofile(path)
    cap = R.RspduoFile(path, n)
    assert cap.n_cpis == 3
    cfg = {"fs": fs, "n_samples": n,
           "ambiguity": {"delayMin": dmin, "delayMax": dmax, "dopplerMin": fmin, "dopplerMax": fmax},
           "detection": {"enable": True, "pfa": pfa, "nGuard": int(ng), "nTrain": int(nt), "minDelay": int(md),
                         "minDoppler": mdop},
           "clutter": {"enable": False}}
    res = R.replay(cap, R.gpu_processor(cfg, 0, batch=2), batch=2)
    assert [r["cpi"] for r in res] == [0, 1, 2]
    for r in res:
        assert abs(r["noisePower"] - g["metrics"][0]) < 1e-3
        assert abs(r["maxPower"] - g["metrics"][1]) < 1e-3
        assert r["delay"] == g["cfar"][0].tolist() and r["doppler"] == g["cfar"][1].tolist()
    cfg["clutter"] = {"enable": True, "delayMin": int(g["clutter_params"][0]), "delayMax": int(g["clutter_params"][1])}
    res = R.replay(cap, R.gpu_processor(cfg, 0, batch=1), batch=1)
    for r in res:
        assert abs(r["noisePower"] - g["chain_metrics"][0]) < 5e-3
