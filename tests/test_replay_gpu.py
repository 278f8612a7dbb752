"""GPU: replay of an .rspduo capture through the HIP engine (int16 device path, the clutter-filter
chain, the device-resident batched CFAR, the JSON frames), checked against the compiled-reference
fixtures."""
import json

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def medium_cfg(g):
    fs, n, dmin, dmax, fmin, fmax, rh = (int(v) for v in g["params"])
    pfa, ng, nt, md, mdop = g["det_params"][:5]
    return {"fs": fs, "n_samples": n,
            "ambiguity": {"delayMin": dmin, "delayMax": dmax, "dopplerMin": fmin, "dopplerMax": fmax},
            "detection": {"enable": True, "pfa": pfa, "nGuard": int(ng), "nTrain": int(nt), "minDelay": int(md),
                          "minDoppler": mdop},
            "clutter": {"enable": False}}


def test_replay_int16_capture(built_lib, tmp_path):
    import blah2_amd
    from blah2_amd import replay as R
    assert blah2_amd.device_count() > 0
    g = load_golden("medium")
    n = int(g["params"][1])
    path = str(tmp_path / "cap.rspduo")
    # three CPIs: the fixture, its negation (same map magnitudes), the fixture again
    np.concatenate([g["iq"], -g["iq"], g["iq"]]).tofile(path)
    cap = R.RspduoFile(path, n)
    assert cap.n_cpis == 3
    cfg = medium_cfg(g)
    res = R.replay(cap, R.gpu_processor(cfg, 0, batch=2), batch=2)
    assert [r["cpi"] for r in res] == [0, 1, 2]
    for r in res:
        assert abs(r["noisePower"] - g["metrics"][0]) < 1e-3
        assert abs(r["maxPower"] - g["metrics"][1]) < 1e-3
        assert r["delay"] == g["cfar"][0].tolist() and r["doppler"] == g["cfar"][1].tolist()
        assert np.allclose(r["snr"], g["cfar"][2], rtol=0, atol=1e-3)
    cfg["clutter"] = {"enable": True, "delayMin": int(g["clutter_params"][0]), "delayMax": int(g["clutter_params"][1])}
    res = R.replay(cap, R.gpu_processor(cfg, 0, batch=2), batch=2)
    for r in res:
        assert abs(r["noisePower"] - g["chain_metrics"][0]) < 1e-3
        assert set(zip(r["delay"], r["doppler"])) == set(zip(g["chain_cfar"][0], g["chain_cfar"][1]))


def test_replay_uploads_out_of_the_page_cache(built_lib):
    """read_mode="mapped": whole pages of the mapped capture are registered with the device and uploaded from,
    the ragged ends of a batch (the fixture's CPI is not a whole number of pages) through the small pinned buffer; seven
    CPIs in batches of three leave a ragged last batch too.  Same results as the pinned-ring path, and the mode must have
    stayed "mapped" (a runtime that refuses the registration falls back to pread -- on /dev/shm it must not)."""
    import os
    from blah2_amd import replay as R
    g = load_golden("medium")
    n = int(g["params"][1])
    path = f"/dev/shm/blah2_test_mapped_{os.getpid()}.rspduo"
    rng = np.random.default_rng(5)
    cpis = [g["iq"], -g["iq"]] + [np.roll(g["iq"], int(k), axis=0) for k in rng.integers(1, 50, 5)]
    np.concatenate(cpis).tofile(path)
    try:
        cfg = medium_cfg(g)
        out = {}
        for mode in ("mapped", "memmove", "pread"):
            cap = R.RspduoFile(path, n)
            chain = R.GpuChain(cfg, 0, batch=3, reader_threads=3, read_mode=mode)
            out[mode] = R.replay(cap, chain, batch=3)
            assert chain.read_mode == mode
            chain.close()
            cap.close()
        assert [r["cpi"] for r in out["mapped"]] == list(range(7))
        for a, b, c in zip(out["mapped"], out["pread"], out["memmove"]):
            assert a == b == c
        for r in out["mapped"][:2]:
            assert r["delay"] == g["cfar"][0].tolist() and r["doppler"] == g["cfar"][1].tolist()
    finally:
        os.remove(path)


def test_replay_skips_cpis_whose_clutter_filter_fails(built_lib, tmp_path):
    """blah2.cpp:270-273: `if (!filter->process(x, y)) continue;` -- an all-zero reference channel makes the
    normal equations singular; that CPI is dropped, its neighbours in the same batch are not."""
    from blah2_amd import replay as R
    g = load_golden("medium")
    n = int(g["params"][1])
    dead = g["iq"].copy()
    dead[:, 0:2] = 0
    path = str(tmp_path / "cap.rspduo")
    np.concatenate([g["iq"], dead, g["iq"]]).tofile(path)
    cfg = medium_cfg(g)
    cfg["clutter"] = {"enable": True, "delayMin": int(g["clutter_params"][0]), "delayMax": int(g["clutter_params"][1])}
    res = R.replay(R.RspduoFile(path, n), R.gpu_processor(cfg, 0, batch=3), batch=3)
    assert [bool(r.get("skipped")) for r in res] == [False, True, False]
    assert abs(res[0]["noisePower"] - g["chain_metrics"][0]) < 1e-3 and abs(res[2]["noisePower"] - g["chain_metrics"][0]) < 1e-3


def test_replay_json_frames(built_lib, tmp_path, capsys):
    """`python -m blah2_amd.replay --json`: per CPI the map and detection documents of blah2.cpp:304-317,
    written by the C++ host classes; the detector chain includes Centroid and Interpolate (nCentroid)."""
    import yaml
    from blah2_amd import replay as R
    from oracle import blah2_oracle as O
    g = load_golden("medium")
    fs, n, dmin, dmax, fmin, fmax, rh = (int(v) for v in g["params"])
    pfa, ng, nt, md, mdop = g["det_params"][:5]
    cap = str(tmp_path / "cap.rspduo")
    np.concatenate([g["iq"], g["iq"]]).tofile(cap)
    cfgp = str(tmp_path / "config.yml")
    yaml.safe_dump({"capture": {"fs": fs},
                    "process": {"data": {"cpi": n / fs},
                                "ambiguity": {"delayMin": dmin, "delayMax": dmax, "dopplerMin": fmin, "dopplerMax": fmax},
                                "clutter": {"enable": False, "delayMin": dmin, "delayMax": dmax},
                                "detection": {"enable": True, "pfa": float(pfa), "nGuard": int(ng), "nTrain": int(nt),
                                              "minDelay": int(md), "minDoppler": float(mdop), "nCentroid": 6}},
                    "network": {"ip": "0.0.0.0", "ports": {"map": 3001, "detection": 3002}}}, open(cfgp, "w"))
    R.main([cap, "-c", cfgp, "--batch", "2", "--json"])
    lines = capsys.readouterr().out.strip().split("\n")
    assert len(lines) == 4  # (map, detection) x 2 CPIs, file order
    t_ms = int(round(1000.0 * n / fs))
    for k in range(2):
        m, d = json.loads(lines[2 * k]), json.loads(lines[2 * k + 1])
        assert list(m) == ["timestamp", "nRows", "nCols", "noisePower", "maxPower", "delay", "doppler", "data"]
        assert list(d) == ["timestamp", "delay", "doppler", "snr"]
        assert m["timestamp"] == d["timestamp"] == k * t_ms
        assert (m["nRows"], m["nCols"]) == g["map"].shape
        want = O.map_db(g["map"], g["metrics"][0])
        got = np.array(m["data"])
        assert np.max(np.abs(got - want)) < 0.01 + 0.005  # two-decimal truncation + the 0.005 dB device gate
        assert abs(m["noisePower"] - g["metrics"][0]) < 0.011 and abs(m["maxPower"] - g["metrics"][1]) < 0.011
        # the fixture's interpolated detections (compiled reference: CFAR -> Centroid -> Interpolate), delay in km
        km = g["interp"][0] * (O.C_LIGHT / fs) / 1000
        assert len(d["delay"]) == len(km)
        assert np.allclose(d["delay"], km, atol=0.011) and np.allclose(d["doppler"], g["interp"][1], atol=0.011)
        assert np.allclose(d["snr"], g["interp"][2], atol=0.011)


def test_replay_runs_the_fused_fir_where_it_is_covered(built_lib, tmp_path):
    """A geometry one 4096-point transform covers (31 pulses of 6149 samples, 2047 taps): GpuChain hands the filter's taps to
    the range kernel (range_fir_kernel) instead of writing the filtered channel; clutter: {fused: false} keeps the two-stage
    chain.  Same detections, metrics to 1e-4 dB; a dead reference channel is skipped either way."""
    import blah2_amd
    from blah2_amd import replay as R
    from test_fused_fir_gpu import synth
    n = 190_647
    x, y = synth(n, n, 77)
    iq = np.stack([x.real, x.imag, y.real, y.imag], axis=-1).astype(np.int16)
    dead = iq.copy()
    dead[:, 0:2] = 0
    path = str(tmp_path / "cap.rspduo")
    np.concatenate([iq, dead, iq]).tofile(path)
    cfg = {"fs": n, "n_samples": n,
           "ambiguity": {"delayMin": -24, "delayMax": 2023, "dopplerMin": -15, "dopplerMax": 15},
           "detection": {"enable": True, "pfa": 1e-5, "nGuard": 2, "nTrain": 6, "minDelay": 5, "minDoppler": 1.0},
           "clutter": {"enable": True, "delayMin": -24, "delayMax": 2023}}
    out = {}
    for fused in (True, False):
        cfg["clutter"]["fused"] = fused
        chain = R.gpu_processor(cfg, 0, batch=2)
        assert chain.fused_fir == fused
        res = R.replay(R.RspduoFile(path, n), chain, batch=2)
        out[fused] = res
        assert [bool(r.get("skipped")) for r in res] == [False, True, False]
        assert chain.amb.info(blah2_amd._lib.INFO_LAST_RANGE_KERNEL) == (blah2_amd._lib.RANGE_FIR if fused else blah2_amd._lib.RANGE_E16)
    for a, b in zip(out[True], out[False]):
        if a.get("skipped"):
            continue
        assert abs(a["noisePower"] - b["noisePower"]) < 1e-4 and abs(a["maxPower"] - b["maxPower"]) < 1e-4
        assert list(zip(a["delay"], a["doppler"])) == list(zip(b["delay"], b["doppler"])) and len(a["delay"]) >= 1
        assert np.allclose(a["snr"], b["snr"], rtol=0, atol=1e-3)
