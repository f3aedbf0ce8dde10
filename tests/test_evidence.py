"""Evidence hygiene: the figures bench.py reports from committed profiles are what the committed summaries say, for
EVERY bench config, and are made by the committed tool."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_hbm_traffic_json_is_what_the_tool_makes_from_the_committed_summaries(tmp_path):
    want = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    keep = open(os.path.join(ROOT, "profiles", "hbm_traffic.json")).read()
    try:
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_hbm_traffic.py")], check=True, capture_output=True)
        got = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    finally:
        open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w").write(keep)
    assert got == want


def test_every_bench_config_has_current_counters_and_both_floors():
    sys.path.insert(0, ROOT)
    import bench
    from imsim_amd import configs
    table = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    names = {c[0] for c in bench.EXTRA_CONFIGS} | {"c3"}
    assert names == set(configs.BENCH_CONFIGS), "every bench config is on the default line (extra.configs)"
    for name in sorted(names):
        step = table[name]["_step"]
        assert step["sq_source"].startswith("profiles/round6_"), (name, step["sq_source"])
        assert os.path.exists(os.path.join(ROOT, step["sq_source"]))
        insts = step.get("valu_wave_insts_per_step") or step.get("valu_wave_insts_per_ccd")
        mix = step.get("class_mix_per_step") or step.get("class_mix_per_ccd")
        assert insts > 0 and mix and set(mix) <= set(bench.CLASS_NS) and 0.3 * insts < sum(mix.values()) <= insts
        # (the 100-object FFT line is a handful of ~10-us launches: GRBM_GUI_ACTIVE counts a little beyond their time stamps, 2.5 - 2.6 "GHz")
        assert 1.5 < step["sustained_clock_ghz"] < (2.8 if name == "fft" else 2.6), name
        kern = configs.BENCH_CONFIGS[name]["kernel"]
        assert kern in table[name] and table[name][kern]["source"].startswith("profiles/round6_"), (name, kern)
    assert {"shoot (k_shoot_photons<2>)"} <= set(table["c4"]["_phases"])
