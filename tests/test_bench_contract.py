"""The bench line's contract (keys the driver and the judge read), checked on the committed round-3 line
(profiles/r3_bench_line.json: `python bench.py` on one MI355X), the strong-scaling variant and the forced-RCCL run."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(name):
    return json.load(open(os.path.join(ROOT, "profiles", name)))


def test_committed_bench_line_has_the_contract_fields():
    line = load("r3_bench_line.json")
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity"):
        assert key in line, key
    assert line["metric"].split(" at ")[0] in base["metric"]            # BASELINE.json's metric
    assert line["unit"] == "columns/s" and line["higher_is_better"] is True and line["scaling"] == "weak"
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 0 and line["steps"] * line["ms_per_step"] >= 3000.0   # timed region >= 3 s
    assert line["vs_baseline"] is None and base["published"] == {}      # no published number for this metric
    assert line["dtype"] == "f64" and line["data"] == "synthetic" and "workload" in line["config"]
    assert "model" not in line["config"]
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and (r["traffic"] is None or r["traffic"] > 0)
    assert r["traffic"] is None or "not measured in this run" in r["traffic_source"]     # carried from the profiled run, and says so
    c = line["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert c["host_cores"] >= c["cores"] and c["one_thread"]["cores"] == 1 and 0 < c["one_thread"]["value"] < c["value"]
    v = line["roofline_valu"]                       # utilisation next to the algorithmic rate
    assert 0.3 < v["issue_utilisation"] < 1.0 and 500 < v["instructions_per_64_lines"] < 2000 and "not measured in this run" in v["utilisation_source"]
    # throughput is whole-job columns over the timed steps
    cols = line["config"]["columns_per_step"]
    assert abs(line["value"] - cols / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-9


def test_bench_line_carries_its_own_parity_against_the_reference():
    """the column the bench prints is compared, in the same run, with the reference's own C on that column."""
    p = load("r3_bench_line.json")["parity"]
    assert p["kind"] == "reference" and p["ok"] is True
    assert p["max_abs_flux_diff_w_m2"] <= p["tolerance_w_m2"] == 1e-3
    assert p["max_tau_err_of_layer_max"] <= 2e-6
    for k in ("max_tau_err_pointwise_rel", "max_transmission_err", "flux_diff_w_m2", "tau_lw", "tau_sw"):
        assert k in p


def test_bench_line_reports_solvers_fine_grids_and_the_unchanged_callers_rate():
    line = load("r3_bench_line.json")
    s = line["roofline_solvers"]
    assert s["lw"]["avg_launch_ms"] > 0 and s["sw"]["avg_launch_ms"] > 0 and "hbm" in s["bound"]
    assert 0.3 < s["sw"]["frac_hbm_with_park"] < 1.0                    # the shortwave solver streams: bandwidth-bound
    assert line["reference_order_columns_per_s"] > 0
    assert line["reference_abi"]["fast0_columns_per_s"] > 0 and line["reference_abi"]["fast3_columns_per_s"] > 0
    g3 = line["fine_grid"]["G3_lw_0.001cm-1"]
    assert g3["n"] == 3249001 and g3["ran"]["tree_levels"] > 0 and g3["tree_vs_ring"]["max_diff_of_layer_max"] < 2e-6
    assert "SELF-COMPARISON" in g3["tree_vs_ring"]["what"]
    assert line["fine_grid"]["G2_lw_0.1cm-1"]["n"] == 32491


def test_strong_scaling_line():
    line = load("r3_strong_100_columns_bench_line.json")       # BASELINE config 3's column count
    assert line["scaling"] == "strong" and line["config"]["columns_per_step"] == 100
    assert abs(line["value"] - 100 / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-9


def test_rccl_path_at_world_size_one_costs_about_one_percent():
    """one gather of the job's fluxes after the last step, not one per step (round 2: -6 %).  What is left,
    0.8-1.4 % over seven runs, is not the gather: with an RCCL communicator alive in the process the line kernels
    themselves run that much slower (the same run with gloo in RCCL's place: no loss) -- DESIGN.md section 6."""
    plain, forced = load("r3_bench_line.json"), load("r3_rccl_world1_bench_line.json")
    assert forced["rccl_ranks"] == 1 and forced["collective"]["backend"] == "nccl" and forced["collective"]["gathers_in_timed_region"] == 1
    assert forced["value"] >= 0.98 * plain["value"]
    slower = forced["kernel_ms_per_step"]["gas_optics_sw"] / plain["kernel_ms_per_step"]["gas_optics_sw"]
    assert 1.0 <= slower < 1.03                     # the loss sits in the kernels' own durations


# ---- round 4 (profiles/r4_*) ------------------------------------------------------------------------------------------- #
def test_round4_line_prices_the_kernel_against_what_binds_it_and_checks_a_discriminating_column():
    """VERDICT r3, task 2: `roofline` is the vector-issue view (the HBM view stays as `roofline_hbm`), and the parity leg
    carries the physically scaled list's column 0 -- an atmosphere whose fluxes depend on tau."""
    line = load("r4_bench_line.json")
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert line["metric"].split(" at ")[0] in base["metric"] and line["unit"] == "columns/s" and line["scaling"] == "weak"
    assert line["n_gpus"] == 1 and line["steps"] * line["ms_per_step"] >= 2800.0 and line["dtype"] == "f64"
    assert line["ms_per_step"] <= 145.0 and line["value"] >= 440.0                       # VERDICT r3, task 1
    assert len(line["ms_per_step_by_rank"]) == 1
    r = line["roofline"]
    assert r["bound"] == "valu_issue" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.3 < r["frac"] < 1.0
    assert "not measured in this run" in r["instructions_source"] and r["traffic"] > 0
    h = line["roofline_hbm"]
    assert h["bound"] == "hbm" and abs(h["frac"] - h["achieved"] / h["peak"]) < 1e-12
    v = line["roofline_valu"]
    assert v["instructions_per_64_lines"] <= 780                                          # VERDICT r3, task 1's alternative
    p = line["parity"]
    assert p["kind"] == "reference" and p["ok"] is True and p["max_abs_flux_diff_w_m2"] <= 1e-3 and p["max_tau_err_of_layer_max"] <= 2e-6
    q = p["physical_list"]
    assert q["ok"] is True and q["max_abs_flux_diff_w_m2"] <= 1e-3 and 0.62 < q["rsds_over_rsdt"] < 0.75
    assert 250.0 < q["reference_fluxes_w_m2"]["rlut"] < 300.0
    assert q["tau_lw"]["of_layer_max"] <= 2e-6 and q["tau_sw"]["of_layer_max"] <= 2e-6
    c = line["cpu_baseline"]
    assert c["kind"] == "reference" and c["cores"] >= 1 and c["value"] > 0 and c["one_thread"]["cores"] == 1
    cols = line["config"]["columns_per_step"]
    assert abs(line["value"] - cols / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-9


def test_round4_lines_of_the_other_configurations():
    g3 = load("r4_g3_pipeline_bench_line.json")                  # the ~3M-point grid through the whole pipeline (task 3-i)
    assert "0.001" in g3["config"]["workload"] and g3["value"] > 15.0
    era = load("r4_era5_like_bench_line.json")                   # config 5's grids at 64 columns per step (task 8)
    assert "@0.1 cm-1" in era["config"]["workload"] and "@10 cm-1" in era["config"]["workload"] and era["config"]["columns_per_step"] == 64
    assert era["value"] > 400.0
    # (round 4 kept its forced-RCCL line from a build 2.5 h older than its final plain line -- 486.8 against 497.3, 97.9 %:
    # two builds, not a measurement of RCCL's cost.  Only the line's own fields are checked here; the cost is measured by
    # the round-5 PAIR below, both runs taken back to back on one build and kept in one file.)
    forced = load("r4_rccl_world1_bench_line.json")
    assert forced["rccl_ranks"] == 1 and forced["collective"]["gathers_in_timed_region"] == 1 and forced["value"] > 440.0
    strong = load("r4_strong_100_columns_bench_line.json")
    assert strong["scaling"] == "strong" and strong["config"]["columns_per_step"] == 100
    drv = load("r4_reference_driver_timing.json")                # the unchanged driver binary; with its opt-in for three rows
    assert drv["fast3"]["columns_per_s"] > 150.0 and drv["fast3rows"]["columns_per_s"] > drv["fast3"]["columns_per_s"]


# ---- round 5 (profiles/r5_*: scripts/evidence_round.sh -- one build, one session) -------------------------------------- #
def _kernel_source_sha():
    import hashlib
    hip = os.path.join(ROOT, "grtcode_amd", "csrc", "hip")
    return hashlib.sha256(b"".join(open(os.path.join(hip, f), "rb").read() for f in
                                   ("k_gas_optics_mp.hip", "mp_general_block.inc", "mp_lean_block.inc", "k_gas_optics_far.hip",
                                    "gas_optics_mp_dev.h", "gas_optics_dev.h"))).hexdigest()


def test_round5_counters_belong_to_the_kernel_source_at_head():
    """VERDICT r4 weak 1 / ADVICE r4: round 4 kept lines of two builds side by side.  The PMC counters bench.py carries into
    its line (profiles/traffic_latest.json) name the kernel source they were taken on: it is the source of this tree."""
    t = load("traffic_latest.json")
    assert t["tag"] == "r5" and t["kernel_source_sha256"] == _kernel_source_sha()
    assert t["cols"] == 64 and t["gas_optics_sw"]["sq"]["SQ_INSTS_VALU"] > 0


def test_round5_headline_line():
    line = load("r5_bench_line.json")
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert line["metric"].split(" at ")[0] in base["metric"] and line["unit"] == "columns/s" and line["scaling"] == "weak"
    assert line["n_gpus"] == 1 and line["steps"] * line["ms_per_step"] >= 2500.0 and line["dtype"] == "f64" and line["vs_baseline"] is None
    assert line["ms_per_step"] <= 112.0 and line["value"] >= 570.0                       # VERDICT r4, task 1's speed target
    k = line["kernel_ms_per_step"]
    assert k["gas_optics_lw"] <= 27.0                                                     # ... its longwave target
    assert k["gas_optics_sw"] <= 74.0                                                     # ... its shortwave target (no core-point kernel: fused)
    assert k["far_field_sw"] + k["sw_solver"] <= 7.0                                      # task 6: the shortwave tail
    assert abs(sum(k.values()) - line["ms_per_step"]) / line["ms_per_step"] < 0.03      # the kernels' durations ARE the step
    r = line["roofline"]
    assert r["bound"] == "valu_issue" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.3 < r["frac"] < 1.0
    assert "round tag r5" in r["instructions_source"] and r["traffic"] > 0
    assert r["algorithmic_fp32_frac"] >= 0.28                                             # round 4: 0.25
    v = line["roofline_valu"]
    assert v["instructions_per_64_lines"] <= 520                                          # round 4: 613
    h = line["roofline_hbm"]
    assert h["bound"] == "hbm" and abs(h["frac"] - h["achieved"] / h["peak"]) < 1e-12
    p = line["parity"]
    assert p["kind"] == "reference" and p["ok"] is True and p["max_abs_flux_diff_w_m2"] <= 1e-3 and p["max_tau_err_of_layer_max"] <= 2e-6
    q = p["physical_list"]
    assert q["ok"] is True and q["max_abs_flux_diff_w_m2"] <= 1e-3 and 0.62 < q["rsds_over_rsdt"] < 0.75
    c = line["cpu_baseline"]
    assert c["kind"] == "reference" and c["cores"] >= 1 and c["value"] > 0 and c["one_thread"]["cores"] == 1
    cols = line["config"]["columns_per_step"]
    assert abs(line["value"] - cols / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-9


def test_round5_rccl_at_world_size_one_is_a_pair_taken_back_to_back():
    """VERDICT r4 task 2: forced and plain runs of ONE build, alternating, in one file."""
    pairs = load("r5_rccl_world1_pairs.json")
    assert len(pairs) >= 3
    for x in pairs:
        plain, forced = x["plain"], x["forced"]
        assert plain["rccl_ranks"] == 0 and forced["rccl_ranks"] == 1 and forced["collective"]["backend"] == "nccl"
        assert forced["collective"]["gathers_in_timed_region"] == 1
        assert 0.98 * plain["value"] <= forced["value"] <= 1.005 * plain["value"]


def test_round5_batch_shapes_of_an_eight_gpu_run():
    """VERDICT r4 task 4: 13 columns per rank and step (100 columns over 8 GPUs) must run at >= 0.92 of the 64-column rate."""
    rate = {x["cols"]: x["columns_per_s"] for x in load("r5_batch_sweep.json")}
    assert set(rate) >= {1, 2, 4, 8, 9, 13, 16, 32, 64, 225}
    assert rate[13] >= 0.92 * rate[64] and rate[9] >= 0.92 * rate[64] and rate[225] >= 0.98 * rate[64]


def test_round5_fine_grid_lines_traffic_and_the_unchanged_caller():
    g3 = load("r5_g3_pipeline_bench_line.json")
    assert "0.001" in g3["config"]["workload"] and g3["value"] > 15.0
    band = g3["roofline_hbm_lw_band"]                         # the north-star's "fraction of HBM roofline on the ~3M grid"
    assert band["far_field_kernel"].startswith("gas_optics_tree_kernel") and 0.0 < band["frac_band"] < band["frac_far_field"] < 1.0
    assert abs(band["algorithmic_bytes_per_column"] - 12.87e9) < 0.05e9
    g3_32 = load("r5_g3_pipeline_32_columns_bench_line.json")         # task 5: 32 columns of 18.7 GB of moments each, in column groups
    assert g3_32["config"]["columns_per_step"] == 32 and g3_32["value"] >= 0.95 * g3["value"]
    t = load("traffic_latest.json")["g3"]
    assert t["tag"] == "r5"
    total = sum(k["hbm_bytes_per_launch"] for k in t["kernels"].values())
    assert total <= 80e9                                      # the two-plane store: 75 GB per column (round 4: 100)
    era = load("r5_era5_like_bench_line.json")
    assert "@0.1 cm-1" in era["config"]["workload"] and era["config"]["columns_per_step"] == 64 and era["value"] > 450.0
    strong = load("r5_strong_100_columns_bench_line.json")
    assert strong["scaling"] == "strong" and strong["config"]["columns_per_step"] == 100 and strong["value"] >= 570.0
    drv = load("r5_reference_driver_timing.json")
    assert drv["fast3"]["columns_per_s"] >= 250.0 and drv["fast3rows"]["columns_per_s"] > drv["fast3"]["columns_per_s"]
    two = load("r5_two_streams_bench_line.json")               # the line named "two streams" must have run two
    assert two["config"]["launches_in_flight"] == 2 and two["config"]["chunk_columns"] == 64 and two["value"] >= 570.0


def test_round5_suite_logs_are_green():
    for name, want in (("r5_pytest_gpu.log", "pytest -m gpu rc=0"), ("r5_pytest_gpu_deterministic.log", "deterministic rc=0")):
        text = open(os.path.join(ROOT, "profiles", name)).read()
        assert want in text and " passed" in text and "failed" not in text and "skipped" not in text
    soak = open(os.path.join(ROOT, "profiles", "r5_soak.log")).read()
    assert soak.count("600 passed") == 2 and "wide=1 rc=0" in soak and "wide=2 rc=0" in soak and "failed" not in soak
