"""`python bench.py --gpus N` must start N ranks by itself (the driver's SCALE run calls it exactly like that), give
every rank its own contiguous realization range and reduce the integer counters once.  --launch-check runs that
machinery on gloo with an integer checksum per realization index in place of the kernel, so it runs without a GPU:
two ranks must cover the same indices as one rank with twice the steps and arrive at the same counters."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--launch-check"] + list(argv), env=env,
                         capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout            # ONE JSON line, printed by rank 0 only
    return json.loads(lines[0])


@pytest.mark.timeout(600)
def test_gpus_flag_launches_that_many_ranks_with_disjoint_ranges():
    one = _bench("--gpus", "1", "--steps", "6", "--batch", "512")
    two = _bench("--gpus", "2", "--steps", "3", "--batch", "512")
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert one["rank_ranges"] == [[0, 3072]]
    assert two["rank_ranges"] == [[0, 1536], [1536, 3072]]          # disjoint, contiguous, same union
    assert two["counters"] == one["counters"]
    assert one["counters"]["n_realizations"] == 3072 and one["counters"]["sym_errors"] > 0


@pytest.mark.timeout(600)
def test_strong_split_covers_the_literal_total_once():
    """The strong-scaling leg (BASELINE config 4: 10^6 realizations over the ranks; here 10 007 to keep the remainder
    visible): contiguous blocks floor(T r / N) .. floor(T (r + 1) / N), union = [0, T), reduced counters = the one-rank ones."""
    one = _bench("--gpus", "1", "--steps", "2", "--batch", "512", "--strong-total", "10007")
    two = _bench("--gpus", "2", "--steps", "1", "--batch", "512", "--strong-total", "10007")
    assert one["strong"]["rank_ranges"] == [[0, 10007]]
    assert two["strong"]["rank_ranges"] == [[0, 5003], [5003, 10007]]
    assert two["strong"]["counters"] == one["strong"]["counters"]
    assert one["strong"]["counters"]["n_realizations"] == 10007 and one["strong"]["total_realizations"] == 10007
    default = _bench("--gpus", "2", "--steps", "1", "--batch", "65536")
    assert default["strong"]["total_realizations"] == 10 ** 6           # config 4's literal count
    assert default["strong"]["rank_ranges"] == [[0, 500000], [500000, 1000000]]


@pytest.mark.timeout(900)
def test_eight_ranks_cover_what_one_rank_covers():
    """The driver's N = 8 launch: eight disjoint contiguous ranges whose union is the one-rank range, the
    strong split of config 4's 10^6 and config 5's 10^5 with their remainders (10^6 / 8 is exact, 10^5 / 8 = 12 500 too; the odd
    total 100 003 is not), reduced counters equal to one rank's -- launcher, split and reduction on gloo, no GPU."""
    one = _bench("--gpus", "1", "--steps", "8", "--batch", "512", "--strong-total", "100003")
    eight = _bench("--gpus", "8", "--steps", "1", "--batch", "512", "--strong-total", "100003")
    assert eight["n_gpus"] == 8 and len(eight["rank_ranges"]) == 8
    assert eight["rank_ranges"] == [[512 * r, 512 * (r + 1)] for r in range(8)] and one["rank_ranges"] == [[0, 4096]]
    assert eight["counters"] == one["counters"] and one["counters"]["n_realizations"] == 4096
    sr = eight["strong"]["rank_ranges"]
    assert sr[0][0] == 0 and sr[-1][1] == 100003 and all(a[1] == b[0] for a, b in zip(sr, sr[1:]))
    assert sr == [[(100003 * r) // 8, (100003 * (r + 1)) // 8] for r in range(8)]
    assert {hi - lo for lo, hi in sr} == {12500, 12501}                  # the remainder of 3 goes to three ranks
    assert eight["strong"]["counters"] == one["strong"]["counters"]
    for cfg, total in (("c4", 10 ** 6), ("c5", 10 ** 5)):                # BASELINE.json's literal counts
        lit8, lit1 = eight["strong_literal"][cfg], one["strong_literal"][cfg]
        assert lit8["total_realizations"] == total and lit1["rank_ranges"] == [[0, total]]
        assert lit8["rank_ranges"] == [[(total * r) // 8, (total * (r + 1)) // 8] for r in range(8)]
        assert lit8["counters"] == lit1["counters"] and lit1["counters"]["n_realizations"] == total


def test_world_size_must_match_gpus():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--launch-check", "--gpus", "2"], env=env,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE=1" in (out.stderr + out.stdout)


def test_profile_table_matches_the_bench_legs():
    """a committed profile is taken at the batch AND the demodulator of the bench leg it documents.  One
    table (bench.profile_specs) drives scripts/prof_r06.sh and scripts/collect_profiles.py; here it is held to bench.py's legs:
    every other_workloads leg (BATCH_SURVEY, slicer), the headline at one dispatch of a step, the matrix-core leg."""
    sys.path.insert(0, REPO)
    import bench
    specs = bench.profile_specs()
    for cfg in ("c2", "c3", "c5", "f1", "f6"):
        for dt, tag in (("f32", cfg), ("f64", cfg + "_f64")):
            sp = specs[tag]
            assert sp["batch"] == bench.BATCH_SURVEY[cfg] and sp["demod"] == "slicer" and sp["dtype"] == dt and sp["config"] == cfg
            assert sp["leg"] == "other_workloads.%s.%s" % (cfg, dt) and not sp["opts"]
    assert specs["c4_f64"]["demod"] == "mindist" and specs["c4_f64"]["batch"] == min(bench.BATCH["c4"], 1 << 18)
    assert specs["c4_f64"]["kernel"] == "k_run_mimo_ofdm_pw" and specs["c4_f64_planar"]["kernel"] == "k_run_mimo_ofdm_planar"
    assert specs["c4_f64_qw"]["kernel"] == "k_run_mimo_ofdm_qw" and specs["c4_f64_qw"]["opts"] == ["f64_threads=260"]
    assert specs["c5_f64"]["kernel"] == specs["f6_f64"]["kernel"] == specs["c5"]["kernel"] == specs["f6"]["kernel"] == "k_link_walk<"
    assert specs["c4md_mfma"]["batch"] == bench.BATCH_SURVEY["c4"] and specs["c4md_mfma"]["kernel"] == "k_run_mimo_ofdm_mfma"
    assert specs["c2"]["kernel"] == "k_run_flat_mfma" and specs["c2_f64"]["kernel"] == "k_run_flat"
    # the driver script reads the SAME table
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--profile-spec", "c2"], capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0 and out.stdout.split() == ["--config", "c2", "--dtype", "f32", "--demod", "slicer", "--batch",
                                                          str(bench.BATCH_SURVEY["c2"])]
    script = open(os.path.join(REPO, "scripts", "prof_r06.sh")).read()
    assert "--profile-spec" in script and "--batch" not in script.replace("--batch of", "")
