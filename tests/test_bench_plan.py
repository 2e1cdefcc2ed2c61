"""bench.py's launcher and sharding plan for the torchrun shapes the driver uses (`--gpus N` with WORLD_SIZE = N): pure arithmetic,
no GPU.  configs[3] / configs[4] are ONE rig whose cameras are sharded over the ranks (strong scaling, the curve
BASELINE.json's north_star asks for); configs[1] / configs[2] give every rank a rig of its own (weak scaling)."""
import json
import os
import subprocess
import sys
import pytest
import bench

BENCH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return env


def _json_lines(text):
    return [json.loads(l) for l in text.splitlines() if l.startswith("{")]


def test_default_run_is_configs1_on_one_gpu():
    a = bench.parse([])
    assert a.config == 1 and a.gpus is None and a.steps > 0 and a.warmup > 0    # --gpus unset: WORLD_SIZE, else 1
    p = bench.plan_for(a.config, 1, 0)
    assert p["name"] == "configs[1]" and p["global_cams"] == [0, 1] and p["scaling"] == "weak" and p["rigs"] == 1
    assert (p["width"], p["height"], p["nfeatures"]) == (640, 480, 1000)


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_weak_configs_give_every_rank_its_own_rig(world):
    for cfg in (1, 2):
        owned = []
        for r in range(world):
            p = bench.plan_for(cfg, world, r)
            assert p["cams_per_rank"] == 2 and p["rigs"] == world and p["scaling"] == "weak" and p["exchange"] == (world > 1)
            owned += p["global_cams"]
        assert owned == list(range(2 * world))


@pytest.mark.parametrize("cfg,cams,worlds", [(3, 4, (1, 2, 4)), (4, 8, (1, 2, 4, 8))])
def test_strong_configs_shard_one_rig_over_the_ranks(cfg, cams, worlds):
    for world in worlds:
        owned = []
        for r in range(world):
            p = bench.plan_for(cfg, world, r)
            assert p["rigs"] == 1 and p["scaling"] == "strong" and p["cams_per_rank"] == cams // world
            owned += p["global_cams"]
        assert owned == list(range(cams))       # every camera exactly once, global camera order == rank order
    assert bench.plan_for(3, 4, 2)["global_cams"] == [2]                       # configs[3]: one camera per GPU
    assert bench.plan_for(4, 1, 0)["text"].startswith("ONE rig of 8 synthetic 1920x1080")
    assert bench.CONFIGS[4]["nfeatures"] == 4000 and bench.CONFIGS[3]["nfeatures"] == 1000


def test_a_world_that_does_not_divide_the_rig_is_refused():
    with pytest.raises(SystemExit):
        bench.plan_for(3, 8, 0)
    with pytest.raises(SystemExit):
        bench.plan_for(4, 3, 0)
    with pytest.raises(SystemExit):
        bench.plan_for(7, 1, 0)


def test_torchrun_arguments_parse_as_the_driver_passes_them():
    a = bench.parse(["--gpus", "4", "--steps", "20", "--warmup", "5", "--config", "3"])
    assert (a.gpus, a.steps, a.warmup, a.config) == (4, 20, 5, 3)
    a = bench.parse(["--config", "4"])
    assert a.steps == 100 and a.warmup == 10       # 1080p steps are milliseconds: shorter default blocks


def test_percentiles():
    xs = list(range(1, 101))
    assert bench.pct(xs, 50) == 50 and bench.pct(xs, 5) == 5 and bench.pct(xs, 95) == 95 and bench.pct([3.0], 95) == 3.0


def test_gpus_n_as_a_plain_process_starts_n_ranks_itself():
    """`python bench.py --gpus 2` without a launcher (how the driver ran round 2's N = 1 line): the parent starts two rank
    processes, they rendezvous (gloo here, RCCL on the GPU box), ONE line comes back and it says n_gpus 2."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--config", "3", "--plan-only"], env=_clean_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = _json_lines(r.stdout)
    assert len(lines) == 1
    assert lines[0]["n_gpus"] == 2 and lines[0]["scaling"] == "strong" and lines[0]["cameras_of_rank"] == [[0, 1], [2, 3]]
    assert lines[0]["exchange"] is True and lines[0]["config"]["cams_per_gpu"] == 2


def test_gpus_n_under_torch_distributed_run():
    """the driver's N > 1 command line: torch.distributed.run sets WORLD_SIZE, bench.py must not launch a second layer"""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", BENCH, "--gpus", "2", "--steps", "20", "--warmup", "5", "--plan-only"],
                       env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["scaling"] == "weak"
    assert lines[0]["cameras_of_rank"] == [[0, 1], [2, 3]] and lines[0]["steps"] == 20 and lines[0]["config"]["rigs"] == 2


def test_one_gpu_plain_process_is_unchanged_and_a_mismatch_is_refused():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--plan-only"], env=_clean_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and _json_lines(r.stdout)[0]["n_gpus"] == 1
    env = dict(_clean_env(), WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--plan-only"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "refusing" in r.stderr and not _json_lines(r.stdout)
    # a rank that dies takes the job down with its status (no line, no hang)
    r = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--config", "3", "--plan-only"], env=_clean_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and not _json_lines(r.stdout)


def test_extractor_algorithmic_bytes_are_the_survey_figures():
    # 1.65 MB / 4.90 MB / 11.01 MB per image
    assert round(bench.extract_alg_bytes(640, 480, 1000) / 1e6, 2) == 1.65
    assert round(bench.extract_alg_bytes(1280, 720, 2000) / 1e6, 2) == 4.90
    assert round(bench.extract_alg_bytes(1920, 1080, 4000) / 1e6, 2) == 11.01


class _FakeClockRt:
    """rt stand-in for _settled_launches: every launch advances a clock by the next duration of a scripted curve."""

    def __init__(self, curve_ms):
        self.curve = list(curve_ms); self.now = 0.0; self.launches = 0
        rt = self

        class Event:
            def record(self, stream):
                self.t = rt.now

            def elapsed_ms(self, other):
                return other.t - self.t
        self.Event = Event

    def launch(self):
        self.now += self.curve[min(self.launches, len(self.curve) - 1)]
        self.launches += 1


def test_settled_launches_times_only_behind_the_transient():
    """The roofline timing waits until three groups of ten launches agree with the three before within 0.3 %, then times `iters`
    launches: a scripted curve that falls for 150 launches and then stands still must give the settled value."""
    curve = [0.300 - 0.001 * i for i in range(100)] + [0.200 - 0.0002 * i for i in range(50)] + [0.190]
    rt = _FakeClockRt(curve)
    ms, used, hist = bench._settled_launches(rt, rt.launch, None, iters=50)
    assert abs(ms - 0.190) < 1e-9 and used % 10 == 0 and 180 <= used <= 400 and len(hist) == used // 10
    # a duration that never settles stops at the group limit and still reports what it timed
    rt = _FakeClockRt([0.5 - 0.0005 * i for i in range(1000)])
    ms, used, hist = bench._settled_launches(rt, rt.launch, None, iters=10, max_groups=12)
    assert used == 120 and 0.0 < ms < 0.5
