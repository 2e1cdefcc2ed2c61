"""bench.py's sharding plan for the torchrun shapes the driver uses (`--gpus N` with WORLD_SIZE = N): pure arithmetic,
no GPU.  configs[3] / configs[4] are ONE rig whose cameras are sharded over the ranks (strong scaling, the curve
BASELINE.json's north_star asks for); configs[1] / configs[2] give every rank a rig of its own (weak scaling)."""
import pytest
import bench


def test_default_run_is_configs1_on_one_gpu():
    a = bench.parse([])
    assert a.config == 1 and a.gpus == 1 and a.steps > 0 and a.warmup > 0
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
