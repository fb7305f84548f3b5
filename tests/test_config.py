"""Network-config ingestion vs the reference's own tables (golden: tests/golden/tables_*.json,
dumped from the shipped headers and from TF2_auto_config's output by oracle/Makefile)."""
import json
import os

import pytest

from tf2_amd import config as cfg

REF_INC = "/root/reference/Runtime_Engine/cnn/host/inc"
SEM = [k for k in cfg.LAYER_KEYS if k not in ("kDDRReadBase", "kDDRWriteBase", "kDDRWriteEnable")]


def _golden(golden_dir, net):
    return json.load(open(os.path.join(golden_dir, f"tables_{net}.json")))


@pytest.mark.parametrize("net", ["resnet50", "googlenet", "resnet50_pruned"])
@pytest.mark.skipif(not os.path.isdir(REF_INC), reason="reference headers only exist in the build container")
def test_header_parser_matches_compiled_header(golden_dir, net):
    t = cfg.parse_net_header(os.path.join(REF_INC, net + ".h"))
    g = _golden(golden_dir, net)
    for k, v in g.items():
        if k in cfg.ARCH:
            continue
        assert t[k] == v, k


def test_fpganetwork_reader_reproduces_tf2_auto_config(golden_dir):
    layers = cfg.read_fpganetwork(os.path.join(golden_dir, "fpganetwork_resnet50.bin"))
    assert len(layers) == 73
    t = cfg.tables_from_fpganetwork(layers, "resnet50")
    g = _golden(golden_dir, "gen_resnet50")
    for k in SEM:
        assert t[k] == g[k], k
    for k in ("NUM_LAYER", "NUM_CONVOLUTIONS", "NUM_Q_LAYERS", "INPUT_IMAGE_C", "INPUT_IMAGE_H", "INPUT_IMAGE_W",
              "FIRST_FILTER_SIZE", "MAX_OUT_CHANNEL"):
        assert t[k] == g[k], k


def test_generated_and_shipped_resnet50_differ_only_in_unused_pooltype(golden_dir):
    a, b = _golden(golden_dir, "resnet50"), _golden(golden_dir, "gen_resnet50")
    diff = [k for k in a if a[k] != b[k]]
    assert diff == ["kPoolType"]


def _tables_from_golden(g):
    t = cfg.NetTables(g)
    t.setdefault("xConv1Rewrite", 1 if g["FIRST_FILTER_SIZE"] == 7 else 0)
    return t


def test_builder_and_plan_resnet50(golden_dir):
    g = _golden(golden_dir, "resnet50")
    t = cfg.resnet50_tables()
    for k in SEM:
        assert t[k] == g[k], k
    plan_g = cfg.build_plan(_tables_from_golden(g))
    plan_b = cfg.build_plan(t)
    assert plan_g == plan_b
    # residual sources derived from the DDR page plan (feature_writer.cl:88-137)
    adds = {L.index: L.add_src for L in plan_g if L.add_src >= 0}
    assert len(adds) == 16
    assert adds[4] == 1 and adds[7] == 4 and adds[10] == 7 and adds[14] == 11 and adds[52] == 49
    assert cfg.model_float_count(t) == 25610205
    assert cfg.q_value_count(t) == 27563                 # lines of resnet50_Q


def test_googlenet_plan_concat_and_ipool(golden_dir):
    plan = cfg.build_plan(_tables_from_golden(_golden(golden_dir, "googlenet")))
    assert len(plan) == 67
    # inception_3a: tails 3,5,7,9 write slices 0,64,192,224 of concat 0; layer 8 is the ipool
    assert [(plan[i].concat, plan[i].n_start) for i in (3, 5, 7, 9)] == [(0, 0), (0, 64), (0, 192), (0, 224)]
    assert plan[8].ipool == 1 and plan[10].src == -2 and plan[10].q_in_row == 68


def test_other_builders_are_consistent():
    for t in (cfg.squeezenet11_tables(), cfg.vgg16_tables(), cfg.tiny_tables()):
        plan = cfg.build_plan(t)
        for L in plan:
            assert L.OH > 0 and L.PH > 0
        assert cfg.model_float_count(t) > 0 and cfg.q_value_count(t) > 3


def test_expression_evaluator():
    env = dict(cfg.ARCH)
    E = lambda s: cfg._Expr(s, env).parse()
    assert E("CEIL(56, W_VECTOR)") == 8
    assert E("NEXT_POWER_OF_2(FW_VECTOR * C_VECTOR)") == 64
    assert E("(3 > 2) ? 10 : 20") == 10 and E("MYMAX2(4, 9) + NEXT_DIVISIBLE(10, 16)") == 25
    with pytest.raises(cfg.ConfigError):
        E("UNKNOWN_MACRO + 1")


def test_bad_tables_are_rejected():
    t = cfg.tiny_tables()
    t["kFilterSize"] = t["kFilterSize"][:-1]
    with pytest.raises(cfg.ConfigError):
        t.validate()
    with pytest.raises(cfg.ConfigError):
        cfg.read_fpganetwork(b"\0" * 200)
