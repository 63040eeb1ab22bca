"""The drop-in against the REAL reference: hopwise_amd model classes through hopwise's own Config -> create_dataset ->
data_preparation -> get_trainer.  Runs only where /root/reference exists (the build container);
the GPU box has no reference and skips.  A subprocess, because which enums / base classes hopwise_amd binds to is
decided at import time (hopwise_amd/utils/enum_type.py)."""

import json
import os
import subprocess
import sys

import pytest

from hopwise_amd.driver import KGBatcher  # noqa: F401  (layout statement lives there)

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "hopwise")), reason="reference not present")


@pytest.fixture(scope="module")
def report(tmp_path_factory):
    cwd = tmp_path_factory.mktemp("dropin")  # hopwise writes ./log, ./log_tensorboard into cwd
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    env.pop("HOPWISE_AMD_STANDALONE", None)
    p = subprocess.run([sys.executable, os.path.join(HERE, "golden", "dropin_reference_check.py")], cwd=cwd, env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("DROPIN_JSON ")]
    assert line, p.stdout[-2000:]
    return json.loads(line[-1][len("DROPIN_JSON "):])


def test_enums_are_the_references(report):
    assert report["using_hopwise_enums"] is True
    for name, rec in report["models"].items():
        assert rec["type_is_reference_member"] and rec["input_type_is_reference_member"], name
        assert rec["model_type"] == "ModelType.KNOWLEDGE"


@pytest.mark.parametrize("name,loader,trainer,input_type", [
    ("RotatE", "KnowledgeBasedDataLoader", "KGTrainer", "InputType.PAIRWISE"),
    ("TransE", "KnowledgeBasedDataLoader", "KGTrainer", "InputType.PAIRWISE"),
    ("TuckER", "KnowledgeBasedDataLoader", "KGTrainer", "InputType.PAIRWISE"),
    ("PGPR", "UserDataLoader", "PGPRTrainer", "InputType.USERWISE"),
    ("CAFE", "UserDataLoader", "CAFETrainer", "InputType.USERWISE"),
])
def test_reference_dispatch(report, name, loader, trainer, input_type):
    rec = report["models"][name]
    assert rec["config_model"] == name
    assert rec["dataset_class"] == "KnowledgeBasedDataset"  # data/utils.py:55 type2class[ModelType.KNOWLEDGE]
    assert rec["train_loader"] == loader
    assert rec["valid_loader"] == "FullSortRecEvalDataLoader"
    assert rec["trainer"] == trainer  # utils/utils.py:117-125
    assert rec["input_type"] == input_type
    if name == "PGPR":  # its constructor builds float64 score tables on the device: loud failure without a GPU
        assert rec["constructed"] is False and "HIP" in rec["error"]
    else:
        assert rec["constructed"] is True and rec["trainer_built"] == "Adam"
        assert rec["state_dict_matches_reference"], rec.get("state_dict_diff")


@pytest.mark.parametrize("name", ["RotatE", "TransE"])
def test_reference_loader_batch_is_what_the_model_indexes(report, name):
    rec = report["models"][name]
    batch = rec["batch"]
    for f in rec["fields"]:  # USER_ID, ITEM_ID, NEG_ITEM_ID, HEAD..., NEG_TAIL...
        assert f in batch and batch[f] == [[64], "torch.int64"], (f, batch.get(f))
    # KGBatcher's layout for the same settings (train_batch_size 64, sample_num 4): rec rows (64 // 4) * 4, kg rows 64
    assert batch["user_id"][0] == [(64 // 4) * 4] and batch["head_id"][0] == [64]
    assert rec["n"] == [201, 301, 6136, 26]
    # no CPU fallback: a CPU batch is refused, not scored (RuntimeError from the ctypes binding, NotImplementedError from
    # the dispatcher when the registered ops are loaded: no CPU kernel exists)
    assert rec["cpu_loss"] in ("RuntimeError", "NotImplementedError")


def test_graphed_trainer_mixin_follows_the_reference_epoch(report):
    """driver.graphed_trainer(KGTrainer): same epoch loss and same parameters after an epoch as KGTrainer._train_epoch on the
    reference fixture (reference TransE on the CPU, the captured step replaced by an eager stand-in -- the capture itself
    is tested on the GPU); the ragged last batch takes the eager path."""
    g = report["graphed_trainer"]
    assert g["graphed_class"][0].startswith("cls") or "Graphed" in " ".join(g["graphed_class"])
    assert g["base_class"] == "KGTrainer" and g["graphed_class"][1] == "GraphedKGTrainer"
    assert abs(g["loss_base"] - g["loss_graphed"]) <= 1e-5 * abs(g["loss_base"])
    assert g["max_param_diff"] <= 1e-6
    assert 0 < g["replayed_steps"] <= g["epoch_len"]
