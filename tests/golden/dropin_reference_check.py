"""Drive the engine's model classes through the REFERENCE's own entry points, in the build container
(run by tests/test_dropin_reference.py in a subprocess; prints one JSON object).

For each class: Config(model=<class>) -> create_dataset -> data_preparation -> <class>(config, dataset) ->
get_trainer(...)(config, model), the script pattern hopwise documents for foreign models
(src_docs/source/developer_guide/create_models.rst:215-258; configurator.py:219-224; data/utils.py:29-75,227-358;
utils/utils.py:117-125).  No GPU here, so nothing is scored: this checks the dispatch (dataset / loader /
trainer classes), the constructor protocol, and that the reference loader's batch is what the models index.
"""

import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import _refimport  # noqa: E402

_refimport.install()

REF = "/root/reference"


def main():
    from hopwise.config import Config
    from hopwise.data import create_dataset, data_preparation
    from hopwise.utils import InputType, KGDataLoaderState, ModelType, get_model, get_trainer

    import hopwise_amd.model as M
    from hopwise_amd.utils import enum_type

    out = {"using_hopwise_enums": enum_type.USING_HOPWISE, "models": {}}
    yaml = os.path.join(REF, "tests/model/test_model.yaml")
    for name in ("RotatE", "TransE", "TuckER", "PGPR", "CAFE"):
        cls = getattr(M, name)
        rec = {"type_is_reference_member": cls.type is ModelType.KNOWLEDGE,
               "input_type_is_reference_member": isinstance(cls.input_type, InputType)}
        config = Config(model=cls, dataset="test", config_file_list=[yaml],
                        config_dict={"data_path": os.path.join(REF, "tests/test_data"), "train_batch_size": 64,
                                     "train_neg_sample_args": {"distribution": "uniform", "sample_num": 4, "alpha": 1.0,
                                                               "dynamic": False, "candidate_num": 0}}
                        if name in ("RotatE", "TransE") else {"data_path": os.path.join(REF, "tests/test_data")})
        rec["config_model"] = config["model"]
        rec["model_type"] = str(config["MODEL_TYPE"])
        rec["input_type"] = str(config["MODEL_INPUT_TYPE"])
        dataset = create_dataset(config)
        rec["dataset_class"] = type(dataset).__name__
        train_data, valid_data, test_data = data_preparation(config, dataset)
        rec["train_loader"] = type(train_data).__name__
        rec["valid_loader"] = type(valid_data).__name__
        trainer_cls = get_trainer(config["MODEL_TYPE"], config["model"])
        rec["trainer"] = trainer_cls.__name__
        try:
            model = cls(config, train_data.dataset)
            rec["constructed"] = True
            rec["state_dict_keys"] = sorted(model.state_dict().keys())[:12]
            # checkpoint compatibility (trainer.py:296-304): same parameter / buffer names and shapes as hopwise's class
            theirs = get_model(name)(config, train_data.dataset).state_dict()
            mine = model.state_dict()
            rec["state_dict_matches_reference"] = (
                {k: tuple(v.shape) for k, v in mine.items()} == {k: tuple(v.shape) for k, v in theirs.items()})
            if not rec["state_dict_matches_reference"]:
                rec["state_dict_diff"] = sorted(set(mine) ^ set(theirs))[:20]
            trainer = trainer_cls(config, model)
            rec["trainer_built"] = type(trainer.optimizer).__name__
        except RuntimeError as e:  # PGPR builds its float64 score tables on the device in __init__: no GPU here
            rec["constructed"] = False
            rec["error"] = str(e)[:120]
        if name in ("RotatE", "TransE"):
            train_data.set_mode(KGDataLoaderState.RSKG)
            batch = next(iter(train_data))
            rec["batch"] = {k: [list(v.shape), str(v.dtype)] for k, v in batch.interaction.items()}
            # the fields the model indexes are in the batch the reference loader yields
            rec["fields"] = [model.USER_ID, model.ITEM_ID, model.NEG_ITEM_ID, model.HEAD_ENTITY_ID, model.RELATION_ID,
                             model.TAIL_ENTITY_ID, model.NEG_TAIL_ENTITY_ID]
            rec["epoch_len"] = len(train_data)
            rec["n"] = [model.n_users, model.n_items, model.n_entities, model.n_relations]
            rec["device"] = str(config["device"])
            # calculate_loss on a CPU batch must fail loudly (no CPU fallback), not return numbers
            try:
                model.calculate_loss(batch)
                rec["cpu_loss"] = "returned"
            except Exception as e:
                rec["cpu_loss"] = type(e).__name__
        if name in ("PGPR", "CAFE"):
            rec["batch"] = {k: [list(v.shape), str(v.dtype)] for k, v in next(iter(train_data)).interaction.items()}
        out["models"][name] = rec
    out["graphed_trainer"] = graphed_trainer_protocol(yaml)
    print("DROPIN_JSON " + json.dumps(out))


def graphed_trainer_protocol(yaml):
    """hopwise_amd.driver.graphed_trainer(KGTrainer) against KGTrainer itself, on the reference's fixture with the reference's
    OWN TransE on the CPU: the mixin's epoch loop (device-side loss sum, one NaN check, eager fallback for the ragged last
    batch) must leave the same parameters and report the same epoch loss as Trainer._train_epoch.  HIP graphs need a GPU, so
    the captured step is replaced by a stand-in that runs the same four calls eagerly -- the loop around it is what is
    checked here; the capture itself is checked on the GPU (tests/test_gpu_driver.py)."""
    import copy

    import torch
    from hopwise.config import Config
    from hopwise.data import create_dataset, data_preparation
    from hopwise.utils import get_model, get_trainer, init_seed

    from hopwise_amd.driver import graphed_trainer

    cfg = {"data_path": os.path.join(REF, "tests/test_data"), "train_batch_size": 256, "epochs": 1, "embedding_size": 8,
           "train_neg_sample_args": {"distribution": "uniform", "sample_num": 1, "alpha": 1.0, "dynamic": False,
                                     "candidate_num": 0}}
    res = {}
    runs = {}
    for kind in ("base", "graphed"):
        config = Config(model="TransE", dataset="test", config_file_list=[yaml], config_dict=cfg)
        init_seed(config["seed"], config["reproducibility"])
        dataset = create_dataset(config)
        train_data, valid_data, test_data = data_preparation(config, dataset)
        init_seed(config["seed"], config["reproducibility"])
        model = get_model("TransE")(config, train_data.dataset)
        base = get_trainer(config["MODEL_TYPE"], config["model"])
        cls = base
        if kind == "graphed":
            class _StandIn:  # GraphedTrainStep's interface with an eager body
                def __init__(self, trainer, example):
                    self.t, self.shapes = trainer, {k: tuple(v.shape) for k, v in example.interaction.items()}
                    self.calls = 0

                def matches(self, inter):
                    return all(tuple(inter[k].shape) == s for k, s in self.shapes.items())

                def __call__(self, inter):
                    self.calls += 1
                    return self.t._eager_step(inter)

                def check_flags(self):  # read once per epoch by the trainer: nothing a CPU stand-in could have raised
                    pass

            class cls(graphed_trainer(base)):  # noqa: N801
                def _graph_safe(self, loss_func):
                    return loss_func is None

                def _make_step(self, interaction):
                    return _StandIn(self, interaction)

        trainer = cls(config, model)
        res[kind + "_class"] = type(trainer).__name__ if kind == "base" else [c.__name__ for c in type(trainer).__mro__[:3]]
        res[kind + "_optimizer"] = type(trainer.optimizer).__name__
        loss = trainer._train_epoch(train_data, 0)
        runs[kind] = (float(loss) if not isinstance(loss, tuple) else float(sum(loss)),
                      copy.deepcopy({k: v.detach().clone() for k, v in model.state_dict().items()}))
        if kind == "graphed":
            res["replayed_steps"] = trainer._graph_step.calls
            res["epoch_len"] = len(train_data)
    res["loss_base"], res["loss_graphed"] = runs["base"][0], runs["graphed"][0]
    res["max_param_diff"] = max(float((runs["base"][1][k] - runs["graphed"][1][k]).abs().max()) for k in runs["base"][1])
    return res


if __name__ == "__main__":
    main()
