"""PGPR training golden: one REINFORCE episode (calculate_loss + backward) of the reference's PGPR on its own
tests/test_data fixture, with every random choice recorded so that it can be replayed:
  * F.dropout -> seeded masks (RecordingDropout), * batch_action_mask's numpy rng seeded, masks recorded,
  * the Categorical draws recorded (select_action wrapper).
Stored: users, per step action masks / sampled actions / rewards / log-probs / values / entropies, the four loss
parts of update() and the gradients of the eight policy tensors.  Called from make_golden.py ("pgpr_train")."""

import os

import numpy as np
import torch
import torch.nn.functional as F
import yaml

from make_golden_pgpr import RecordingDropout, build, enc_paths


def gen_pgpr_train(out_dir):
    real_dropout = F.dropout
    os.makedirs("/tmp/scratch", exist_ok=True)
    os.chdir("/tmp/scratch")
    with open("/root/reference/hopwise/properties/quick_start_config/knowledge_base_on_ml-100k.yaml") as f:
        constraints = yaml.safe_load(f)["path_constraint"]
    for tag, extra in (("a", {"max_acts": 50, "action_dropout": 0.5}),
                       ("b", {"max_acts": 250, "action_dropout": 0.0, "gamma": 0.9, "weight_factor": 0.01})):
        config, ds, model = build(dict(extra, path_constraint=constraints))
        torch.manual_seed(99)
        for m in (model.l1, model.l2, model.actor, model.critic):
            torch.nn.init.normal_(m.bias, std=0.1)
        out = {f"w.{k}": v.numpy().copy() for k, v in model.state_dict().items()}
        out["max_acts"], out["gamma"] = np.int64(model.max_acts), np.float64(model.gamma)
        out["action_dropout"], out["weight_factor"] = np.float64(model.action_dropout), np.float64(model.weight_factor)
        rel2id = ds.field2token_id["relation_id"]
        out["patterns"] = np.array([[1 + len(p) - 1] + [rel2id.get(n, -2) for n in p[1:]] for p in model.patterns], np.int64)
        active = np.unique(ds.inter_feat[ds.uid_field].numpy())
        users = torch.from_numpy(np.concatenate([active[:40], [0]]).astype(np.int64))
        out["users"] = users.numpy()

        model.rng = np.random.default_rng(4321)
        rec = RecordingDropout(55)
        F.dropout = rec
        masks, acts, rewards, stats, parts = [], [], [], {}, {}
        real_mask, real_select, real_step, real_update = (model.batch_action_mask, model.select_action, model.batch_step,
                                                          model.update)

        def mask_wrap(dropout):
            m = real_mask(dropout)
            masks.append(m.copy())
            return m

        def select_wrap(state, mask):
            a = real_select(state, mask)
            acts.append(np.array(a, np.int64))
            return a

        def step_wrap(idx):
            s, r, d = real_step(idx)
            rewards.append(np.array(r, np.float64))
            return s, r, d

        def update_wrap():
            stats["logp"] = np.stack([sa.log_prob.detach().numpy() for sa in model.saved_actions])
            stats["value"] = np.stack([sa.value.detach().numpy()[:, 0] for sa in model.saved_actions])
            stats["entropy"] = np.stack([e.detach().numpy() for e in model.entropy])
            res = real_update()
            parts["v"] = np.array([float(x) for x in res], np.float64)
            return res

        model.batch_action_mask, model.select_action = mask_wrap, select_wrap
        model.batch_step, model.update = step_wrap, update_wrap
        torch.manual_seed(7)
        model.zero_grad()
        loss = model.calculate_loss({model.USER_ID: users})
        loss.backward()
        F.dropout = real_dropout
        out["masks"] = np.stack(masks).astype(np.uint8)
        out["acts"] = np.stack(acts)
        out["rewards"] = np.stack(rewards)
        for k, v in stats.items():
            out[k] = v
        out["loss_parts"] = parts["v"]
        out["loss"] = np.float64(loss.item())
        for i in range(len(acts)):
            out[f"drop1.{i}"], out[f"drop2.{i}"] = rec.masks[2 * i], rec.masks[2 * i + 1]
        for k, p in model.named_parameters():
            out[f"g.{k}"] = p.grad.numpy().copy()
        out["final_paths"] = enc_paths(model._batch_path)
        np.savez_compressed(os.path.join(out_dir, f"pgpr_train_{tag}.npz"), **out)
        print("pgpr_train", tag, "loss", out["loss_parts"], "nonzero rewards", int((out["rewards"] > 0).sum()),
              "acts", out["acts"].shape)
