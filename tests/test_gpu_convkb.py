"""ConvKB on the device against the reference goldens (eval mode and train mode with recorded dropout masks)."""

import os

import numpy as np
import pytest
import torch

from tests.helpers import FakeDataset, make_config
from tests.test_gpu_conve import check_grads
from tests.test_oracle_golden_convkb import FILES, NE, NI, NR, NU, load

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda", 0)


def build(z, p):
    from hopwise_amd.model import ConvKB

    d = p["user_embedding.weight"].shape[1]
    cfg = make_config(dev(), d, 1.0, out_channels=int(z["out_channels"]), kernel_size=int(z["kernel_size"]), dropout_prob=0.3)
    cfg["lambda"] = 0.2
    m = ConvKB(cfg, FakeDataset(NU, NI, NE, NR)).to(dev())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
    return m


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[:-4] for p in FILES])
def test_against_reference_golden(path):
    z, p, inter_np = load(path)
    inter = {k: torch.from_numpy(v).to(dev()) for k, v in inter_np.items()}
    m = build(z, p)
    m.eval()
    with torch.no_grad():
        np.testing.assert_allclose(m.predict(inter).cpu().numpy(), z["eval.predict"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(m.predict_kg(inter).cpu().numpy(), z["eval.predict_kg"], rtol=1e-4, atol=1e-5)
        with pytest.raises(NotImplementedError):
            m.full_sort_predict(inter)
    m.zero_grad()
    loss = m.calculate_loss(inter)
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(z["eval.loss"]), rtol=1e-5)
    check_grads(m, z, "eval.grad")
    m = build(z, p)
    m.train()
    m.forced_masks = [torch.from_numpy(z[f"train.mask{i}"]).to(dev()) for i in range(4)]
    m.zero_grad()
    loss = m.calculate_loss(inter)
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(z["train.loss"]), rtol=1e-5)
    check_grads(m, z, "train.grad")
    for k, v in m.state_dict().items():
        if "running" in k:
            np.testing.assert_allclose(v.cpu().numpy(), z[f"train.after.{k}"], rtol=1e-4, atol=1e-6, err_msg=k)
