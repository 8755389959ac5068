"""§8 f4 on the GPU: a dataset file in the reference's ``.npz`` layout (main.py:780-820) and partition shard files feed the
HIP path.  The dataset is written here from the golden fixture g1_toy (whose logits were produced by main.GraphTextLM on
the same features / edges / strings), read back through ``load_npz_dataset`` and run through
``GraphTextLM.forward(x, edge_index, list[str], mask)`` exactly as main.py:545 calls it."""
import numpy as np
import pytest
import torch

from helpers import load_golden, model_state_template, t
from param_recipe import recipe_state_dict
from test_gpu_model import dev, hf_bert  # noqa: F401  (dev: the module-scoped device fixture)

pytestmark = pytest.mark.gpu


def test_npz_dataset_feeds_the_hip_path(dev, tmp_path):
    from transformers import BertTokenizer
    import gmlm_amd
    from gmlm_amd.data import load_npz_dataset
    g = load_golden("g1_toy")
    cfg = g["config"]
    n = cfg["n"]
    mask = g["node_mask"].astype(bool)
    path = tmp_path / "toy.npz"
    np.savez(path, node_features=g["x_soft_masked"], edges=g["edge_index"], node_labels=g["y"],
             node_texts=np.array(g["texts"].tolist()), label_texts=np.array([f"class {i}" for i in range(cfg["c"])]),
             train_masks=mask, val_masks=~mask, test_masks=np.zeros(n, bool))
    data, nf, nc = load_npz_dataset(str(path))
    assert (nf, data.num_nodes) == (cfg["f_in"], n) and data.node_texts == g["texts"].tolist()
    data = data.to(dev)
    tok = BertTokenizer(vocab={w: i for i, w in enumerate(g["vocab"].tolist())})
    m = gmlm_amd.GraphTextLM(nf, cfg["hc"], cfg["c"], dropout_rate=0.0, plm_encoder=hf_bert(cfg["plm"]), plm_tokenizer=tok,
                             plm_max_length=cfg["max_len"])
    m.load_state_dict(recipe_state_dict(model_state_template(cfg["f_in"], cfg["hc"], cfg["c"], cfg["plm"]), cfg["seed"]))
    m = m.to(dev).eval()
    with torch.no_grad():
        logits = m(data.x, data.edge_index, data.node_texts, data.train_mask, plm_batch_size=cfg["plm_batch_size"])
    np.testing.assert_allclose(logits.cpu().numpy(), g["logits"], rtol=0, atol=1e-4)       # the reference's own logits
    assert np.array_equal(m.graph(data.edge_index, n).edge_type.cpu().numpy(), g["edge_type"])
