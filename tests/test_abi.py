"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol
include/gmlm_hip.h declares (no compute calls here: there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT


def _header_symbols():
    hdr = open(os.path.join(ROOT, "include", "gmlm_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(gmlm_[a-z0-9_]+)\s*\(", hdr)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()
    import gmlm_amd
    return gmlm_amd


def test_library_exports_every_declared_symbol(built):
    handle = ctypes.CDLL(built.LIB_PATH)
    names = _header_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/gmlm_hip.h but not exported"
    from gmlm_amd._lib import SIGNATURES
    assert sorted(SIGNATURES) == names, "python binding and header disagree"
    assert built.lib().gmlm_version() == 1
    assert built.lib().gmlm_last_error() is not None


def test_argument_validation_happens_on_the_host(built):
    """Bad arguments are rejected before any launch, so this is safe without a GPU."""
    lib = built.lib()
    rc = lib.gmlm_rgcn_mean_spmm(None, 0, 4, None, None, None, 1, 10, 8, None, 8, 0, 0, None, None, None, 0, 0, None, None)   # stride < f
    assert rc == -1 and b"stride" in lib.gmlm_last_error()
    rc = lib.gmlm_attention_fwd(None, None, None, None, 1, 8, 16, 16, 128, 1024, 1024, 1024, 1.0, 0.0, 0, None, None, None, None, 0, None, 0, None, 0, None)
    assert rc == -1 and b"head dim" in lib.gmlm_last_error()
    rc = lib.gmlm_bias_res_layernorm_fwd(None, None, None, None, None, 4, 770, 1e-5, 0, 0.0, 0, None, None, None, None, 0, None)
    assert rc == -1 and b"multiple" in lib.gmlm_last_error()
    assert lib.gmlm_segment_sort_workspace_bytes(1000) > 3 * 4000


def test_ops_fail_loudly_without_gpu(built):
    with pytest.raises(built.GmlmHipError):
        built.degree(torch.zeros(3, dtype=torch.long), 3)
    with pytest.raises(built.GmlmHipError):
        built.soft_masking_gnn_input(torch.zeros(3, 4), torch.ones(3, dtype=torch.bool), torch.zeros(1, 4))


def test_module_surface_matches_reference_state_dict():
    """Constructor signature, attribute names and state-dict keys of main.GraphTextLM (main.py:183-248)."""
    import inspect
    from transformers import BertConfig, BertModel
    import gmlm_amd
    from helpers import model_state_template
    sig = inspect.signature(gmlm_amd.GraphTextLM.__init__)
    assert list(sig.parameters)[1:9] == ["gnn_in_channels", "hidden_channels", "num_classes", "num_relations", "num_bases",
                                         "dropout_rate", "model_name", "plm_max_length"]
    assert sig.parameters["model_name"].default == "thenlper/gte-base" and sig.parameters["plm_max_length"].default == 256
    fsig = inspect.signature(gmlm_amd.GraphTextLM.forward)
    assert list(fsig.parameters)[1:] == ["gnn_input_features", "edge_index", "all_node_texts", "text_processing_node_mask",
                                         "edge_type", "plm_batch_size"]
    assert list(inspect.signature(gmlm_amd.GraphTextLM.get_graph_embeddings).parameters)[1:] == ["x_feat", "edge_index", "edge_type"]
    plm = dict(hidden=64, layers=2, heads=4, inter=128, max_pos=64, vocab=200)
    enc = BertModel(BertConfig(vocab_size=200, hidden_size=64, num_hidden_layers=2, num_attention_heads=4,
                               intermediate_size=128, max_position_embeddings=64))
    m = gmlm_amd.GraphTextLM(32, 16, 5, plm_encoder=enc)
    want = model_state_template(32, 16, 5, plm)
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert got == {k: tuple(v) for k, v in want.items()}
    assert m.plm_encoder.base_model_prefix == "bert" and m.plm_encoder.config.hidden_size == 64
    # name-based parameter grouping of setup_optimizer (main.py:379-390) sees the same three groups
    gnn = [n for n, _ in m.named_parameters() if any(s in n for s in ("rgcn1", "rgcn2", "rgcn3", "gnorm1", "gnorm2", "gnorm3",
                                                                      "residual_proj"))]
    assert len(gnn) == 4 * 3 + 3 * 3 + 6


def test_kernels_are_built_without_the_slp_vectoriser():
    """csrc/Makefile must keep ``-fno-slp-vectorize``: with the vectoriser's packed-fp32 code the basis backward kernel returned
    wrong coefficient gradients in rare launches while another stream shared the device (DESIGN.md section 5 b,
    profiles/r03_packed_fp32_soak.txt); the GPU replay tests only catch that statistically."""
    import os
    import re
    mk = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gmlm_amd", "csrc", "Makefile")).read()
    assert re.search(r"^CXXFLAGS\s*=.*-fno-slp-vectorize", mk, re.M)
